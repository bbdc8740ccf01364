"""Where a host-pointer call's wall time goes (NRV_HOST_TRACE=1: registration, pipeline, unregistration) for nrv_predict
at several call sizes.  python3 scripts/gpu_hostpath.py [lib.so]"""
import os, sys, time
import numpy as np
os.environ.setdefault("NRV_HOST_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from nanoreviser_amd import workload as O
m1, m2 = load_species("ecoli")
T = 13
rv = Reviser(m1.with_window(T), m2.with_window(T), lib_path=(os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else None))
sig0, rd0 = O.synth_windows(4096, T)
rv.predict_pair(sig0, rd0)
for G in (8, 32):
    sig, rd = np.tile(sig0, (G, 1, 1)), np.tile(rd0, (G, 1, 1))
    for rep in range(4):
        t0 = time.perf_counter(); rv.predict_pair(sig, rd); dt = time.perf_counter() - t0
        print(f"nrv_predict G={G}: {G*4096/dt/1e6:.2f} M bases/s ({dt*1e3:.3f} ms)", flush=True)
