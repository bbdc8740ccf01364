"""Host-pointer entry points with small launch groups: lanes on / off, results vs groups of 4096."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from nanoreviser_amd import workload as O
m1, m2 = load_species("ecoli")
T = 11
n = 40000
sig, rd = O.synth_windows(n, T)
N = 120_000
rng = np.random.default_rng(0)
sig_ev = np.clip(rng.normal(-0.1, 1.36, (N, 50)), -8.4, 4.8).astype(np.float32)
feat_ev = np.abs(rng.normal(0.5, 0.3, (N, 6))).astype(np.float32)
rv = Reviser(m1, m2)
ref_w = rv.predict_pair(sig, rd); ref_r = rv.predict_read(sig_ev, feat_ev)
rv.close()
for lanes in ("1", "0"):
    os.environ["NRV_LANES"] = lanes
    for batch in (512, 2048):
        rv = Reviser(m1, m2, batch=batch)
        rv.predict_pair(sig[:4096], rd[:4096])
        t0 = time.perf_counter(); w = rv.predict_pair(sig, rd); dw = time.perf_counter() - t0
        t0 = time.perf_counter(); r = rv.predict_read(sig_ev, feat_ev); dr = time.perf_counter() - t0
        same = all(np.array_equal(a, b) for a, b in zip(w, ref_w)) and all(np.array_equal(a, b) for a, b in zip(r, ref_r))
        print(f"lanes={lanes} batch={batch}: nrv_predict {n / dw / 1e6:.2f} M bases/s, nrv_predict_read {(N - T) / dr / 1e6:.2f} M bases/s, identical to groups of 4096: {same}", flush=True)
        rv.close()
