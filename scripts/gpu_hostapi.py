"""PCIe-inclusive rates of the host-pointer entry points (never bench.py's `value`)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from nanoreviser_amd import workload as O
m1, m2 = load_species("ecoli")
T = 13
rv = Reviser(m1.with_window(T), m2.with_window(T))
n = 65536
sig, rd = O.synth_windows(n, T)
rv.predict_pair(sig[:4096], rd[:4096])
for rep in range(2):
    t0 = time.perf_counter(); rv.predict_pair(sig, rd); dt = time.perf_counter() - t0
    print(f"nrv_predict      (window mode, {n} windows, {sig.nbytes/1e6:.0f}+{rd.nbytes/1e6:.0f} MB host->device): {n/dt/1e6:.2f} M bases/s", flush=True)
N = 200_000
rng = np.random.default_rng(0)
sig_ev = np.clip(rng.normal(-0.1, 1.36, (N, 50)), -8.4, 4.8).astype(np.float32)
feat_ev = np.abs(rng.normal(0.5, 0.3, (N, 6))).astype(np.float32)
for rep in range(2):
    t0 = time.perf_counter(); rv.predict_read(sig_ev, feat_ev); dt = time.perf_counter() - t0
    print(f"nrv_predict_read (read mode, {N} events, {sig_ev.nbytes/1e6:.0f} MB host->device): {(N-T)/dt/1e6:.2f} M bases/s", flush=True)
