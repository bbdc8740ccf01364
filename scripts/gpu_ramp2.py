"""Per-step wall time after an idle gap: how long does the GPU take to reach its steady clock?"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from oracle import nrv_oracle as O
m1, m2 = load_species("ecoli"); T = 13
rv = Reviser(m1.with_window(T), m2.with_window(T))
sig, rd = O.synth_windows(4096, T)
ds = torch.from_numpy(sig).cuda(); dr = torch.from_numpy(rd).cuda()
p1 = torch.empty(4096, 6, device="cuda"); p2 = torch.empty(4096, 5, device="cuda")
a1 = torch.empty(4096, dtype=torch.int8, device="cuda"); a2 = torch.empty(4096, dtype=torch.int8, device="cuda")
args = (ds.data_ptr(), dr.data_ptr(), 4096, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
for idle in (0.0, 0.3, 1.0):
    rv.predict_device(*args); rv.sync()
    time.sleep(idle)
    ts = []
    for i in range(120):
        t0 = time.perf_counter(); rv.predict_device(*args); rv.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"idle {idle:.1f}s: step ms:", " ".join(f"{x:.2f}" for x in ts[:12]), "... 20:", f"{ts[20]:.2f}", "40:", f"{ts[40]:.2f}", "80:", f"{ts[80]:.2f}", "119:", f"{ts[119]:.2f}", flush=True)
# queued (no per-step sync), bursts of 10
for idle in (0.3,):
    time.sleep(idle)
    for b in range(8):
        t0 = time.perf_counter()
        for i in range(10): rv.predict_device(*args)
        rv.sync()
        print(f"burst {b}: {(time.perf_counter()-t0)*100:.3f} ms/step", flush=True)
