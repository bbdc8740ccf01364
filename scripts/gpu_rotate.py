"""Why a host-pointer stage costs ~8 % more than a device-resident step: the device-resident loop of bench.py with
(a) one input buffer (bench.py's loop), (b) three rotating input buffers, (c) one buffer + a concurrent 12 MB H2D per step on
a side stream, (d) both.   python3 scripts/gpu_rotate.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from nanoreviser_amd import workload as W
T, B = 13, 4096
m1, m2 = load_species("ecoli")
rv = Reviser(m1.with_window(T), m2.with_window(T), device=0, batch=B)
rv.set_stream(torch.cuda.current_stream().cuda_stream)
dev = "cuda:0"
bufs = []
for k in range(3):
    sig, rd = W.synth_windows(B, T, seed=20260 + k)
    bufs.append((torch.from_numpy(sig).to(dev), torch.from_numpy(rd).to(dev)))
o = (torch.empty(B, 6, device=dev), torch.empty(B, 5, device=dev), torch.empty(B, dtype=torch.int8, device=dev), torch.empty(B, dtype=torch.int8, device=dev))
host = torch.from_numpy(np.tile(W.synth_windows(B, T, seed=1)[0], (1, 1, 1))).pin_memory()
side = torch.cuda.Stream()
dst = [torch.empty_like(host, device=dev) for _ in range(3)]
def run(nbuf, h2d, steps=300):
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            if h2d:
                with torch.cuda.stream(side):
                    dst[i % 3].copy_(host, non_blocking=True)
            s, r = bufs[i % nbuf]
            rv.predict_device(s.data_ptr(), r.data_ptr(), B, *(x.data_ptr() for x in o))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    return dt * 1e3
for _ in range(600):
    rv.predict_device(bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), B, *(x.data_ptr() for x in o))
torch.cuda.synchronize()
for rep in range(2):
    print(f"one buffer {run(1, False):.4f} ms | three rotating {run(3, False):.4f} | one + H2D 12 MB/step {run(1, True):.4f} | three + H2D {run(3, True):.4f}", flush=True)
