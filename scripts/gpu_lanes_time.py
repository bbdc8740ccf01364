"""Throughput of small launch groups with and without stream lanes (device-resident read mode, E. coli, T = 11)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
m1, m2 = load_species("ecoli")
T, N = 11, 200_000
g = torch.Generator(device="cuda").manual_seed(5)
sig_ev = (torch.randn(N, 50, device="cuda", generator=g) * 1.36 - 0.10).clamp_(-8.4, 4.8)
feat_ev = torch.rand(N, 6, device="cuda", generator=g)
k = N - T
o = (torch.empty(k, 6, device="cuda"), torch.empty(k, 5, device="cuda"),
     torch.empty(k, dtype=torch.int8, device="cuda"), torch.empty(k, dtype=torch.int8, device="cuda"))
for lanes in ("1", "0"):
    os.environ["NRV_LANES"] = lanes
    for batch in (256, 512, 1024, 2048, 4096):
        rv = Reviser(m1, m2, batch=batch)
        for _ in range(3):
            rv.predict_read_device(sig_ev.data_ptr(), feat_ev.data_ptr(), N, *[x.data_ptr() for x in o])
        rv.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            rv.predict_read_device(sig_ev.data_ptr(), feat_ev.data_ptr(), N, *[x.data_ptr() for x in o])
        rv.sync()
        dt = (time.perf_counter() - t0) / 5
        print(f"lanes={lanes} batch={batch}: {k / dt / 1e6:.2f} M bases/s", flush=True)
        rv.close()
