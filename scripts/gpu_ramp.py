import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from oracle import nrv_oracle as O
t_start = time.time()
m1, m2 = load_species("ecoli")
T, B = 13, 4096
rv = Reviser(m1.with_window(T), m2.with_window(T))
sig, rd = O.synth_windows(B, T)
ds, dr = torch.from_numpy(sig).cuda(), torch.from_numpy(rd).cuda()
p1 = torch.empty(B, 6, device="cuda"); p2 = torch.empty(B, 5, device="cuda")
a1 = torch.empty(B, dtype=torch.int8, device="cuda"); a2 = torch.empty(B, dtype=torch.int8, device="cuda")
print("setup %.1fs" % (time.time() - t_start), flush=True)
t0 = time.time()
while time.time() - t0 < 14:
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(100):
        rv.predict_device(ds.data_ptr(), dr.data_ptr(), B, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
    rv.sync(); b = time.perf_counter()
    print("t=%.2fs  %.4f ms/step" % (time.time() - t0, (b - a) * 10), flush=True)
    if os.environ.get("IDLE"): time.sleep(float(os.environ["IDLE"]))
