"""Per-kernel times of the bench workload with idle gaps between steps (is the back-to-back loop power-capped?)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from nanoreviser_amd import workload as W
T, B = 13, 4096
m1, m2 = load_species("ecoli")
rv = Reviser(m1.with_window(T), m2.with_window(T), device=0, batch=B, precision="f16x2")
rv.set_stream(torch.cuda.current_stream().cuda_stream)
sig, rd = W.synth_windows(B, T, seed=20260)
d_sig, d_rd = torch.from_numpy(sig).cuda(), torch.from_numpy(rd).cuda()
p1, p2 = torch.empty(B, 6, device="cuda"), torch.empty(B, 5, device="cuda")
a1, a2 = torch.empty(B, dtype=torch.int8, device="cuda"), torch.empty(B, dtype=torch.int8, device="cuda")
ptrs = (d_sig.data_ptr(), d_rd.data_ptr(), B, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
for _ in range(300):
    rv.predict_device(*ptrs)
torch.cuda.synchronize()
for gap_ms in (0.0, 0.5, 2.0, 10.0, 0.0):
    rv.prof_enable(1); rv.prof_read()
    for _ in range(200 if gap_ms < 5 else 60):
        rv.predict_device(*ptrs)
        if gap_ms:
            torch.cuda.synchronize(); time.sleep(gap_ms * 1e-3)
    torch.cuda.synchronize()
    k = {a: round(b / max(c, 1) * 1e3, 1) for a, (b, c) in rv.prof_read().items() if c > 0}
    print(json.dumps({"gap_ms": gap_ms, "sum_us": round(sum(k.values()), 1), "kernel_us": k}), flush=True)
