"""Times one build of the engine (NRV_LIB selects it) on the bench workload WITHOUT checking results:
for the experiment builds of tools/lstm_exp.sh, whose results are wrong by construction.  Prints the
step time and the per-kernel times.  Not a benchmark: bench.py refuses to time a wrong result."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nanoreviser_amd.engine import Reviser  # noqa: E402
from nanoreviser_amd.weights import load_species  # noqa: E402
from nanoreviser_amd import workload as W  # noqa: E402

T, B = 13, int(os.environ.get("EXP_BATCH", "4096"))
m1, m2 = load_species("ecoli")
m1, m2 = m1.with_window(T), m2.with_window(T)
rv = Reviser(m1, m2, device=0, batch=B, precision=os.environ.get("EXP_PREC", "f16x2"))
rv.set_stream(torch.cuda.current_stream().cuda_stream)
sig, rd = W.synth_windows(B, T, seed=20260)
d_sig, d_rd = torch.from_numpy(sig).cuda(), torch.from_numpy(rd).cuda()
p1, p2 = torch.empty(B, 6, device="cuda"), torch.empty(B, 5, device="cuda")
a1, a2 = torch.empty(B, dtype=torch.int8, device="cuda"), torch.empty(B, dtype=torch.int8, device="cuda")
ptrs = (d_sig.data_ptr(), d_rd.data_ptr(), B, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
for _ in range(300):
    rv.predict_device(*ptrs)
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    rv.predict_device(*ptrs)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
rv.prof_enable(1)
rv.prof_read()
for _ in range(32):
    rv.predict_device(*ptrs)
torch.cuda.synchronize()
k = {a: round(b / max(c, 1) * 1e3, 1) for a, (b, c) in rv.prof_read().items() if c > 0}
print(json.dumps({"lib": os.path.basename(os.environ.get("NRV_LIB", "product")), "ms_per_step": round(ms, 4), "kernel_us": k}))
