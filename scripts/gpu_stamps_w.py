#!/usr/bin/env python3
"""Where a step of lstm_h2w_kernel (eight waves, two groups) spends its cycles: reads the s_memtime stamps of a
diagnostic build (-DNRV_STAMP=1) after a few hundred bench steps on the bench's synthetic windows.
  python3 scripts/gpu_stamps_w.py lib.so > gpurun_out/stamps_w.json
Slots per (workgroup, wave, loop iteration): 0 top (second half of step s) | 1 A: gates done, B: in() done |
2 second half done | 3 behind barrier 2 | 4 rec() starts | 5 rec() done | 6 behind barrier 1 | 7..11 in() k blocks 1..5 start |
12..14 rec() k blocks 1..3 start | 15 arrival at barrier 1 (7..15: s_memtime not waited for until the end of the
iteration).
Step row 14: s_memrealtime at kernel start / end.  Read SHARES, not lengths (the stamps' fences forbid some overlap)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanoreviser_amd.engine import Reviser  # noqa: E402
from nanoreviser_amd.weights import load_species  # noqa: E402
from nanoreviser_amd import workload as W  # noqa: E402

lib = os.path.abspath(sys.argv[1])
T, B = 13, 4096
m1, m2 = load_species("ecoli")
m1, m2 = m1.with_window(T), m2.with_window(T)
sig, rd = W.synth_windows(B, T, seed=20260)
dev = "cuda:0"
d_sig, d_rd = torch.from_numpy(sig).to(dev), torch.from_numpy(rd).to(dev)
rv = Reviser(m1, m2, device=0, batch=B, precision="f16x2", lib_path=lib)
rv.set_stream(torch.cuda.current_stream().cuda_stream)
o = (torch.empty(B, 6, device=dev), torch.empty(B, 5, device=dev), torch.empty(B, dtype=torch.int8, device=dev),
     torch.empty(B, dtype=torch.int8, device=dev))
ptrs = (d_sig.data_ptr(), d_rd.data_ptr(), B) + tuple(x.data_ptr() for x in o)
for _ in range(int(os.environ.get("STAMP_STEPS", "600"))):
    rv.predict_device(*ptrs)
torch.cuda.synchronize()

NBLK, NS = 256, 15
buf = np.zeros((2, NBLK, 4, NS, 32), dtype=np.uint64)
cl = C.CDLL(lib)
cl.nrv_exp_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert cl.nrv_exp_stamps(buf.ctypes.data, buf.nbytes) == 0
s = buf[0].reshape(NBLK, 8, NS, 16).astype(np.int64)   # [blk][wave][iteration][slot]


def med(x):
    return float(np.median(x))


rt = s[:, :, NS - 1, :2]
wall_us = (rt[..., 1] - rt[..., 0]) / 100.0
it = slice(1, T - 2)                                      # steady iterations
out = {"kernel_wall_us_median": med(wall_us)}
tot = s[:, :, T - 2, 6] - s[:, :, 0, 0]
out["loop_cycles_median"] = med(tot)
out["clock_ghz_median"] = med(tot / wall_us / 1e3)
out["iteration_cycles"] = med(s[:, :, 2:T - 1, 0] - s[:, :, 1:T - 2, 0])
for name, ws in (("group_A", slice(0, 4)), ("group_B", slice(4, 8))):
    g = s[:, ws, it, :]
    out[name] = {
        "first_part (A gates | B in)": med(g[..., 1] - g[..., 0]),
        "second_part (A in | B gates)": med(g[..., 2] - g[..., 1]),
        "wait_barrier2": med(g[..., 3] - g[..., 2]),
        "copy_out_and_requests": med(g[..., 4] - g[..., 3]),
        "rec": med(g[..., 5] - g[..., 4]),
        "gates_in_front_of_barrier1 (A, deferred)": med(g[..., 15] - g[..., 5]),
        "wait_barrier1": med(g[..., 6] - g[..., 15]),
    }
    in0 = g[..., 0] if name == "group_B" else g[..., 1]          # in() starts at the top (B) or behind the gates (A)
    in_end = g[..., 1] if name == "group_B" else g[..., 2]
    e = [in0] + [g[..., 7 + k] for k in range(5)] + [in_end]
    out[name]["in_blocks"] = [med(e[k + 1] - e[k]) for k in range(6)]
    r = [g[..., 4]] + [g[..., 12 + k] for k in range(3)] + [g[..., 5]]
    out[name]["rec_blocks"] = [med(r[k + 1] - r[k]) for k in range(4)]
    out[name]["deferred_stamps_monotonic"] = bool(all((e[k + 1] >= e[k]).all() for k in range(6)) and
                                                  all((r[k + 1] >= r[k]).all() for k in range(4)))
if os.environ.get("STAMP_REC_ENTRIES") == "1":             # one-off build -DNRV_STAMP_REC_ENTRIES=1: slots 7..14 = entries 1..8 of rec()
    for name, ws in (("group_A", slice(0, 4)), ("group_B", slice(4, 8))):
        g = s[:, ws, it, :]
        e = [g[..., 4]] + [g[..., 7 + k] for k in range(8)]
        out[name]["rec_entries_0_7"] = [med(e[k + 1] - e[k]) for k in range(8)]
json.dump(out, sys.stdout, indent=1)
print()
