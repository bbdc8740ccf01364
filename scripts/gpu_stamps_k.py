#!/usr/bin/env python3
"""Where a step of lstm_h2k_kernel (256->64 layer: eight waves, the reduction split between the two waves of a SIMD) spends
its cycles: s_memtime stamps of a diagnostic build (-DNRV_STAMP=1) after a few hundred bench steps on the bench's windows.
  python3 scripts/gpu_stamps_k.py lib.so > gpurun_out/stamps_k.json
Slots per (workgroup, wave, step): 0 top | 1 A: rec() + parking done, B: in_B() done | 2 A: staging stores / requests, B:
copy-out issued | 3 behind barrier Y | 4 A: in_A() done, B: gates + staging done | 5 behind barrier X.  Step row 14: s_memrealtime at kernel start / end.  Read SHARES, not lengths."""
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
s = buf[1].reshape(NBLK, 8, NS, 16).astype(np.int64)   # [blk][wave][step][slot]


def med(x):
    return float(np.median(x))


rt = s[:, :, NS - 1, :2]
wall_us = (rt[..., 1] - rt[..., 0]) / 100.0
it = slice(1, T - 2)
out = {"kernel_wall_us_median": med(wall_us)}
tot = s[:, :, T - 1, 5] - s[:, :, 0, 0]
out["loop_ticks_median"] = med(tot)
out["step_ticks"] = med(s[:, :, 2:T - 1, 0] - s[:, :, 1:T - 2, 0])
for name, ws, n1, n2, n4 in (("group_A", slice(0, 4), "rec_and_park", "staging", "in_A"),
                            ("group_B", slice(4, 8), "in_B", "copy_out", "gates_and_staging")):
    g = s[:, ws, it, :]
    out[name] = {n1: med(g[..., 1] - g[..., 0]), n2: med(g[..., 2] - g[..., 1]), "wait_barrier_Y": med(g[..., 3] - g[..., 2]),
                 n4: med(g[..., 4] - g[..., 3]), "wait_barrier_X": med(g[..., 5] - g[..., 4])}
gA, gB = s[:, 0:4, it, :], s[:, 4:8, it, :]
eb = [gB[..., 0], gB[..., 6], gB[..., 7], gB[..., 8], gB[..., 1]]
out["group_B"]["in_B_blocks"] = [med(eb[k + 1] - eb[k]) for k in range(4)]
out["group_A"]["rec_blocks"] = [med(gA[..., 10] - gA[..., 0]), med(gA[..., 1] - gA[..., 10])]
ea = [gA[..., 3], gA[..., 11], gA[..., 12], gA[..., 13], gA[..., 4]]
out["group_A"]["in_A_blocks"] = [med(ea[k + 1] - ea[k]) for k in range(4)]
json.dump(out, sys.stdout, indent=1)
print()
