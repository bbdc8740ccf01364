#!/usr/bin/env python3
"""Where a step of lstm_h2s_kernel (the 256->64 layer; until round 5 also the 192->128 one, with -DNRV_L3_WS=0: that switch is gone since r06) spends its cycles: reads the s_memtime stamps of the diagnostic build
(tools/lstm_exp.sh stamp -> csrc/exp/libnanorev_hip_stamp.so, -DNRV_STAMP=1) after a few hundred bench steps on the
bench's own synthetic windows (the product's data, so the product's clock).  Read SHARES, not lengths.
  python3 scripts/gpu_stamps.py [lib.so] > gpurun_out/stamps.json
Slots per (workgroup, wave, step): 0 loop top | 1..4 rec k-block kr | 8..15 in k-block kk | 24 before the barrier |
25 behind it | 26 end of in() | 27 end of the step (Z <- N).  Step row 14: s_memrealtime at kernel start / end."""
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

lib = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else
                      os.path.join(ROOT, "nanoreviser_amd", "csrc", "exp", "libnanorev_hip_stamp.so"))
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

NBLK, NW, NS, NSL = 256, 4, 15, 32
buf = np.zeros((2, NBLK, NW, NS, NSL), dtype=np.uint64)
cl = C.CDLL(lib)
cl.nrv_exp_stamps.argtypes = [C.c_void_p, C.c_size_t]
rc = cl.nrv_exp_stamps(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
st = buf.astype(np.int64)


def med(x):
    return float(np.median(x))


out = {}
# The 192->128 layer runs lstm_h2w_kernel by default (its stamps: scripts/gpu_stamps_w.py); STAMP_L3_H2S=1 reads its
# region in this layout too, for a library built with -DNRV_L3_WS=0 (rounds 4-5 only).
layers = (("lstm3 192->128", 6, 4), ("lstm4 256->64", 8, 2))
for li, (name, kk_in, kk_rec) in enumerate(layers):
    if li == 0 and os.environ.get("STAMP_L3_H2S", "0") != "1":
        continue
    s = st[li]                                       # [blk][wave][step][slot]
    rt = s[:, :, NS - 1, :2]
    wall_us = (rt[..., 1] - rt[..., 0]) / 100.0      # s_memrealtime: 100 MHz
    tot_cyc = s[:, :, T - 1, 27] - s[:, :, 0, 0]
    ghz = tot_cyc / wall_us / 1e3
    steps = slice(1, T - 1)                          # steady steps: rec() and in() both present
    top, end = s[:, :, steps, 0], s[:, :, steps, 27]
    r = {"kernel_wall_us_median": med(wall_us), "loop_cycles_median": med(tot_cyc), "clock_ghz_median": med(ghz),
         "clock_ghz_min_max": [float(ghz.min()), float(ghz.max())], "step_cycles": med(end - top)}
    rec_edges = [s[:, :, steps, 1 + k] for k in range(kk_rec)] + [s[:, :, steps, 8]]
    r["mk_base"] = med(rec_edges[0] - top)
    r["rec_blocks"] = [med(rec_edges[k + 1] - rec_edges[k]) for k in range(kk_rec)]
    in_edges = [s[:, :, steps, 8 + k] for k in range(kk_in)] + [s[:, :, steps, 26]]
    bar0, bar1 = s[:, :, steps, 24], s[:, :, steps, 25]
    blocks = []
    for k in range(kk_in):
        d = in_edges[k + 1] - in_edges[k]
        inside = (bar0 >= in_edges[k]) & (bar0 < in_edges[k + 1])
        d = np.where(inside, d - (bar1 - bar0), d)   # the barrier wait is reported on its own
        blocks.append(med(d))
    r["in_blocks_without_barrier_wait"] = blocks
    r["barrier_block"] = int(np.median(np.argmax(np.stack([(bar0 >= in_edges[k]) & (bar0 < in_edges[k + 1])
                                                           for k in range(kk_in)]), axis=0)))
    wait = bar1 - bar0
    r["barrier_wait"] = {"median": med(wait), "mean": float(wait.mean()),
                         "by_wave_mean": [float(wait[:, w].mean()) for w in range(NW)]}
    arr = bar0 - bar0.min(axis=1, keepdims=True)     # arrival skew inside a workgroup
    r["barrier_arrival_skew_max_mean"] = float(arr.max(axis=1).mean())
    r["z_from_n"] = med(end - in_edges[-1])
    r["rec_total"] = med(rec_edges[-1] - rec_edges[0])
    r["in_total"] = med(in_edges[-1] - in_edges[0])
    r["last_step"] = {"rec": med(s[:, :, T - 1, 26] - s[:, :, T - 1, 1]), "gates_copy": med(s[:, :, T - 1, 27] - s[:, :, T - 1, 26])}
    r["step0"] = med(s[:, :, 0, 27] - s[:, :, 0, 0])
    r["prologue_to_loop_by_wall"] = None
    ticks_rec, ticks_in = (96, 96) if li == 0 else (48, 48)
    r["mfma_pipe_cycles_per_block"] = ticks_rec * 16
    out[name] = r
print(json.dumps(out, indent=1))
