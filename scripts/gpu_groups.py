#!/usr/bin/env python3
"""Launch-group size: device-resident rates of window mode (C4), read mode (C5: one 200 k-event read, human weights,
T = 13) and the fixture reads (C3 shape, T = 11) at `batch` = 512 ... 16384 windows per launch group, plus the HBM
footprint of the handle.  python3 scripts/gpu_groups.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanoreviser_amd.engine import Reviser  # noqa: E402
from nanoreviser_amd.weights import load_species  # noqa: E402
from nanoreviser_amd import workload as W  # noqa: E402

dev = "cuda:0"
T = 13
e1, e2 = load_species("ecoli")
h1, h2 = load_species("human")
N = 200_000
sev, fev = W.synth_read(N)
d_sev, d_fev = torch.from_numpy(sev).to(dev), torch.from_numpy(fev).to(dev)
nw = 65536
sig, rd = W.synth_windows(nw, T, seed=20260)
d_sig, d_rd = torch.from_numpy(sig).to(dev), torch.from_numpy(rd).to(dev)


def outs(n):
    return (torch.empty(n, 6, device=dev), torch.empty(n, 5, device=dev), torch.empty(n, dtype=torch.int8, device=dev),
            torch.empty(n, dtype=torch.int8, device=dev))


def rate(fn, n, reps=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return n * reps / (time.perf_counter() - t0)


res = {}
for batch in (512, 4096, 8192, 16384, 32768):
    free0 = torch.cuda.mem_get_info()[0]
    rv = Reviser(e1.with_window(T), e2.with_window(T), device=0, batch=batch)
    rv.set_stream(torch.cuda.current_stream().cuda_stream)
    o = outs(nw)
    r_win = rate(lambda: rv.predict_device(d_sig.data_ptr(), d_rd.data_ptr(), nw, *[x.data_ptr() for x in o]), nw)
    rv.close()
    rv = Reviser(h1.with_window(T), h2.with_window(T), device=0, batch=batch)
    rv.set_stream(torch.cuda.current_stream().cuda_stream)
    o = outs(N - T)
    r_read = rate(lambda: rv.predict_read_device(d_sev.data_ptr(), d_fev.data_ptr(), N, *[x.data_ptr() for x in o]), N - T)
    used = (free0 - torch.cuda.mem_get_info()[0]) / 2 ** 20
    rv.close()
    res[batch] = {"window_mode_M_bases_s": round(r_win / 1e6, 3), "read_mode_C5_M_bases_s": round(r_read / 1e6, 3),
                  "handle_MiB": round(used)}
    print(batch, res[batch], flush=True)
print(json.dumps(res))
