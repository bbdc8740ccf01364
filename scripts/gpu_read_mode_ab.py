#!/usr/bin/env python3
"""Read mode (windows formed on the device from per-event arrays), device-resident: bases/s and per-kernel microseconds of
several builds of the engine.  usage: python3 scripts/gpu_read_mode_ab.py lib1.so lib2.so ..."""
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

T, N = 11, 200000
m1, m2 = load_species("ecoli")
sig, feat = W.synth_read(N, seed=20265)
dev = "cuda:0"
d_sig, d_feat = torch.from_numpy(sig).to(dev), torch.from_numpy(feat).to(dev)
n = N - T
ref = None
for rep in range(int(os.environ.get("VAR_REPS", "2"))):
    for lib in sys.argv[1:]:
        rv = Reviser(m1, m2, device=0, batch=4096, lib_path=os.path.abspath(lib))
        rv.set_stream(torch.cuda.current_stream().cuda_stream)
        o = (torch.empty(n, 6, device=dev), torch.empty(n, 5, device=dev), torch.empty(n, dtype=torch.int8, device=dev),
             torch.empty(n, dtype=torch.int8, device=dev))
        args = (d_sig.data_ptr(), d_feat.data_ptr(), N) + tuple(x.data_ptr() for x in o)
        for _ in range(5):
            rv.predict_read_device(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            rv.predict_read_device(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        rv.prof_enable(1); rv.prof_read()
        rv.predict_read_device(*args)
        torch.cuda.synchronize()
        k = {nm.split()[0]: m_ / max(c, 1) * 1e3 for nm, (m_, c) in rv.prof_read().items()}
        rv.prof_enable(0)
        out = [x.cpu().numpy() for x in o]
        if ref is None:
            ref = out
        same = all(np.array_equal(a, b) for a, b in zip(out, ref))
        print(f"{os.path.basename(lib):36s} {n / dt / 1e6:6.2f} M bases/s | " + " ".join(f"{a}:{v:6.1f}" for a, v in k.items()) + f" | identical: {same}", flush=True)
        rv.close()
