#!/usr/bin/env python3
"""Time several builds of the engine (same C-ABI) in ONE process: per-kernel microseconds of the bench step
(4096 synthetic 13-event windows) from hipEvent brackets, and the un-bracketed step time.
  python3 scripts/gpu_variants.py nanoreviser_amd/csrc/libnanorev_hip.so nanoreviser_amd/csrc/exp/libnanorev_hip_*.so"""
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

T, B = 13, 4096
m1, m2 = load_species("ecoli")
m1, m2 = m1.with_window(T), m2.with_window(T)
sig, rd = W.synth_windows(B, T, seed=20260)
dev = "cuda:0"
d_sig, d_rd = torch.from_numpy(sig).to(dev), torch.from_numpy(rd).to(dev)
ref = None
prec = os.environ.get("VAR_PRECISION", "f16x2")
for rep in range(int(os.environ.get("VAR_REPS", "1"))):
    for lib in sys.argv[1:]:
        rv = Reviser(m1, m2, device=0, batch=B, precision=prec, lib_path=os.path.abspath(lib))
        rv.set_stream(torch.cuda.current_stream().cuda_stream)
        o = (torch.empty(B, 6, device=dev), torch.empty(B, 5, device=dev), torch.empty(B, dtype=torch.int8, device=dev),
             torch.empty(B, dtype=torch.int8, device=dev))
        ptrs = (d_sig.data_ptr(), d_rd.data_ptr(), B) + tuple(x.data_ptr() for x in o)
        for _ in range(300):
            rv.predict_device(*ptrs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            rv.predict_device(*ptrs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 10
        rv.prof_enable(1); rv.prof_read()
        for _ in range(32):
            rv.predict_device(*ptrs)
        torch.cuda.synchronize()
        k = {n.split()[0]: m_ / max(c, 1) * 1e3 for n, (m_, c) in rv.prof_read().items()}
        rv.prof_enable(0)
        out = [x.cpu().numpy() for x in o]
        if ref is None:
            ref = out
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out, ref))
        dmax = max(float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) for a, b in zip(out[:2], ref[:2]))
        print(f"{os.path.basename(lib):44s} step {ms:.4f} ms | " + " ".join(f"{n}:{v:6.1f}" for n, v in k.items()) +
              f" | bit-identical to first: {same} (max|dp| {dmax:.1e})", flush=True)
        rv.close()
