#!/usr/bin/env python3
"""Socket power and shader clock per launch of the f16x2 step: the diagnostic build (-DNRV_STAMP=1) launches ONE stage over and
over for a few seconds (nrv_exp_only_stage) while rocm-smi is sampled.  Stage -1 = the whole step.
  python3 scripts/gpu_power_stage.py lib.so > gpurun_out/power_stage.json"""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import threading
import time

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
cl = C.CDLL(lib)
for _ in range(50):
    rv.predict_device(*ptrs)                          # every buffer holds real data from here on
torch.cuda.synchronize()

samples = []
stop = False


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        p = re.search(r"Package Power \(W\): ([0-9.]+)", out)
        c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        if p and c:
            samples.append((time.time(), float(p.group(1)), int(c.group(1))))


names = {-1: "whole step", 0: "cnn_r_kernel (+ lstm1)", 2: "lstm2_u_kernel", 3: "lstm_h2w_kernel (192->128)",
         4: "lstm_h2s_kernel (256->64)", 5: "head_h2_kernel"}
res = {}
for k in [int(x) for x in os.environ.get("POWER_STAGES", "-1,0,2,3,4,5").split(",")]:
    cl.nrv_exp_only_stage(k)
    torch.cuda.synchronize()
    samples.clear()
    stop = False
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < 4.0:
        for _ in range(200):
            rv.predict_device(*ptrs)
        n += 200
        torch.cuda.synchronize()
    dt = time.time() - t0
    stop = True
    th.join()
    late = [s for s in samples if s[0] - t0 > 1.5]     # the settled part
    res[names[k]] = {"launch_or_step_us": dt / n * 1e6, "samples": len(late),
                     "power_w_median": float(np.median([s[1] for s in late])) if late else None,
                     "sclk_mhz_median": float(np.median([s[2] for s in late])) if late else None}
cl.nrv_exp_only_stage(-1)
json.dump(res, sys.stdout, indent=1)
print()
