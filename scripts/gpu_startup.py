#!/usr/bin/env python3
"""Where the command line's start-up goes (one process, no torch): imports, weight files, nrv_create, first call."""
import os, sys, time
t0 = time.perf_counter()
os.environ.setdefault("NRV_NO_TORCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
t1 = time.perf_counter()
from nanoreviser_amd import cli, hoststage as hs
from nanoreviser_amd.weights import load_species
t2 = time.perf_counter()
m1, m2 = load_species("ecoli")
t3 = time.perf_counter()
from nanoreviser_amd.engine import Reviser, load_library
lib = load_library()
t4 = time.perf_counter()
n = lib.nrv_device_count()
t5 = time.perf_counter()
rv = Reviser(m1, m2)
t6 = time.perf_counter()
import glob
p = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "fast5", "*.fast5")))[0]
rd, fq = cli.parse_read(p, "Basecall_1D_000", "BaseCalled_template")
rt = hs.read_tensors_raw(rd)
t7 = time.perf_counter()
out = rv.predict_reads_raw([rt.raw], [rt.starts], [rt.feat_ev], [rt.shift], [rt.scale])
t8 = time.perf_counter()
out = rv.predict_reads_raw([rt.raw], [rt.starts], [rt.feat_ev], [rt.shift], [rt.scale])
t9 = time.perf_counter()
rv2 = Reviser(m1, m2)
t10 = time.perf_counter()
print(f"numpy import {t1-t0:.3f}  package import {t2-t1:.3f}  weights (h5lite) {t3-t2:.3f}  dlopen {t4-t3:.3f}  "
      f"device_count (HIP init) {t5-t4:.3f}  nrv_create {t6-t5:.3f}  parse+host stage one read {t7-t6:.3f}  "
      f"first predict {t8-t7:.3f}  second predict {t9-t8:.3f}  second nrv_create {t10-t9:.3f}  total {t10-t0:.3f}")
