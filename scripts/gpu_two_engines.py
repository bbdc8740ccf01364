#!/usr/bin/env python3
"""Two engines on one device driven by two host threads: are the results those of one engine alone?
(the command line's NRV_CLI_ENGINES path).  usage: python3 scripts/gpu_two_engines.py [n_threads]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanoreviser_amd.engine import Reviser  # noqa: E402
from nanoreviser_amd.weights import load_species  # noqa: E402

T = 11
m1, m2 = load_species("ecoli")
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(5)


def make_bundle(seed, n_reads=10, n_ev=6785):
    r = np.random.default_rng(seed)
    raws, starts, feats, shifts, scales = [], [], [], [], []
    for _ in range(n_reads):
        lens = r.integers(5, 40, n_ev)
        st = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
        raws.append(r.integers(300, 700, int(lens.sum()) + 64).astype(np.int16))
        starts.append(st)
        f = r.random((n_ev, 6)).astype(np.float32)
        feats.append(f)
        shifts.append(500.0)
        scales.append(60.0)
    return raws, starts, feats, shifts, scales


bundles = [make_bundle(s) for s in range(6)]
ref_rv = Reviser(m1, m2, device=0, batch=4096)
ref = [ref_rv.predict_reads_raw(*b) for b in bundles]
ref2 = [ref_rv.predict_reads_raw(*b) for b in bundles]
print("one engine, repeated: identical", all(all(np.array_equal(x, y) for x, y in zip(a, b)) for a, b in zip(ref, ref2)))
engines = [ref_rv] + [Reviser(m1, m2, device=0, batch=4096) for _ in range(nt - 1)]
bad = []


def work(rv, k):
    for rep in range(20):
        for i, b in list(enumerate(bundles))[k::nt]:          # DIFFERENT data in flight on the two engines
            out = rv.predict_reads_raw(*b)
            if not all(np.array_equal(x, y) for x, y in zip(out, ref[i])):
                d = max(float(np.abs(x.astype(np.float64) - y.astype(np.float64)).max()) for x, y in zip(out[:2], ref[i][:2]))
                bad.append((k, rep, i, d, int((out[2] != ref[i][2]).sum())))


th = [threading.Thread(target=work, args=(rv, k)) for k, rv in enumerate(engines)]
[t.start() for t in th]
[t.join() for t in th]
print(f"{nt} engines / threads: {len(bad)} of {nt * 20 * len(bundles)} calls differ from the single-engine result (each thread its own bundles)", bad[:6])
