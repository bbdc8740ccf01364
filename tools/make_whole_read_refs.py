#!/usr/bin/env python3
"""Oracle outputs on EVERY window of the five fixture reads, both species (40 885 windows each):
NumPy fp64 (the arbiter), NumPy f32 and the C f32 port.  ~10 minutes of CPU, so the result is cached
in tests/golden/_local/whole_reads_ref.npz - git-ignored (14 MB), shipped to the GPU box with the
tree - and consumed by scripts/gpu_precision_report.py, which compares the HIP path with all three
window by window.  Run:  python3 tools/make_whole_read_refs.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanoreviser_amd import hoststage as hs          # noqa: E402
from nanoreviser_amd.weights import load_species     # noqa: E402
from oracle import c_oracle as CO                    # noqa: E402
from oracle import nrv_oracle as O                   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(G, "_local", "whole_reads_ref.npz")


def main():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    out = {}
    index = json.load(open(os.path.join(G, "reads", "index.json")))
    models = {sp: load_species(sp) for sp in ("ecoli", "human")}
    for ent in index:
        key = ent["key"]
        g = np.load(os.path.join(G, "reads", key + ".npz"))
        rd = hs.collapse_events(g["ev_start"], g["ev_mean"], g["ev_stdv"], g["ev_model_state"], g["ev_move"],
                                g["raw_signal"])
        rt = hs.read_tensors(rd)
        sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
        sw, fw = np.ascontiguousarray(sw), np.ascontiguousarray(fw)
        for sp, (m1, m2) in models.items():
            p1, p2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float64)
            q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float32)
            c1, _ = CO.predict(m1.flat(), 11, 6, sw, fw, threads=8)
            c2, _ = CO.predict(m2.flat(), 11, 5, sw, fw, threads=8)
            for nm, v in (("p64_1", p1), ("p64_2", p2), ("np32_1", q1), ("np32_2", q2), ("c32_1", c1), ("c32_2", c2)):
                out[f"{key}/{sp}/{nm}"] = v
            print(key, sp, len(fw), flush=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
