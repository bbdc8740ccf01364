#!/usr/bin/env python3
"""Model-stage goldens: outputs of the ORACLE (NumPy fp64 restatement), not of the reference.

The reference's Keras/TF path cannot run here, so these vectors pin
(a) the oracle against its own regressions and (b) the HIP path against a fixed fp64 result on
the GPU box, where only /root/repo exists.  They do NOT pin the oracle to Keras; that link is
the property tests of tests/test_oracle_properties.py.

Writes tests/golden/model_goldens.npz:
  per fixture read r and species s: windows idx[r] (first 256 + 128 from the middle), T=11:
      p1/p2 (fp64), a1/a2
  synthetic windows (inputs stored, numpy Generator streams are not version-stable):
      T=11 (256 windows, shipped weights) and T=13 (128 windows, shipped weights + seeded
      synthetic `feature` kernel), ecoli + human.
Run:  python3 tools/make_model_goldens.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanoreviser_amd import hoststage as hs          # noqa: E402
from nanoreviser_amd.weights import load_species     # noqa: E402
from oracle import nrv_oracle as O                   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def main():
    out = {}
    index = json.load(open(os.path.join(G, "reads", "index.json")))
    models = {sp: load_species(sp) for sp in ("ecoli", "human")}
    for ent in index:
        key = ent["key"]
        g = np.load(os.path.join(G, "reads", key + ".npz"))
        rd = hs.collapse_events(g["ev_start"], g["ev_mean"], g["ev_stdv"], g["ev_model_state"],
                                g["ev_move"], g["raw_signal"])
        rt = hs.read_tensors(rd)
        T = 11
        sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, T)
        n = sw.shape[0]
        idx = np.concatenate([np.arange(256), np.arange(n // 2, n // 2 + 128)])
        out[f"{key}/idx"] = idx
        for sp, (m1, m2) in models.items():
            p1, p2, a1, a2 = O.predict_pair(m1.tensors, m2.tensors, sw[idx], fw[idx], np.float64)
            out[f"{key}/{sp}/p1"], out[f"{key}/{sp}/p2"] = p1, p2
            out[f"{key}/{sp}/a1"], out[f"{key}/{sp}/a2"] = a1, a2
            print(key, sp, "done", flush=True)
    for T, n in ((11, 256), (13, 128)):
        sig, rd_ = O.synth_windows(n, T, seed=20260 + T)
        out[f"synth{T}/signal"], out[f"synth{T}/read"] = sig, rd_
        for sp, (m1, m2) in models.items():
            a, b = m1.with_window(T), m2.with_window(T)
            p1, p2, a1, a2 = O.predict_pair(a.tensors, b.tensors, sig, rd_, np.float64)
            out[f"synth{T}/{sp}/p1"], out[f"synth{T}/{sp}/p2"] = p1, p2
            out[f"synth{T}/{sp}/a1"], out[f"synth{T}/{sp}/a2"] = a1, a2
    np.savez_compressed(os.path.join(G, "model_goldens.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(G, "model_goldens.npz")), "bytes")


if __name__ == "__main__":
    main()
