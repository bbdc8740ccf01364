#!/usr/bin/env python3
"""Window-by-window precision report of the HIP path on every window of the five fixture reads, both
species, every precision mode: max |dp| vs the fp64 arbiter and vs both f32 restatements of the
oracle (NumPy-f32, C port), how many windows exceed 1e-4 and whether those are the windows on which
the f32 restatements THEMSELVES leave the bar (i.e. the fp32 noise floor of an ill-conditioned
window, not an engine error), and the argmax differences with the arbiter's margin.
Needs tests/golden/_local/whole_reads_ref.npz (tools/make_whole_read_refs.py).  Writes JSON to argv[1]."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nanoreviser_amd import hoststage as hs          # noqa: E402
from nanoreviser_amd.engine import Reviser, PRECISIONS           # noqa: E402
from nanoreviser_amd.weights import load_species     # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
BAR = 1e-4


def main():
    ref = np.load(os.path.join(G, "_local", "whole_reads_ref.npz"))
    index = json.load(open(os.path.join(G, "reads", "index.json")))
    wins = {}
    for ent in index:
        key = ent["key"]
        g = np.load(os.path.join(G, "reads", key + ".npz"))
        rd = hs.collapse_events(g["ev_start"], g["ev_mean"], g["ev_stdv"], g["ev_model_state"], g["ev_move"],
                                g["raw_signal"])
        rt = hs.read_tensors(rd)
        sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
        wins[key] = (np.ascontiguousarray(sw), np.ascontiguousarray(fw))
    report = {}
    for sp in ("ecoli", "human"):
        m1, m2 = load_species(sp)
        for mode in sorted(PRECISIONS):
            rv = Reviser(m1, m2, precision=mode)
            acc = {"windows": 0}
            for mi in (1, 2):
                acc[f"m{mi}"] = {"max_vs_fp64": 0.0, "max_vs_np32": 0.0, "max_vs_c32": 0.0, "over_bar_vs_fp64": 0,
                                 "over_bar_where_f32_oracles_also_over_half_bar": 0, "max_on_well_conditioned": 0.0,
                                 "f32_oracle_max_vs_fp64": 0.0, "f32_oracle_over_bar": 0, "argmax_diff_vs_fp64": 0,
                                 "argmax_diff_margins": [], "rms_vs_fp64": 0.0, "rms_np32_vs_fp64": 0.0, "worst": []}
            for key, (sw, fw) in wins.items():
                out = rv.predict_pair(sw, fw)
                acc["windows"] += len(fw)
                for mi in (1, 2):
                    p, a = out[mi - 1], out[mi + 1]
                    p64, q, c = (ref[f"{key}/{sp}/{n}_{mi}"] for n in ("p64", "np32", "c32"))
                    e = np.abs(p - p64).max(-1)
                    nf = np.maximum(np.abs(q - p64).max(-1), np.abs(c - p64).max(-1))
                    r = acc[f"m{mi}"]
                    r["max_vs_fp64"] = max(r["max_vs_fp64"], float(e.max()))
                    r["max_vs_np32"] = max(r["max_vs_np32"], float(np.abs(p - q).max()))
                    r["max_vs_c32"] = max(r["max_vs_c32"], float(np.abs(p - c).max()))
                    r["over_bar_vs_fp64"] += int((e > BAR).sum())
                    r["over_bar_where_f32_oracles_also_over_half_bar"] += int(((e > BAR) & (nf > BAR / 2)).sum())
                    well = nf <= BAR / 2
                    r["max_on_well_conditioned"] = max(r["max_on_well_conditioned"], float(e[well].max()))
                    r["f32_oracle_max_vs_fp64"] = max(r["f32_oracle_max_vs_fp64"], float(nf.max()))
                    r["f32_oracle_over_bar"] += int((nf > BAR).sum())
                    r["rms_vs_fp64"] += float((np.abs(p - p64) ** 2).sum())
                    r["rms_np32_vs_fp64"] += float((np.abs(q - p64) ** 2).sum())
                    srt = np.sort(p64, -1)
                    for i in np.nonzero(a != p64.argmax(-1))[0]:
                        r["argmax_diff_vs_fp64"] += 1
                        r["argmax_diff_margins"].append([key, int(i), float(srt[i, -1] - srt[i, -2]), float(nf[i])])
                    for i in np.argsort(-e)[:3]:
                        r["worst"].append([key, int(i), float(e[i]), float(nf[i])])
            for mi in (1, 2):
                r = acc[f"m{mi}"]
                cls = 6 if mi == 1 else 5
                r["rms_vs_fp64"] = (r["rms_vs_fp64"] / (acc["windows"] * cls)) ** 0.5
                r["rms_np32_vs_fp64"] = (r["rms_np32_vs_fp64"] / (acc["windows"] * cls)) ** 0.5
                r["worst"] = sorted(r["worst"], key=lambda t: -t[2])[:5]
            report[f"{sp}/{mode}"] = acc
            rv.close()
            print(sp, mode, {k: (v["max_vs_fp64"], v["over_bar_vs_fp64"], v["f32_oracle_max_vs_fp64"], v["argmax_diff_vs_fp64"])
                             for k, v in acc.items() if k != "windows"}, flush=True)
    json.dump(report, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)


if __name__ == "__main__":
    main()
