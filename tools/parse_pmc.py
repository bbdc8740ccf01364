#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc/pass*/.../*_counter_collection.csv) into one
JSON: per kernel, the mean counter value per dispatch.  FETCH_SIZE is doubled (gfx950 reports
half the bytes of wide coalesced reads, MI355X_MICROARCH.md 'HBM'); FETCH_SIZE/WRITE_SIZE are KiB.

  python3 tools/parse_pmc.py gpurun_out/pmc profiles/r01_pmc_summary.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"lstm_(?:layer|split|pair|h2o?|h2s|h2w)_kernel<(\d+), (\d+), (\d+)", name)
    if m:
        return {"0, 0, 16": "lstm1", "8, 0, 64": "lstm2", "32, 16, 128": "lstm3", "64, 0, 64": "lstm4"}.get(
            ", ".join(m.groups()), name)
    if "lstm1_kernel" in name:
        return "lstm1"
    if "lstm2_t_kernel" in name or "lstm2_u_kernel" in name:
        return "lstm2"
    if "cnn_r_kernel" in name:
        return "cnn_r_kernel"
    if "cnn_m_kernel" in name:
        return "cnn_m_kernel"
    if "head_h2_kernel" in name:
        return "head_h2_kernel"
    if "head_mlp_split_kernel" in name:
        return "head_mlp_kernel"
    if "cnn_h2_kernel" in name:
        return "cnn_h2_kernel"
    for k in ("cnn_kernel", "head_mlp_kernel", "head_final_kernel", "head_kernel"):
        if k in name:
            return k
    return None


def main(src, dst):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    full = {}
    for f in sorted(glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        # skip warm-up dispatches: keep the last 8 per kernel (bench --steps 8)
        per = defaultdict(list)
        for r in rows:
            k = short(r["Kernel_Name"])
            if k:
                full[k] = r["Kernel_Name"]
                per[(k, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for (k, c), vals in per.items():
            vals.sort()
            for _, v in vals[-8:]:
                acc[k][c][0] += v
                acc[k][c][1] += 1
    out = {}
    for k, cs in acc.items():
        out[k] = {c: s / n for c, (s, n) in cs.items()}
        o = out[k]
        o["kernel_name"] = full.get(k, "")
        if "FETCH_SIZE" in o:
            o["hbm_read_bytes_corrected"] = o["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in o:
            o["hbm_write_bytes"] = o["WRITE_SIZE"] * 1024
        if "hbm_read_bytes_corrected" in o and "hbm_write_bytes" in o:
            o["hbm_bytes_per_launch"] = o["hbm_read_bytes_corrected"] + o["hbm_write_bytes"]
        if "TCC_HIT_sum" in o:
            o["l2_hit_rate"] = o["TCC_HIT_sum"] / max(o["TCC_HIT_sum"] + o["TCC_MISS_sum"], 1)
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k in sorted(out):
        print(k, json.dumps({a: (round(b, 1) if isinstance(b, float) else b) for a, b in sorted(out[k].items())}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
