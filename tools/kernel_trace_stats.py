#!/usr/bin/env python3
"""Per-kernel duration statistics of a rocprofv3 --kernel-trace CSV over the TIMED REGION of bench.py only.

`rocprofv3 --kernel-trace --stats` averages every dispatch of the process, including the ~300 priming steps
whose first launches run on a cold clock (round 2: 165.0 us for lstm3 in the summary against 155.6 us from the
hipEvent pairs inside the timed region).  bench.py's dispatch sequence is fixed - 1 parity-guard step, P priming
steps, S settling steps (blocks of 200 until the step time is stable; the count is in the run's JSON line), W warm-up
steps, K timed steps, then an untimed all-kernel pass - so the timed region is the dispatches
[1 + P + S + W, 1 + P + S + W + K) of every kernel.  This tool keeps exactly those and writes a CSV in the layout of
rocprofv3's *_kernel_stats.csv, so that its average must agree with the JSON line's roofline.avg_launch_us.

  python3 tools/kernel_trace_stats.py <dir with *kernel_trace.csv> <out.csv> --prime 300 --warmup 10 --steps 50
"""
import argparse
import csv
import glob
import os
import statistics
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--prime", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--guard", type=int, default=1, help="steps before the priming (bench.py's parity guard)")
    ap.add_argument("--settle-from", default=None, help="log holding the run's JSON line: its config.settle_ms_per_step "
                    "blocks (200 untimed steps each, behind the priming) are skipped too")
    a = ap.parse_args()
    settle = 0
    if a.settle_from:
        import json
        lines = [ln for ln in open(a.settle_from) if ln.startswith('{"metric')]
        if lines:
            settle = 200 * len(json.loads(lines[-1]).get("config", {}).get("settle_ms_per_step", []))
    files = glob.glob(os.path.join(a.src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {a.src}")
    per = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    lo = a.guard + a.prime + settle + a.warmup
    rows = []
    for k, v in per.items():
        v.sort()
        # kernels launched once per step (the engine's six; segment_kernel etc. are not part of the bench step)
        if len(v) < lo + a.steps:
            continue
        d = [(e - s) for s, e in v[lo:lo + a.steps]]
        rows.append((k, len(d), sum(d), statistics.mean(d), min(d), max(d), statistics.pstdev(d)))
    tot = sum(r[2] for r in rows) or 1
    rows.sort(key=lambda r: -r[2])
    with open(a.dst, "w", newline="") as fp:
        w = csv.writer(fp)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for k, n, t, m, mn, mx, sd in rows:
            w.writerow([k, n, t, f"{m:.3f}", f"{100.0 * t / tot:.2f}", mn, mx, f"{sd:.3f}"])
    for k, n, t, m, mn, mx, sd in rows:
        print(f"{m / 1e3:9.2f} us  x{n:4d}  min {mn / 1e3:8.2f} max {mx / 1e3:8.2f}  {k[:110]}")


if __name__ == "__main__":
    main()
