#!/bin/bash
# Effective engine clock per kernel: GRBM_GUI_ACTIVE (cycles) next to the kernel-trace duration.
export TMPDIR=/tmp
for sp in 0 14; do
  export NRV_SPLIT=$sp
  rm -rf gpurun_out/clk$sp
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/clk$sp -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-prof > gpurun_out/clk$sp.log 2>&1
  echo "split=$sp rc=$?"
  python3 - <<PY
import csv, glob, collections
cc = glob.glob("gpurun_out/clk$sp/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("gpurun_out/clk$sp/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0, 0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    n, d = dur[r["Dispatch_Id"]]
    a = agg[n[:60]]; a[0] += float(r["Counter_Value"]); a[1] += d; a[2] += 1
for n, (c, d, k) in agg.items():
    print(f"{n:62s} n={k:3d} dur={d/k/1e3:8.1f} us  cycles={c/k:12.0f}  cyc/ns={c/d:.3f}")
PY
done
