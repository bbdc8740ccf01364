#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE) and duration of each kernel for a list of library builds.
export TMPDIR=/tmp
O=gpurun_out/fetch; mkdir -p $O
for LIB in "$@"; do
  n=$(basename $LIB .so)
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/p
    VAR_REPS=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p -- python3 scripts/gpu_variants.py $LIB > $O/$n.$set.log 2>&1
    python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:48]].append(float(r["Counter_Value"]))
print("$n $set", {k: round(sum(v[-20:])/len(v[-20:])*1024/1e6*(2 if "$set"=="FETCH_SIZE" else 1),1) for k,v in acc.items() if "nrv" in k})
PY
  done
  grep step $O/$n.WRITE_SIZE.log | cut -c1-200
done
rm -rf $O/p
