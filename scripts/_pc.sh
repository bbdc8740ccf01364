export TMPDIR=/tmp
for set in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA"; do
  rm -rf gpurun_out/pp
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pp -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
cc = glob.glob("gpurun_out/pp/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0.0,0])
for r in csv.DictReader(open(cc)):
    if "cnn_kernel" not in r["Kernel_Name"]: continue
    a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({c: round(v[0]/v[1]/2048) for c, v in agg.items()})
PY
done
