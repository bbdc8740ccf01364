#!/bin/bash
# LDS counters of one library build: scripts/gpu_pmc_lds.sh <lib.so> <outdir>
export TMPDIR=/tmp
LIB=$1; O=${2:-gpurun_out/pmc_lds}
mkdir -p $O
i=0
for set in "SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1)); rm -rf $O/pass$i
  VAR_REPS=1 timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/pass$i -- python3 scripts/gpu_variants.py $LIB > $O/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,c in acc.items():
    print(k, {n: round(sum(v[-20:])/len(v[-20:])) for n,v in sorted(c.items())})
PY
rm -rf $O/pass*/
