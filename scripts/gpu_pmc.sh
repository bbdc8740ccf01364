#!/bin/bash
# PMC passes (each in its own run, no tracing flags), per MI355X_MICROARCH.md "rocprofv3 PMC slots".
export TMPDIR=/tmp
mkdir -p ${PMC_OUT:-gpurun_out/pmc}
rocprofv3 -L > ${PMC_OUT:-gpurun_out/pmc}/counters_list.txt 2>&1
ARGS="python3 bench.py --steps 8 --warmup 3 --prime 20 --no-cpu-baseline --no-extras --no-prof ${BENCH_EXTRA}"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf ${PMC_OUT:-gpurun_out/pmc}/pass$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d ${PMC_OUT:-gpurun_out/pmc}/pass$i -- $ARGS > ${PMC_OUT:-gpurun_out/pmc}/pass$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
find ${PMC_OUT:-gpurun_out/pmc} -name "*counter_collection.csv" | head -20
