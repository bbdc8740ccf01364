#!/bin/bash
# PMC passes for the split-bf16 LSTM kernels (NRV_SPLIT mask from $1, default 14).
export TMPDIR=/tmp
export NRV_SPLIT=${1:-14}
mkdir -p gpurun_out/pmc
ARGS="python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-prof"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc/pass$i
  timeout 600 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc/pass$i -- $ARGS > gpurun_out/pmc/pass$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python3 tools/parse_pmc.py gpurun_out/pmc gpurun_out/pmc/summary.json
