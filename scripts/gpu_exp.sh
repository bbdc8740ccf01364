#!/bin/bash
# Times the experiment builds of tools/lstm_exp.sh (GPU box).  usage: scripts/gpu_exp.sh 1 3 7 15
mkdir -p gpurun_out/exp
for v in 0 "$@"; do
  if [ "$v" = 0 ]; then unset NRV_LIB; else export NRV_LIB=$PWD/nanoreviser_amd/csrc/exp/libnanorev_hip_exp$v.so; fi
  timeout 300 python3 scripts/gpu_exp_time.py 2> gpurun_out/exp/exp$v.err | tee -a gpurun_out/exp/times.jsonl
  [ -s gpurun_out/exp/exp$v.err ] && tail -3 gpurun_out/exp/exp$v.err
done
