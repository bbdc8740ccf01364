#!/bin/bash
# Builds experiment variants of the engine (-DNRV_EXP=<bits>, see nrv_lstm_f16x2.h: parts of lstm_h2o_kernel
# compiled out; results are WRONG, only the timing means something) into nanoreviser_amd/csrc/exp/.
# usage: tools/lstm_exp.sh 1 3 7 15 ...     then on the GPU box: scripts/gpu_exp.sh 1 3 7 15
cd /root/repo/nanoreviser_amd/csrc || exit 1
mkdir -p exp
for v in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DNRV_EXP=$v \
      -mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000 \
      -o exp/libnanorev_hip_exp$v.so nrv_api.hip > exp/build$v.log 2>&1; echo "exp$v rc=$?" ) &
done
wait
