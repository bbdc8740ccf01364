#!/bin/bash
# Experiments builds of the engine into nanoreviser_amd/csrc/exp/ (git-ignored; they travel to the GPU box).
# The product library (__graft_entry__.build) ships only the kernels a documented precision mode reaches;
# -DNRV_EXPERIMENTS adds the alternative kernels / geometries behind NRV_MFMA16, NRV_HT, NRV_GEO, NRV_PAIR,
# NRV_SPLIT (DESIGN.md 3 "Environment knobs"), and -DNRV_EXP=<bits> compiles parts of the f16x2 Bi-LSTM kernels
# out (nrv_lstm_f16x2.h: results are WRONG, only the timing means something).
#   tools/lstm_exp.sh knobs          -> exp/libnanorev_hip_experiments.so   (tests/test_gpu_knobs.py, NRV_LIB=...)
#   tools/lstm_exp.sh 1 3 7 15 ...   -> exp/libnanorev_hip_exp<bits>.so     (scripts/gpu_exp.sh 1 3 7 15)
cd "$(dirname "$0")/../nanoreviser_amd/csrc" || exit 1
mkdir -p exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DNRV_EXPERIMENTS=1 -fno-slp-vectorize -mllvm -enable-post-misched=0 -mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000"
for v in "$@"; do
  if [ "$v" = knobs ]; then
    ( /opt/rocm/bin/hipcc $FLAGS -o exp/libnanorev_hip_experiments.so nrv_api.hip > exp/build_knobs.log 2>&1; echo "knobs rc=$?" ) &
  elif [ "$v" = stamp ]; then
    # product kernels + s_memtime stamps at the phase edges of lstm_h2s_kernel (scripts/gpu_stamps.py)
    ( /opt/rocm/bin/hipcc ${FLAGS/-DNRV_EXPERIMENTS=1/} -DNRV_STAMP=1 $EXTRA -o exp/libnanorev_hip_stamp.so nrv_api.hip > exp/build_stamp.log 2>&1; echo "stamp rc=$?" ) &
  elif [[ "$v" == D:* ]]; then
    # product kernels with extra -D flags: tools/lstm_exp.sh D:name:-DNRV_X=1,-DNRV_Y=2 -> exp/libnanorev_hip_name.so
    IFS=: read -r _ name defs <<< "$v"
    ( /opt/rocm/bin/hipcc ${FLAGS/-DNRV_EXPERIMENTS=1/} ${defs//,/ } -o exp/libnanorev_hip_$name.so nrv_api.hip > exp/build_$name.log 2>&1; echo "$name rc=$?" ) &
  else
    ( /opt/rocm/bin/hipcc $FLAGS -DNRV_EXP=$v -o exp/libnanorev_hip_exp$v.so nrv_api.hip > exp/build$v.log 2>&1; echo "exp$v rc=$?" ) &
  fi
done
wait
