#!/bin/bash
# Development builds of the engine into nanoreviser_amd/csrc/exp/ (git-ignored; they travel to the GPU box and are
# loaded with NRV_LIB=... / Reviser(lib_path=...); scripts/gpu_variants.py times several in one process).
# Always the product's kernels and flags plus:
#   tools/lstm_exp.sh stamp                      -> exp/libnanorev_hip_stamp.so   -DNRV_STAMP=1: s_memtime stamps at the phase
#                                                   edges of lstm_h2s_kernel (scripts/gpu_stamps.py) and lstm_h2w_kernel
#                                                   (scripts/gpu_stamps_w.py)
#   tools/lstm_exp.sh D:name:-DNRV_X=1,-DNRV_Y=2 -> exp/libnanorev_hip_name.so    any -D flags (round 6 removed the experiment switches
#                                                   from the sources: a knock-out is a temporary #if block now, built this way,
#                                                   timed by scripts/gpu_variants.py and never committed - HISTORY.md r06)
# FAST=1 adds -DNRV_DEV_FAST (f16x2 mode with hard_sigmoid only: half the compile time; never the product).
cd "$(dirname "$0")/../nanoreviser_amd/csrc" || exit 1
mkdir -p exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -fno-slp-vectorize -mllvm -enable-post-misched=0 -mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000"
[ -n "$FAST" ] && FLAGS="$FLAGS -DNRV_DEV_FAST=1"
for v in "$@"; do
  if [ "$v" = stamp ]; then
    ( /opt/rocm/bin/hipcc $FLAGS -DNRV_STAMP=1 -o exp/libnanorev_hip_stamp.so nrv_api.hip > exp/build_stamp.log 2>&1; echo "stamp rc=$?" ) &
  elif [[ "$v" == D:* ]]; then
    IFS=: read -r _ name defs <<< "$v"
    ( /opt/rocm/bin/hipcc $FLAGS ${defs//,/ } -o exp/libnanorev_hip_$name.so nrv_api.hip > exp/build_$name.log 2>&1; echo "$name rc=$?" ) &
  else
    echo "usage: tools/lstm_exp.sh stamp | D:name:-Dflag[,-Dflag...]" >&2
  fi
done
wait
