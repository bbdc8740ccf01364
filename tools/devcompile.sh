#!/bin/bash
# Device-only compile of the engine with register / spill remarks for kernels matching $1 (regex).
# Leaves the disassembly of the device code object in /tmp/study/dev.s.   usage: tools/devcompile.sh lstm_h2o
mkdir -p /tmp/study
cd /root/repo/nanoreviser_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c nrv_api.hip \
  -Rpass-analysis=kernel-resource-usage -fno-slp-vectorize -mllvm -enable-post-misched=0 -mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000 \
  -o /tmp/study/dev.o 2> /tmp/study/res.txt
grep -E "error" /tmp/study/res.txt | head
grep -A11 "Function Name: .*${1:-lstm}" /tmp/study/res.txt | grep "Name\|VGPRs:\|AGPRs\|Spill\|ScratchSize\|LDS" | sed 's/.*remark: //;s/\[-R.*//'
cd /tmp/study && /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=dev.co 2>/dev/null && /opt/rocm/lib/llvm/bin/llvm-objdump -d dev.co > dev.s
