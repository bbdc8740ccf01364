#!/bin/bash
# One GPU-box visit: gpu tests, smoke, bench, rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/.
TAG=${1:-r01}
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python3 bench.py > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_$TAG.log
export TMPDIR=/tmp
rm -rf gpurun_out/prof_kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/prof_kt.log 2>&1; echo "rocprof rc=$?" >> gpurun_out/prof_kt.log
bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1
tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench_$TAG.log | cut -c1-900
