#!/bin/bash
# Round 6: the GPU suite + smoke + one driver-style bench line (after a source change), optionally the clock_vs_fill microbench.
# usage: gpu_r6_suite.sh TAG [micro]
TAG=${1:-r06s}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
if [ "${2:-}" = micro ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/clock_vs_fill tools/microbench/clock_vs_fill.hip 2>/dev/null && \
  timeout 600 /tmp/clock_vs_fill 2000 $O/clock_vs_fill.tsv 1 > $O/clock_vs_fill.txt 2>&1; echo "clock_vs_fill rc=$?"
fi
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -1 $O/bench_driver.json | cut -c1-400
