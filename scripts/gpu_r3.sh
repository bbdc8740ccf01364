#!/bin/bash
# One GPU-box visit of round 3.  usage: gpu_r3.sh TAG [tests|bench|all]   Outputs under gpurun_out/<TAG>/.
TAG=${1:-r03a}; WHAT=${2:-all}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then
  timeout 2700 python3 -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  for i in 1 2; do
    timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_driver_$i.log 2>&1
  done
  timeout 900 python3 bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
fi
grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
tail -2 $O/smoke.log
for f in $O/bench_driver_*.log; do tail -1 $f | cut -c1-300; done
tail -2 $O/bench.log | cut -c1-3000
