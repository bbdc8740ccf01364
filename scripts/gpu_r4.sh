#!/bin/bash
# One GPU-box visit of round 4.  usage: gpu_r4.sh TAG [steps...]   steps: range stamps variants tests bench
TAG=${1:-r04a}; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for WHAT in "$@"; do
  case $WHAT in
    range)
      timeout 900 python3 -m pytest tests/test_gpu_range.py -m gpu -q -s -x > $O/pytest_range.log 2>&1; echo "pytest rc=$?" >> $O/pytest_range.log
      grep -E "passed|failed|error|^SPIKE|^RANGE" $O/pytest_range.log | tail -30 ;;
    stamps)
      timeout 300 python3 scripts/gpu_stamps.py > $O/stamps.json 2> $O/stamps.err; echo "stamps rc=$?"; tail -3 $O/stamps.err
      cat $O/stamps.json ;;
    variants)
      VAR_REPS=${VAR_REPS:-2} timeout 600 python3 scripts/gpu_variants.py nanoreviser_amd/csrc/libnanorev_hip.so $(ls nanoreviser_amd/csrc/exp/libnanorev_hip_v_*.so 2>/dev/null) > $O/variants.log 2>&1
      cat $O/variants.log | cut -c1-400 ;;
    tests)
      timeout 2700 python3 -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
      timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
      grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20; tail -2 $O/smoke.log ;;
    bench)
      for i in 1 2; do
        timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_driver_$i.log 2>&1
      done
      timeout 900 python3 bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
      for f in $O/bench_driver_*.log; do tail -1 $f | cut -c1-300; done
      tail -2 $O/bench.log | cut -c1-3000 ;;
  esac
done
