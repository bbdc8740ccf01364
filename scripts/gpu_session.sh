#!/bin/bash
# One GPU-box visit of round 2: probes, gpu tests, precision report, bench (driver's flags and default).
# Outputs under gpurun_out/<TAG>/.
TAG=${1:-r02a}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
(cd tools/microbench && timeout 60 ./f16_mfma_probe) > $O/f16_probe.log 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 600 python3 scripts/gpu_precision_report.py $O/precision_report.json > $O/precision.log 2>&1; echo "rc=$?" >> $O/precision.log
for i in 1 2 3; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_driver_$i.log 2>&1
done
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --prime 64 --no-extras --no-cpu-baseline > $O/bench_driver_prime64.log 2>&1
timeout 900 python3 bench.py > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log
timeout 300 python3 scripts/gpu_hostapi.py > $O/hostapi.log 2>&1
NRV_HOST_REGISTER=0 timeout 300 python3 scripts/gpu_hostapi.py > $O/hostapi_noreg.log 2>&1
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; cat $O/f16_probe.log; tail -5 $O/precision.log | cut -c1-400
for f in $O/bench_driver_*.log; do tail -1 $f | cut -c1-260; done
tail -2 $O/bench.log | cut -c1-1500; cat $O/hostapi.log $O/hostapi_noreg.log
