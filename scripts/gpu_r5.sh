#!/bin/bash
# End-of-round GPU visit of round 5: full GPU suite, smoke, the driver's bench command twice + the default bench line, the
# kernel trace + PMC passes of the timed region (scripts/gpu_prof.sh), and the N-rank rehearsal with the fast5-fed CLI leg.
# usage: gpu_r5.sh TAG
TAG=${1:-r05z}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
for i in 1 2; do timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err; done
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
bash scripts/gpu_prof.sh $TAG f16x2 > $O/prof_f16x2.log 2>&1
timeout 900 python3 bench.py --gpus 2 --share-device --steps 20 --warmup 5 --cli-reps 500 > $O/bench_2ranks_share_device.json 2> $O/bench_2ranks.err
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log
for f in $O/bench_driver_1.json $O/bench_driver_2.json $O/bench_default.json $O/bench_2ranks_share_device.json; do tail -1 $f | cut -c1-200; done
tail -12 $O/prof_f16x2.log | cut -c1-400
