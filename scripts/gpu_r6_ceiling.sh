#!/bin/bash
# Round 6, VERDICT r05 weak #4: the instrumented clock_vs_fill (every wave's start / end realtime, CU ids) under sustained load,
# then one bench line on the same box as the round's baseline.
# usage: gpu_r6_ceiling.sh TAG [SUSTAIN] [FULL 0|1] [bench|nobench]
TAG=${1:-r06a}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/clock_vs_fill tools/microbench/clock_vs_fill.hip 2>/dev/null || { echo "build failed"; exit 1; }
timeout 600 /tmp/clock_vs_fill ${2:-2000} $O/clock_vs_fill.tsv ${3:-0} > $O/clock_vs_fill.txt 2>&1; echo "clock_vs_fill rc=$?"
if [ "${4:-bench}" = bench ]; then timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; fi
grep -c wall $O/clock_vs_fill.txt
[ -f $O/bench_driver.json ] && tail -1 $O/bench_driver.json | cut -c1-300
