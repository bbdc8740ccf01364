#!/bin/bash
# Kernel trace + PMC passes of the bench workload in one precision mode.  usage: gpu_prof.sh TAG PRECISION
TAG=${1:-r02b}; PREC=${2:-f16x2}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python3 bench.py --precision $PREC --no-cpu-baseline --no-extras > $O/bench_$PREC.log 2>&1
rm -rf $O/prof_kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py --precision $PREC --steps 50 --warmup 10 --prime 64 --no-cpu-baseline --no-extras > $O/prof_kt.log 2>&1
PMC_OUT=$O/pmc BENCH_EXTRA="--precision $PREC" bash scripts/gpu_pmc.sh > $O/pmc.log 2>&1
python3 tools/parse_pmc.py $O/pmc $O/pmc_summary.json > /dev/null 2>&1
find $O/prof_kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof_kt $O/pmc/pass*/
tail -1 $O/bench_$PREC.log | cut -c1-600; head -12 $O/kernel_stats.csv
