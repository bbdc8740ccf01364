#!/bin/bash
# Kernel trace + PMC passes of the bench workload in one precision mode.  usage: gpu_prof.sh TAG PRECISION
# The kernel-trace summary covers the TIMED REGION only (tools/kernel_trace_stats.py), so that its averages
# are the same quantity as roofline.avg_launch_us of the JSON line printed by the very same command.
TAG=${1:-r03}; PREC=${2:-f16x2}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
PRIME=300; WARM=10; STEPS=50
rm -rf $O/prof_kt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py --precision $PREC --steps $STEPS --warmup $WARM --prime $PRIME --no-cpu-baseline --no-extras > $O/prof_kt_$PREC.log 2>&1
python3 tools/kernel_trace_stats.py $O/prof_kt $O/kernel_stats_timed_region_$PREC.csv --prime $PRIME --warmup $WARM --steps $STEPS --settle-from $O/prof_kt_$PREC.log > $O/kernel_stats_timed_region_$PREC.txt 2>&1
find $O/prof_kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_all_dispatches_$PREC.csv \;
PMC_OUT=$O/pmc_$PREC BENCH_EXTRA="--precision $PREC" bash scripts/gpu_pmc.sh > $O/pmc_$PREC.log 2>&1
python3 tools/parse_pmc.py $O/pmc_$PREC $O/pmc_summary_$PREC.json > /dev/null 2>&1
rm -rf $O/prof_kt $O/pmc_$PREC/pass*/
grep "^{\"metric" $O/prof_kt_$PREC.log | tail -1 | cut -c1-900; cat $O/kernel_stats_timed_region_$PREC.txt
