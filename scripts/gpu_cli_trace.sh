#!/bin/bash
# CLI end to end on REP x the two fixture reads with the timeline summary (NRV_CLI_TRACE).  usage: REP=2000 gpu_cli_trace.sh [extra CLI args]
B=${SCRATCH:-/dev/shm}; D=$B/nrv_e2e_in; O=$B/nrv_e2e_out/
rm -rf $D $O; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5; do
  for k in $(seq 1 ${REP:-2000}); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
export NRV_CLI_TRACE=1
for rep in ${REPS:-1 2}; do
  t0=$(date +%s%N)
  python3 NanoReviser.py -d $D -o $O -S ecoli --thread ${THREADS:-16} "$@" 2>&1 | grep -E "trace|s:::.*bases/s|Error" | tail -4
  t1=$(date +%s%N); echo "wall $(( (t1 - t0) / 1000000 )) ms"
done
rm -rf $D $O
