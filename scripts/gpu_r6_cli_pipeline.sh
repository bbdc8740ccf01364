#!/bin/bash
# Round 6: the command line with its device calls pipelined two deep (nrv_reads_raw_begin / _end; default), one whole-call at a
# time (NRV_CLI_PIPELINE=0), and round 5's staged calls (NRV_CLI_PIPELINE=0 NRV_RAW_STAGED=1): 4000 human-model reads (the five
# fixture reads x 800, symlinked on tmpfs), one GPU, each setting twice, alternating; then the same on the two-fixture E. coli set.
D=/dev/shm/nrv_clip_in; O=/dev/shm/nrv_clip_out/
for SP in human ecoli; do
  rm -rf $D $O; mkdir -p $D
  i=0
  if [ $SP = human ]; then FILES="tests/golden/fast5/*.fast5 tests/golden/fast5_more/*.fast5"; REP=800; else FILES="tests/golden/fast5/*.fast5"; REP=2000; fi
  for f in $FILES; do
    for k in $(seq 1 $REP); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
  done
  for rep in 1 2; do
    for env in "NRV_CLI_PIPELINE=1" "NRV_CLI_PIPELINE=0" "NRV_CLI_PIPELINE=0 NRV_RAW_STAGED=1"; do
      rm -rf $O
      echo -n "$SP $env: "
      env $env python3 NanoReviser.py -d $D -o $O -S $SP --thread 16 2>&1 | grep -E "bases/s end to end|Error" | tail -1
    done
  done
done
rm -rf $D $O
