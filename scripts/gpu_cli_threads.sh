#!/bin/bash
# CLI end to end (REP x the two fixture reads) at several parser-worker counts, native host stage and the Python one.
D=/tmp/nrv_e2e_in; O=/tmp/nrv_e2e_out/
rm -rf $D; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5; do
  for k in $(seq 1 ${REP:-2000}); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
for cfg in "2" "4" "8" "16" "16 NRV_HOST_THREADS=0" "16 NRV_HOST_LIB=0"; do
  set -- $cfg
  rm -rf $O
  t0=$(date +%s%N)
  env $2 NRV_CLI_TRACE=1 python3 NanoReviser.py -d $D -o $O -S ecoli --thread $1 2>&1 | grep -E "s:::.*bases/s|process_files" | cut -c1-330 | tail -2
  t1=$(date +%s%N); echo "threads $1 $2: wall $(( (t1 - t0) / 1000000 )) ms"
done
