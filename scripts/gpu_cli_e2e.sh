#!/bin/bash
# End-to-end CLI throughput on ~1k reads (BASELINE configs[1] shape): the two committed fixture reads
# replicated 500x each via symlinks (fast5 parsing + host stage + device + merge + FASTA writes).
D=/tmp/nrv_e2e_in; O=/tmp/nrv_e2e_out/
rm -rf $D $O; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5; do
  for k in $(seq 1 ${REP:-500}); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
ls $D | wc -l
for th in ${THREADS:-16 4}; do
  python3 NanoReviser.py -d $D -o $O -S ecoli --thread $th --batch ${BATCH:-4096} 2>&1 | grep -E "s:::|Error" | tail -4
done
ls $O | wc -l
