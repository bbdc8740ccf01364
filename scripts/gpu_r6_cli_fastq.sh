D=/dev/shm/nrv_fq_in; O=/dev/shm/nrv_fq_out/
rm -rf $D $O; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5 tests/golden/fast5_more/*.fast5; do
  for k in $(seq 1 800); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
for fmt in fasta fastq fasta fastq; do
  rm -rf $O
  echo -n "$fmt: "
  python3 NanoReviser.py -d $D -o $O -S human -F $fmt --thread 16 2>&1 | grep -E "bases/s end to end|Error" | tail -1
done
rm -rf $D $O
