D=/dev/shm/nrv_clig_in; O=/dev/shm/nrv_clig_out/
rm -rf $D $O; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5 tests/golden/fast5_more/*.fast5; do
  for k in $(seq 1 400); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
for g in 8 16 32; do
  rm -rf $O
  NRV_HOST_TRACE=1 NRV_CLI_GROUPS=$g python3 NanoReviser.py -d $D -o $O -S human --thread 16 > /tmp/o_$g.txt 2> /tmp/e_$g.txt
  grep "bases/s end to end" /tmp/o_$g.txt
  grep "host trace" /tmp/e_$g.txt | awk -v g=$g '{w+=$4; p+=$(NF-5); r+=$9; n++} END {printf "GROUPS=%d: %d calls, %.0f windows per call, pipeline %.3f ms per call = %.1f ns per window, register %.3f ms per call\n", g, n, w/n, p/n, 1e6*p/w, r/n}'
  grep "host trace" /tmp/e_$g.txt | sed -n 20,23p
done
rm -rf $D $O
