D=/dev/shm/nrv_cg_in; O=/dev/shm/nrv_cg_out/
rm -rf $D $O; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5 tests/golden/fast5_more/*.fast5; do
  for k in $(seq 1 3000); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
for rep in 1 2; do
  for g in 16 32 24 48; do
    rm -rf $O
    echo -n "15000 reads NRV_CLI_GROUPS=$g: "
    NRV_CLI_GROUPS=$g python3 NanoReviser.py -d $D -o $O -S human --thread 16 2>&1 | grep -E "bases/s end to end|Error" | tail -1
  done
done
rm -rf $D $O
