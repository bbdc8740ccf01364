#!/bin/bash
# Round 6: the command line's device-call size (NRV_CLI_GROUPS launch groups per call, default 16) against its end-to-end rate:
# 4000 reads (the five fixture reads x 800, symlinked on tmpfs), human weights, one GPU; each setting twice, alternating.
D=/dev/shm/nrv_clig_in; O=/dev/shm/nrv_clig_out/
rm -rf $D $O; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5 tests/golden/fast5_more/*.fast5; do
  for k in $(seq 1 ${REP:-800}); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
for rep in 1 2; do
  for g in ${GROUPS_LIST:-16 32 64 8}; do
    rm -rf $O
    echo -n "NRV_CLI_GROUPS=$g: "
    NRV_CLI_GROUPS=$g python3 NanoReviser.py -d $D -o $O -S human --thread 16 2>&1 | grep -E "bases/s end to end|Error" | tail -1
  done
done
rm -rf $D $O
