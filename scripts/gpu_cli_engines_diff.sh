#!/bin/bash
# Do N engines per device (NRV_CLI_ENGINES) write the same files as one?  usage: REP=500 gpu_cli_engines_diff.sh
D=/tmp/nrv_e2e_in; A=/tmp/nrv_out_a/; B=/tmp/nrv_out_b/
rm -rf $D $A $B; mkdir -p $D
i=0
for f in tests/golden/fast5/*.fast5; do
  for k in $(seq 1 ${REP:-500}); do ln -s $(realpath $f) $D/r${i}_$k.fast5; done; i=$((i+1))
done
NRV_CLI_ENGINES=1 python3 NanoReviser.py -d $D -o $A -S ecoli --thread 16 2>&1 | grep -E "s:::.*bases/s"
for env in "NRV_CLI_ENGINES=2" "NRV_CLI_ENGINES=2 NRV_HOST_REGISTER=0" "NRV_CLI_ENGINES=2 NRV_LANES=0" "NRV_CLI_ENGINES=2 NRV_PRECISION=f32" "NRV_CLI_ENGINES=1"; do
  rm -rf $B
  env $env python3 NanoReviser.py -d $D -o $B -S ecoli --thread 16 2>&1 | grep -E "s:::.*bases/s" | cut -c1-80
  nd=0; for f in $(ls $A); do cmp -s $A/$f $B/$f || nd=$((nd+1)); done
  echo "$env: $nd of $(ls $A | wc -l) files differ"
done
