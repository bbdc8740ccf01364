#!/bin/bash
# HISTORICAL (r05): -DNRV_LSTM_MAP was removed from the sources in round 6; the result is in profiles/r05_energy_map.json.
# r05 energy experiment (VERDICT r04 #4a): forward and backward workgroup of a row block on the SAME XCD (-DNRV_LSTM_MAP=1) against
# the default placement, on lstm_h2w_kernel (192->128) and lstm_h2s_kernel (256->64): microseconds (same process), socket power and
# shader clock (one launch looped for 4 s), FETCH_SIZE / WRITE_SIZE / TCC hits and misses (rocprofv3 --pmc, separate passes).
# usage: gpu_energy_map.sh TAG     (needs exp/libnanorev_hip_{base,map1,stamp0,stampmap}.so: tools/lstm_exp.sh)
O=gpurun_out/${1:-r05_energy}; mkdir -p $O
E=nanoreviser_amd/csrc/exp
export TMPDIR=/tmp
VAR_REPS=2 timeout 600 python3 scripts/gpu_variants.py $E/libnanorev_hip_base.so $E/libnanorev_hip_map1.so > $O/variants.log 2>&1
for v in stamp0 stampmap; do
  POWER_STAGES=3,4 timeout 300 python3 scripts/gpu_power_stage.py $E/libnanorev_hip_$v.so > $O/power_$v.json 2> $O/power_$v.err
done
ARGS="python3 bench.py --steps 8 --warmup 3 --prime 20 --no-cpu-baseline --no-extras --no-prof"
for v in base map1; do
  export NRV_LIB=$PWD/$E/libnanorev_hip_$v.so
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rm -rf $O/pmc_$v/pass$i
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$v/pass$i -- $ARGS > $O/pmc_${v}_pass$i.log 2>&1
    echo "$v pass $i ($set) rc=$?"
  done
  unset NRV_LIB
  python3 tools/parse_pmc.py $O/pmc_$v $O/pmc_summary_$v.json > /dev/null 2>&1
  rm -rf $O/pmc_$v
done
cat $O/variants.log | grep -v amdgpu.ids | cut -c1-260
for v in stamp0 stampmap; do echo "== $v"; cat $O/power_$v.json | head -40; done
python3 - <<PY
import json
for v in ("base", "map1"):
    try:
        j = json.load(open("$O/pmc_summary_%s.json" % v))
        for k in ("lstm3", "lstm4"):
            r = j[k]
            print(v, k, {x: r.get(x) for x in ("hbm_bytes_per_launch", "hbm_read_bytes_corrected", "hbm_write_bytes", "TCC_HIT_sum", "TCC_MISS_sum", "l2_hit_rate")})
    except Exception as e:
        print(v, "pmc summary:", e)
PY
