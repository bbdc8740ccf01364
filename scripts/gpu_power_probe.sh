#!/bin/bash
# Socket power and shader clock (rocm-smi, every 0.25 s) while (1) the matrix-pipe microbenchmark and (2) the product's step
# loop run for a few seconds each.  usage: gpu_power_probe.sh TAG
O=gpurun_out/${1:-power}; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/clock_vs_fill tools/microbench/clock_vs_fill.hip 2>/dev/null
probe() {  # $1 = label; samples until the file $O/stop exists
  while [ ! -e $O/stop ]; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk" | tr '\n' ' ' | sed "s/^/$1 /"; echo
    sleep 0.25
  done
}
rm -f $O/stop; probe microbench > $O/probe_microbench.txt & 
timeout 300 /tmp/clock_vs_fill 12000 > $O/clock_vs_fill.txt 2>&1
touch $O/stop; wait; rm -f $O/stop
probe product > $O/probe_product.txt &
STAMP_STEPS=20000 timeout 300 python3 scripts/gpu_stamps_w.py nanoreviser_amd/csrc/exp/libnanorev_hip_stamp.so > $O/stamps_w.json 2> $O/err.log
touch $O/stop; wait; rm -f $O/stop
echo "== microbench"; awk '{print $0}' $O/probe_microbench.txt | sed -n '1,200p' | awk '{for(i=1;i<=NF;i++) if ($i ~ /^\(W\):/) p=$(i+1); else if ($i ~ /Mhz/) c=$i; print p, c}' | sort | uniq -c | sort -rn | head -12
echo "== product"; awk '{for(i=1;i<=NF;i++) if ($i ~ /^\(W\):/) p=$(i+1); else if ($i ~ /Mhz/) c=$i; print p, c}' $O/probe_product.txt | sort | uniq -c | sort -rn | head -12
head -5 $O/stamps_w.json; grep -E "2 waves/SIMD, random operands|HBM stream \(" $O/clock_vs_fill.txt
