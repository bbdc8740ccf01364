#!/bin/bash
# Shader clock of the 192->128 layer inside experiment builds made with bit 64 (device printf of
# clock64 / wall_clock64 deltas of one workgroup).  usage: scripts/gpu_clk.sh 64 80 ...
mkdir -p gpurun_out/exp
for v in "$@"; do
  export NRV_LIB=$PWD/nanoreviser_amd/csrc/exp/libnanorev_hip_exp$v.so
  timeout 300 python3 scripts/gpu_exp_time.py 2> gpurun_out/exp/clk$v.err > gpurun_out/exp/clk$v.out
  python3 - "$v" <<'P'
import sys,json
v=sys.argv[1]
c=[];w=[];js=None
for ln in open(f"gpurun_out/exp/clk{v}.out"):
    if ln.startswith("CLK"):
        a,b=ln.split()[1:3]; c.append(int(a)); w.append(int(b))
    elif ln.startswith("{"): js=json.loads(ln)
n=len(c)//2
import statistics as st
cc=st.median(c[n:]); ww=st.median(w[n:])
print("exp",v,"lstm3 cycles",cc,"wall_us",ww/100.0,"GHz",round(cc/ww/10,3), js["kernel_us"] if js else None)
P
done
