#!/bin/bash
# Round 6, VERDICT r05 next #6 measured instead of argued: what the three small launches cost per 4096 windows when a launch
# covers 8192 or 16384 windows (their weights staged once per workgroup for twice / four times the rows): bench.py's per-kernel
# brackets at --batch 4096 / 8192 / 16384, same box, same process order twice.
TAG=${1:-r06g}; O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
  for B in 4096 8192 16384; do
    timeout 600 python3 bench.py --batch $B --steps 40 --warmup 5 --prime 200 --no-extras --no-cpu-baseline --prof-all --blocks 1 2>/dev/null | tail -1 > $O/bench_b${B}_$rep.json
    python3 - $O/bench_b${B}_$rep.json $B <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read()); B = int(sys.argv[2]); k = j["kernel_us"]
per = {n.split()[0]: v * 4096 / B for n, v in k.items()}
print(f"batch {B:6d}: step {j['ms_per_step'] * 4096 / B:.4f} ms per 4096 windows | " + " ".join(f"{n}:{v:6.1f}" for n, v in per.items()) +
      f" | small three {per['cnn_kernel'] + per['lstm2'] + per['head_mlp+head_final']:.1f} us per 4096 windows")
PY
  done
done
