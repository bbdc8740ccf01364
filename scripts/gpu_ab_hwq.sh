#!/bin/bash
# A/B of GPU_MAX_HW_QUEUES (HIP's hardware queues per process) on the small-group configuration (C2: batch 512 on stream lanes).
for q in default 8 16; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-e2e 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); h=d['host_inclusive']
print('GPU_MAX_HW_QUEUES=$q', 'step', round(d['ms_per_step'],4), {k:round(v['bases_per_s']/1e6,2) for k,v in h.items() if 'bases_per_s' in v},
      {k:(round(v['bases_per_s_device_resident']/1e6,2), round(v['bases_per_s_host_inclusive_raw_reads']/1e6,2)) for k,v in d['configs'].items()},
      'read_mode', round(d['read_mode']['bases_per_s']/1e6,2))"
done
