#!/bin/bash
# A/B of NRV_READ_STAGE (launch groups per pipeline stage of the read-mode host entry points) on the bench's host-inclusive / multi-group blocks.
for rep in 1 2; do for v in 1 2 4; do
  NRV_READ_STAGE=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-e2e 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); h=d['host_inclusive']
print('NRV_READ_STAGE=$v', 'step', round(d['ms_per_step'],4), {k:round(v['bases_per_s']/1e6,2) for k,v in h.items() if 'bases_per_s' in v},
      {k:(round(v['bases_per_s_device_resident']/1e6,2), round(v['bases_per_s_host_inclusive_raw_reads']/1e6,2)) for k,v in d['configs'].items()},
      'read_mode', round(d['read_mode']['bases_per_s']/1e6,2))"
done; done
