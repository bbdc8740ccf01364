#!/bin/bash
# Round 6: what changed on the host side - the host-pipeline tests, the new multirank guard, the driver's bench command with its
# new legs (C2 / C3 at 1000 / 10 000 reads, cli_e2e_human on 10 000 copied files), box facts, optionally the microbench.
# usage: gpu_r6_bench.sh TAG [micro]
TAG=${1:-r06d}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s); (free -g; df -h /dev/shm /tmp; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null) > $O/box.txt 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_hostpipe.py tests/test_gpu_range.py "tests/test_gpu_multirank.py::test_two_ranks_sharing_the_device_deliver_the_one_rank_total" -q -x -s > $O/pytest_host.log 2>&1; echo "pytest rc=$?" >> $O/pytest_host.log
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
echo "bench seconds: $(( $(date +%s) - T0 ))"
if [ "${2:-}" = micro ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/clock_vs_fill tools/microbench/clock_vs_fill.hip 2>/dev/null && \
  timeout 600 /tmp/clock_vs_fill 2000 $O/clock_vs_fill.tsv 0 > $O/clock_vs_fill.txt 2>&1; echo "clock_vs_fill rc=$?"
fi
cat $O/box.txt; tail -4 $O/pytest_host.log; tail -1 $O/bench_driver.json | cut -c1-600
