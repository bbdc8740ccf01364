"""Multi-GPU path on CPU: world_size-2 gloo run of the read-sharded driver logic (no data-path
collective; only the barrier + max-over-ranks timing reduction of bench.py)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
from nanoreviser_amd import shard


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 3, 8):
            parts = [shard.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def test_shard_reads_balanced_and_complete():
    rng = np.random.default_rng(0)
    sizes = rng.integers(500, 40000, 1003).tolist()
    for w in (1, 2, 4, 8):
        parts = shard.shard_reads(sizes, w)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(sizes)))          # nothing dropped (cf. NanoReviser.py:212)
        loads = [sum(sizes[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(sizes)
    assert shard.shard_reads([], 4) == [[], [], [], []]


def test_split_read_windows_halo():
    N, T = 200_000, 13
    parts = shard.split_read_windows(N, T, 8)
    got = []
    for lo, hi in parts:
        n_win = max(hi - lo - T, 0)
        got.extend(range(lo, lo + n_win))
    assert got == list(range(N - T))
    assert shard.split_read_windows(5, 13, 2) == [(0, 0), (0, 0)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import bench
    res = bench.run_distributed_cpu_selftest()
    q.put((rank, res))


def test_bench_sharding_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # both ranks agree on the reduced numbers; work adds up; time is the max over ranks
    assert res[0]["total_units"] == res[1]["total_units"] == 2 * 4096 * 3
    assert res[0]["max_ms"] == res[1]["max_ms"]
    assert res[0]["max_ms"] >= max(res[0]["my_ms"], res[1]["my_ms"]) - 1e-6
    assert res[0]["shard"] != res[1]["shard"]


def test_bench_gpus2_typed_as_is_dry_run():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts both ranks itself
    (gloo dry run of the launch / sharding / timing control path) and relays ONE JSON line."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world_size"] == 2 and j["steps"] == 3 and j["scaling"] == "weak"
    assert abs(j["value"] - 2 * 4096 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-3 * j["value"]
    # every rank's identity reaches rank 0's line (the real run gathers device index / PCI / UUID the same way)
    rk = j["config"]["ranks"]
    assert [x["rank"] for x in rk] == [0, 1] and len({x["pid"] for x in rk}) == 2 and rk[0]["shard"] != rk[1]["shard"]


def test_bench_gpus8_dry_run_eight_ranks():
    """The driver's widest launch, rehearsed on CPU: eight ranks, one JSON line, eight distinct shards."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--batch", "512",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["world_size"] == 8 and j["config"]["world_size"] == 8
    rk = j["config"]["ranks"]
    assert [x["rank"] for x in rk] == list(range(8)) and len({tuple(x["shard"]) for x in rk}) == 8


def test_bench_refuses_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_bench_self_launch_propagates_a_dead_rank():
    """A rank that dies must end the run with its exit code (the others would sit in the barrier)."""
    import subprocess
    import bench
    code = ("import os,sys,time\n"
            "sys.exit(7) if os.environ['RANK']=='1' else time.sleep(60)\n")
    script = os.path.join(ROOT, "tests", "_dead_rank_tmp.py")
    with open(script, "w") as fp:
        fp.write(code)
    try:
        real = bench.__file__
        bench.__file__ = script
        t0 = __import__("time").time()
        rc = bench.self_launch(2, [])
        assert rc == 7 and __import__("time").time() - t0 < 30
    finally:
        bench.__file__ = real
        os.remove(script)
