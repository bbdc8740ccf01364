"""The N-process paths with the REAL engine on a 1-GPU box (MI355X only, -m gpu): `bench.py --gpus N
--share-device` (ranks map to device r % device_count; control plane on gloo, no RCCL) and the command line's
per-GPU workers (cli.run_workers) with two real workers on device 0.  What the reference does with a Pool
(NanoReviser.py:203-219); the 8-GPU run itself is the driver's."""
import glob
import json
import os
import shutil
import subprocess
import sys

import pytest

from conftest import GOLD
from nanoreviser_amd import cli
from echo_engine import shared_device_factory

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST5 = os.path.join(GOLD, "fast5")


@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_n_ranks_share_one_device(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--share-device",
                        "--steps", "6", "--warmup", "2", "--prime", "30", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                        # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["world_size"] == n and j["steps"] == 6 and j["scaling"] == "weak"
    assert j["value"] > 0 and abs(j["value"] - n * 4096 * 6 / (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]
    assert "--share-device" in j["config"]["parallelism"] and "gloo" in j["config"]["parallelism"]
    # every rank's device identity is in the line: n entries, and on this one-GPU box ONE distinct device
    import torch
    devs = j["config"]["devices"]
    assert [x["rank"] for x in devs] == list(range(n)) and j["config"]["world_size"] == n
    assert j["config"]["distinct_devices"] == min(n, torch.cuda.device_count())
    assert all(x["pci"] and x["name"] for x in devs)
    mm = j["rank_ms_per_step"]
    assert 0 < mm["min"] <= mm["max"] <= j["ms_per_step"] * 1.0001
    assert j["roofline"]["frac"] > 0 and j["f16x2_range_guard"]["pending_after_timed_region"] == 0
    print(f"MULTIRANK n={n}: {j['value']:.3e} bases/s, ms/step {j['ms_per_step']:.3f}, per rank {mm}")


def test_bench_refuses_more_ranks_than_devices_without_share_device():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--prime", "2", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs")
    assert r.returncode != 0 and "--share-device" in r.stderr


def test_cli_two_real_workers_on_one_device_match_one_worker(tmp_path):
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    for i in range(6):
        shutil.copy(src[i % 2], d / f"read{i}.fast5")
    one, two = str(tmp_path) + "/one/", str(tmp_path) + "/two/"
    assert cli.main(["-d", str(d), "-o", one, "-S", "ecoli", "--thread", "2", "--gpus", "1"]) == 0
    assert cli.main(["-d", str(d), "-o", two, "-S", "ecoli", "--thread", "2"],
                    worker_factory=shared_device_factory, world=2) == 0
    names = sorted(f for f in os.listdir(one) if f.endswith("_out.fasta"))
    assert len(names) == 6 and names == sorted(f for f in os.listdir(two) if f.endswith("_out.fasta"))
    for f in names:
        assert open(one + f, "rb").read() == open(two + f, "rb").read(), f
    assert open(two + "failed_reads.txt").read() == ""


def test_cli_eight_real_workers_on_one_device_match_one_worker(tmp_path, monkeypatch):
    """The widest fan-out the command line will see (one worker per GPU of an 8-GPU node), rehearsed on one device:
    outputs byte-identical to one worker, and the parser processes of all workers together stay within the cores
    the process may use (each worker used to size its pool for the whole machine)."""
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    for i in range(24):
        shutil.copy(src[i % 2], d / f"read{i:02d}.fast5")
    one, eight = str(tmp_path) + "/one/", str(tmp_path) + "/eight/"
    assert cli.main(["-d", str(d), "-o", one, "-S", "ecoli", "--thread", "4", "--gpus", "1"]) == 0
    assert cli.main(["-d", str(d), "-o", eight, "-S", "ecoli", "--thread", "100"],
                    worker_factory=shared_device_factory, world=8) == 0
    names = sorted(f for f in os.listdir(one) if f.endswith("_out.fasta"))
    assert len(names) == 24 and names == sorted(f for f in os.listdir(eight) if f.endswith("_out.fasta"))
    for f in names:
        assert open(one + f, "rb").read() == open(eight + f, "rb").read(), f
    assert open(eight + "failed_reads.txt").read() == ""
    assert 8 * cli.parser_pool_size(100, cli.usable_cores(), 8, 3) <= max(8, cli.usable_cores())
