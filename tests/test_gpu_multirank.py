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


def test_two_ranks_sharing_the_device_deliver_the_one_rank_total():
    """VERDICT r05 next #10c, a regression guard on the control plane: two ranks on ONE device share it, so the whole-job
    `value` (units of all ranks / max-over-ranks time) must come out at about the one-rank value - a rank that is not
    counted, or counted twice, or a barrier that lets one rank's clock run alone, shows up as a factor of two.  Also: every
    rank's own ms_per_step sits next to its device identity, and the line carries its five timed blocks."""
    def run(n):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5", "--prime", "200",
               "--no-extras", "--no-cpu-baseline"] + (["--share-device"] if n > 1 else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    one, two = run(1), run(2)
    ratio = two["value"] / one["value"]
    assert 0.7 < ratio < 1.2, (one["value"], two["value"])
    for j, n in ((one, 1), (two, 2)):
        b = j["ms_per_step_blocks"]
        assert len(b) == 5 and b[0] == pytest.approx(j["ms_per_step"], rel=1e-4)
        lo, med, hi = j["ms_per_step_blocks_min_median_max"]
        assert lo <= med <= hi and lo == pytest.approx(min(b), rel=1e-4) and hi == pytest.approx(max(b), rel=1e-4)
        devs = j["config"]["devices"]
        assert len(devs) == n and all(x["ms_per_step"] > 0 for x in devs)
        assert max(x["ms_per_step"] for x in devs) <= j["ms_per_step"] * 1.0001
    print(f"SHARE: 1 rank {one['value']:.3e}, 2 ranks on one device {two['value']:.3e} bases/s (x{ratio:.2f}); "
          f"blocks {one['ms_per_step_blocks']}")


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


def test_bench_two_ranks_carry_a_fast5_fed_cli_leg():
    """VERDICT r04 #1a: an N > 1 line must also measure the metric as BASELINE.json words it - fast5 in, revised reads
    out, N GPU workers (NanoReviser.py:203-219) - not only the device-resident weak-scaling loop."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "6",
                        "--warmup", "2", "--prime", "30", "--no-cpu-baseline", "--cli-reps", "60"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    c = j["host_inclusive"]["cli_e2e"]
    assert "error" not in c, c
    assert c["n_gpus"] == 2 and c["share_device"] and c["reads"] == 2 * 60 * 2 == c["files_written"]
    assert c["bases_per_s"] > 0 and j["value_cli_e2e"] == c["bases_per_s"] and j["n_gpus"] == 2
    print(f"MULTIRANK cli leg: {c['bases_per_s']:.3e} bases/s wall with 2 workers on one device, {c['wall_s']:.2f} s")


def test_long_read_split_eight_ways_equals_the_unsplit_read(monkeypatch):
    """BASELINE config 5 (long reads, 8 GPUs) / SURVEY 8e: a 200 k-event read cut into 8 window ranges with a T-1-event
    halo (shard.split_read_windows -> cli.revise_part, the slice body of a GPU worker) on the real engine: the slices'
    calls, concatenated, are the unsplit read's calls - probabilities bit for bit - and the merged record is the same text."""
    import numpy as np
    from nanoreviser_amd import hoststage as hs
    from nanoreviser_amd.engine import Reviser
    from nanoreviser_amd.weights import load_species
    from conftest import load_read
    import json as _json
    keys = [e["key"] for e in _json.load(open(os.path.join(GOLD, "reads", "index.json")))]
    raws, starts, feats, bases = [], [], [], []
    off = 0
    while sum(len(s) for s in starts) < 200_000:
        for k in keys:
            _, rd, _ = load_read(k)
            rt = hs.read_tensors_raw(rd)
            raws.append(rt.raw)
            starts.append(rt.starts.astype(np.int64) + off)
            feats.append(rt.feat_ev)
            bases.append(np.asarray(rt.bases))
            off += len(rt.raw)
    big = hs.RawReadTensors(np.concatenate(raws), np.concatenate(starts).astype(np.int32), np.concatenate(feats),
                            np.concatenate(bases), rt.shift, rt.scale)
    N = len(big.feat_ev)
    assert N >= 200_000
    m1, m2 = load_species("human")
    rv = Reviser(m1, m2, device=0)
    T = rv.T
    whole = rv.predict_reads_raw([big.raw], [big.starts], [big.feat_ev], [big.shift], [big.scale])
    monkeypatch.setattr(cli, "_load_one", lambda job, native=True: (job[1], big, None, None, 0.0))
    args = cli.get_args(["-d", "unused/", "-o", "unused/", "-S", "human", "-F", "fastq"])
    got = {}
    for k in range(8):
        payload, err = cli.revise_part(args, rv, "long.fast5", k, 8)
        assert err is None
        got[k] = (payload, None)
    a1 = np.concatenate([got[k][0]["a1"] for k in range(8)])
    a2 = np.concatenate([got[k][0]["a2"] for k in range(8)])
    qc = np.concatenate([got[k][0]["qc"] for k in range(8)])
    assert len(a1) == N - T and np.array_equal(a1, whole[2]) and np.array_equal(a2, whole[3])
    assert np.array_equal(qc, cli.phred_chars(*whole))
    # one slice again, with the probabilities: bit for bit the unsplit read's rows
    from nanoreviser_amd.shard import split_read_windows
    lo, hi = split_read_windows(N, T, 8)[5]
    p1, p2, _, _ = rv.predict_reads_raw([big.raw], [big.starts[lo:hi]], [big.feat_ev[lo:hi]], [big.shift], [big.scale])
    assert np.array_equal(p1.view(np.uint32), whole[0][lo:hi - T].view(np.uint32))
    assert np.array_equal(p2.view(np.uint32), whole[1][lo:hi - T].view(np.uint32))
    rv.close()


def test_cli_splits_reads_over_real_workers_on_one_device(tmp_path, monkeypatch):
    """The file-level path of the same: `--split_reads_above` below the fixtures' size, 3 real workers on device 0,
    byte-identical to the one-worker run (FASTQ: the qualities cross the process boundary too)."""
    one, three = str(tmp_path) + "/one/", str(tmp_path) + "/three/"
    assert cli.main(["-d", FAST5, "-o", one, "-S", "ecoli", "-F", "fastq", "--thread", "2", "--gpus", "1"]) == 0
    assert cli.main(["-d", FAST5, "-o", three, "-S", "ecoli", "-F", "fastq", "--thread", "2", "--split_reads_above", "0.2"],
                    worker_factory=shared_device_factory, world=3) == 0
    names = sorted(f for f in os.listdir(one) if f.endswith("_out.fastq"))
    assert len(names) == 2
    for f in names:
        assert open(one + f, "rb").read() == open(three + f, "rb").read(), f
    assert open(three + "failed_reads.txt").read() == ""
