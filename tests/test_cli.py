"""The NanoReviser.py counterpart (nanoreviser_amd/cli.py): flag surface, file contract, failure
contract.  CPU tests drive it with a stand-in engine object (host logic only, no compute); the
GPU test runs the real thing end to end on two reference fixture reads."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLD, load_read
from nanoreviser_amd import cli
from nanoreviser_amd import hoststage as hs
from echo_engine import (HashEngine, hash_factory, hash_dies_on_slice_factory, EchoEngine, _ExclusiveEcho, exclusive_factory, echo_factory, dying_factory, dying_midway_factory, broken_factory,
                         fails_then_dies_factory)

FAST5 = os.path.join(GOLD, "fast5")


def test_flag_surface_matches_reference():
    # NanoReviser.py:42-95
    a = cli.get_args(["-d", "in/", "-o", "out/", "-F", "fastq", "-S", "ecoli", "--thread", "7", "-t", "tmp/",
                      "-e", "bad.txt", "-g", "Basecall_1D_001", "-s", "BaseCalled_complement", "--test_mode",
                      "--model1_predict_dir", "a.h5", "--model2_predict_dir", "b.h5"])
    assert (a.fast5_base_dir, a.output_dir, a.output_format, a.species, a.thread) == ("in/", "out/", "fastq", "ecoli", 7)
    assert (a.temp_dir, a.failed_reads_filename, a.basecall_group, a.basecall_subgroup) == \
        ("tmp/", "bad.txt", "Basecall_1D_001", "BaseCalled_complement")
    assert a.test_mode and a.model1_predict_dir == "a.h5" and a.model2_predict_dir == "b.h5"
    d = cli.get_args(["-d", "x"])
    assert (d.output_dir, d.output_format, d.species, d.thread, d.temp_dir, d.failed_reads_filename,
            d.basecall_group, d.basecall_subgroup) == ("./unitest/nanorev_output/", "fasta", "human", 100,
                                                       "./unitest/tmp/", "failed_reads.txt", "Basecall_1D_000",
                                                       "BaseCalled_template")
    with pytest.raises(SystemExit):
        cli.get_args(["-v"])
    with pytest.raises(SystemExit):
        cli.get_args([])                        # -d missing -> help, exit (NanoReviser.py:93-95)
    p1, p2 = cli.model_paths(cli.get_args(["-d", "x", "-S", "ecoli"]))
    assert p1.endswith("model/ecoli/ecoli_win13_50ep_model1.h5") and p2.endswith("ecoli_win13_50ep_model2.h5")
    with pytest.raises(RuntimeError):
        cli.model_paths(cli.get_args(["-d", "x", "-S", "yeast"]))


def test_parse_read_equals_reference_get_read_data():
    for p in sorted(glob.glob(os.path.join(FAST5, "*.fast5"))):
        key = "_".join(os.path.basename(p).split("_")[-3:-1])
        g, rd_ref, _ = load_read(key)
        rd, fq = cli.parse_read(p, "Basecall_1D_000", "BaseCalled_template")
        assert rd.abs_event_start == int(g["rd_abs_event_start"])
        assert np.array_equal(rd.start, g["rd_start"]) and np.array_equal(rd.bases, g["rd_bases"])
        assert np.array_equal(rd.ab_mean, g["rd_ab_mean"]) and np.array_equal(rd.length, g["rd_length"])
        assert hs.trim_fastq(fq) == (bytes(g["fq_bases"]).decode(), bytes(g["fq_qual"]).decode())


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
def test_file_contract_with_echo_engine(tmp_path, fmt):
    out = str(tmp_path) + "/out/"
    rc = cli.main(["-d", FAST5, "-o", out, "-F", fmt, "-S", "ecoli", "--thread", "2"],
                  reviser_factory=lambda args, dev: EchoEngine())
    assert rc == 0
    files = sorted(os.listdir(FAST5))
    for fn in files:
        key = "_".join(fn.split("_")[-3:-1])
        _, rd, _ = load_read(key)
        orig = "".join(b.decode() for b in rd.bases.tolist())
        path = out + fn.split(".")[0] + "_out." + fmt                  # NanoReviser.py:137/163
        text = open(path).read()
        if fmt == "fasta":
            assert text == ">" + fn + "\n" + orig                      # output_handeler.py:37-38, no trailing \n
        else:
            head, rest = text.split("\n", 1)
            assert head == "@" + fn
            seq, qual = rest.split("+\n")                              # output_handeler.py:52-55 (no \n before +)
            assert seq == orig and len(qual) == len(seq)
            assert qual[:5] == "#####" and set(qual[5:-6]) == {chr(33 + 11)}   # conf 0.9167 -> Q11
    assert open(out + "failed_reads.txt").read() == ""


def test_failure_contract_writes_original_bases(tmp_path):
    out = str(tmp_path) + "/o/"
    files = sorted(os.listdir(FAST5))
    _, _, rt0 = load_read("_".join(files[0].split("_")[-3:-1]))
    eng = EchoEngine(fail_marker=rt0.feat_ev[0])
    rc = cli.main(["-d", FAST5, "-o", out, "-S", "ecoli", "--thread", "1", "-e", "bad.txt"],
                  reviser_factory=lambda args, dev: eng)
    assert rc == 0 and eng.calls == 1 + len(files)                     # the batch, then every read of it on its own
    failed = open(out + "bad.txt").read().split()
    assert failed == [files[0]]                                        # --failed_read is honoured
    for fn in files:                                                   # every read still produces a file
        key = "_".join(fn.split("_")[-3:-1])
        _, rd, _ = load_read(key)
        orig = "".join(b.decode() for b in rd.bases.tolist())
        assert open(out + fn.split(".")[0] + "_out.fasta").read() == ">" + fn + "\n" + orig
    # fastq fallback = the original record trimmed as extract_fastq does (nanorev_fast5_handeler.py:152-171)
    out2 = str(tmp_path) + "/q/"
    cli.main(["-d", FAST5, "-o", out2, "-F", "fastq", "-S", "ecoli", "--thread", "1"],
             reviser_factory=lambda args, dev: EchoEngine(fail_marker=rt0.feat_ev[0]))
    g, _, _ = load_read("_".join(files[0].split("_")[-3:-1]))
    text = open(out2 + files[0].split(".")[0] + "_out.fastq").read()
    assert text == "@" + files[0] + "\n" + bytes(g["fq_bases"]).decode() + "+\n" + bytes(g["fq_qual"]).decode()


def test_broken_fast5_is_logged_not_fatal(tmp_path):
    d = tmp_path / "in"
    d.mkdir()
    (d / "broken.fast5").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\x00" * 64)
    out = str(tmp_path) + "/o/"
    rc = cli.main(["-d", str(d), "-o", out, "-S", "ecoli"], reviser_factory=lambda a, dev: EchoEngine())
    assert rc == 0 and open(out + "failed_reads.txt").read().split() == ["broken.fast5"]
    assert not glob.glob(out + "*_out.fasta")


def test_parser_pool_is_a_share_of_the_cores_per_gpu_worker(tmp_path, monkeypatch):
    """eight GPU workers on a 16-core cgroup must start 8 x 2 parser processes, not 8 x 16."""
    assert cli.parser_pool_size(100, 16, 1, 4000) == 16
    assert cli.parser_pool_size(100, 16, 8, 4000) == 2
    assert cli.parser_pool_size(100, 16, 32, 4000) == 1            # never zero
    assert cli.parser_pool_size(4, 256, 8, 4000) == 4              # --thread still caps it
    assert cli.parser_pool_size(100, 256, 2, 4000) == 32           # ... and so does 32
    assert cli.parser_pool_size(100, 16, 1, 3) == 3
    assert 8 * cli.parser_pool_size(100, 16, 8, 4000) <= 16
    # through the real call: process_files reports the pool it used
    monkeypatch.setattr(cli, "usable_cores", lambda: 6)
    args = cli.get_args(["-d", FAST5, "-o", str(tmp_path) + "/o/", "-S", "ecoli"])
    os.makedirs(args.output_dir, exist_ok=True)
    st = cli.process_files(args, sorted(os.listdir(FAST5)), EchoEngine(), lambda m: None, gpu_workers=3)
    assert st["parser_workers"] == 2 and st["reads"] == 2


def test_trace_summary_survives_a_run_without_device_calls(tmp_path, monkeypatch):
    """with NRV_CLI_TRACE=1 and no device call at all (every file unparsable) the summary used to index
    calls[0]; the failed reads must still come back."""
    monkeypatch.setenv("NRV_CLI_TRACE", "1")
    d = tmp_path / "in"
    d.mkdir()
    (d / "broken.fast5").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\x00" * 64)
    out = str(tmp_path) + "/o/"
    logged = []
    args = cli.get_args(["-d", str(d), "-o", out, "-S", "ecoli"])
    os.makedirs(out, exist_ok=True)
    st = cli.process_files(args, ["broken.fast5"], EchoEngine(), logged.append)
    assert st["failed"] == ["broken.fast5"] and any("no device call" in m for m in logged)


def _orig(fn):
    _, rd, _ = load_read("_".join(fn.split("_")[-3:-1]))
    return "".join(b.decode() for b in rd.bases.tolist())


def test_multi_gpu_worker_path_world2(tmp_path):
    """cli.run_workers / _worker with two spawned worker processes (one per 'GPU') and an injected
    engine factory: reads are sharded, both shards are revised, stats and failed_reads are merged."""
    out = str(tmp_path) + "/o/"
    rc = cli.main(["-d", FAST5, "-o", out, "-S", "ecoli", "--thread", "1"], worker_factory=echo_factory, world=2)
    assert rc == 0
    for fn in sorted(os.listdir(FAST5)):
        assert open(out + fn.split(".")[0] + "_out.fasta").read() == ">" + fn + "\n" + _orig(fn)
    assert open(out + "failed_reads.txt").read() == ""


@pytest.mark.parametrize("factory", [dying_factory, dying_midway_factory, broken_factory])
def test_multi_gpu_worker_that_dies_does_not_hang_the_cli(tmp_path, factory):
    """A worker that dies hard (os._exit: what a HIP memory fault / segfault / OOM kill looks like
    from outside) posts nothing to the result queue.  The parent must notice, let the other worker
    finish, write the ORIGINAL basecalls for the dead worker's reads (NanoReviser.py:146-152), list
    them in failed_reads and return non-zero.  broken_factory: the engine cannot be created (raises)."""
    import time
    out = str(tmp_path) + "/o/"
    t0 = time.time()
    rc = cli.main(["-d", FAST5, "-o", out, "-S", "ecoli", "--thread", "1", "-e", "bad.txt"],
                  worker_factory=factory, world=2)
    assert rc == 3 and time.time() - t0 < 120
    files = sorted(os.listdir(FAST5))
    failed = open(out + "bad.txt").read().split()
    assert len(failed) == 1 and failed[0] in files                     # the dead worker's single read
    for fn in files:                                                   # every read still has an output
        assert open(out + fn.split(".")[0] + "_out.fasta").read() == ">" + fn + "\n" + _orig(fn)


def test_resume_skips_reads_whose_output_exists(tmp_path):
    """--resume (the reference clears temp_dir and re-revises / overwrites every read, NanoReviser.py:196-201; SURVEY.md
    5: one file per read makes skip-if-exists a free resume): finished reads are not touched, not even parsed; an EMPTY
    output is not a finished one; a read the earlier run wrote UNREVISED (listed in its failed-reads file) is revised again
    and replaced - or, with --resume_keep_failed, left alone, still listed, and the run returns 3."""
    import shutil
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    for i in range(6):
        shutil.copy(src[i % 2], d / f"r{i}.fast5")
    (d / "broken.fast5").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\x00" * 64)
    out = str(tmp_path) + "/o/"
    os.makedirs(out)
    open(out + "r1_out.fasta", "w").write("FINISHED EARLIER")          # kept as it is
    open(out + "r2_out.fasta", "w").write("")                           # empty: not finished
    open(out + "r4_out.fasta", "w").write(">r4.fast5\nACGT")           # an earlier run's fallback output ...
    open(out + "failed_reads.txt", "w").write("r4.fast5\nr5.fast5\n")   # ... listed as failed; r5 has no output: redone
    keep = EchoEngine()                                                  # --resume_keep_failed: r4 is left alone, exit code 3
    assert cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "1", "--resume", "--resume_keep_failed"],
                    reviser_factory=lambda a, dev: keep) == 3
    assert open(out + "r4_out.fasta").read() == ">r4.fast5\nACGT"
    assert open(out + "failed_reads.txt").read().split() == ["r4.fast5", "broken.fast5"]
    for i in (0, 2, 3, 5):
        os.remove(out + f"r{i}_out.fasta")
    open(out + "r2_out.fasta", "w").write("")
    open(out + "failed_reads.txt", "w").write("r4.fast5\nr5.fast5\n")
    eng = EchoEngine()
    assert cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "1", "--resume"], reviser_factory=lambda a, dev: eng) == 0
    assert open(out + "r1_out.fasta").read() == "FINISHED EARLIER"
    for i in (0, 2, 3, 4, 5):                                           # r4: the earlier run's fallback output is REPLACED
        assert open(out + f"r{i}_out.fasta").read() == f">r{i}.fast5\n" + _orig(os.path.basename(src[i % 2]))
    assert open(out + "failed_reads.txt").read().split() == ["broken.fast5"]
    assert not [f for f in os.listdir(out) if ".tmp" in f]
    # a second --resume has nothing left but the unparsable file; without the flag everything is redone
    eng2 = EchoEngine()
    assert cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "1", "--resume"], reviser_factory=lambda a, dev: eng2) == 0
    assert eng2.calls == 0 and open(out + "failed_reads.txt").read().split() == ["broken.fast5"]
    assert cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "1"], reviser_factory=lambda a, dev: EchoEngine()) == 0
    assert open(out + "r1_out.fasta").read() == ">r1.fast5\n" + _orig(os.path.basename(src[1]))
    assert open(out + "failed_reads.txt").read().split() == ["broken.fast5"]


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
def test_long_read_is_split_over_the_gpu_workers_with_a_halo(tmp_path, fmt):
    """BASELINE config 5 / SURVEY 8e: a read too long for one worker's share is cut by WINDOW RANGE, T-1 events of halo
    (shard.split_read_windows), every slice revised by another worker process, the parent merges.  The engine's calls
    depend on the centre event (deletions, insertions, disagreements included), so the files equal the unsplit run's
    byte for byte only if the slices tile the read exactly.  Threshold 0.2 MB, three workers: each 0.7-0.9 MB fixture is above
    the fair share (0.53 MB) and becomes two slices."""
    one, three = str(tmp_path) + "/one/", str(tmp_path) + "/three/"
    assert cli.main(["-d", FAST5, "-o", one, "-S", "ecoli", "-F", fmt, "--thread", "1"], reviser_factory=lambda a, dev: HashEngine()) == 0
    units, _ = cli.plan_splits(["a", "b"], [900_000, 100_000], 3, 0.2)
    assert units == [("a", 0, 3), ("a", 1, 3), ("a", 2, 3), "b"]            # 0.9 MB against a fair share of 0.33 MB
    assert cli.plan_splits(["a"], [900_000], 1, 0.2)[0] == ["a"] and cli.plan_splits(["a"], [900_000], 8, 0)[0] == ["a"]
    # a long read among many: it does not unbalance anything, and every slice would cost another parse of the file
    many = cli.plan_splits(["big"] + [f"r{i}" for i in range(400)], [40 << 20] + [800_000] * 400, 8, 4)[0]
    assert many[0] == "big" and len(many) == 401
    # three long reads on eight GPUs: each above its fair share (37.5 MB) -> 3 slices each
    assert [u for u in cli.plan_splits(["x", "y", "z"], [100 << 20] * 3, 8, 4)[0] if u[0] == "x"] == [("x", k, 3) for k in range(3)]
    assert cli.main(["-d", FAST5, "-o", three, "-S", "ecoli", "-F", fmt, "--thread", "1", "--split_reads_above", "0.2"],
                    worker_factory=hash_factory, world=3) == 0
    names = sorted(f for f in os.listdir(one) if f.endswith("_out." + fmt))
    assert len(names) == 2 and names == sorted(f for f in os.listdir(three) if f.endswith("_out." + fmt))
    for f in names:
        a, b = open(one + f, "rb").read(), open(three + f, "rb").read()
        assert a == b and len(a) > 6000, f
    orig = _orig(sorted(os.listdir(FAST5))[0])
    assert open(one + names[0]).read().split("\n")[1] != orig               # the hash engine really edits the read
    assert open(three + "failed_reads.txt").read() == ""


def test_split_read_whose_worker_dies_gets_its_original_bases(tmp_path, monkeypatch):
    out = str(tmp_path) + "/o/"
    monkeypatch.setattr(cli, "plan_splits", lambda names, sizes, world, mb: (
        [(names[0], 0, 2), (names[0], 1, 2), (names[1], 0, 2), (names[1], 1, 2)], [1, 1, 1, 1]))
    monkeypatch.setattr(cli, "shard_reads", lambda sizes, world: [[0, 2], [1, 3]])     # worker 1 (the one that dies) holds a slice of each
    rc = cli.main(["-d", FAST5, "-o", out, "-S", "ecoli", "--thread", "1", "--split_reads_above", "0.4"],
                  worker_factory=hash_dies_on_slice_factory, world=2)
    assert rc == 3
    files = sorted(os.listdir(FAST5))
    assert sorted(open(out + "failed_reads.txt").read().split()) == files       # both reads had a slice on the dead worker
    for fn in files:
        assert open(out + fn.split(".")[0] + "_out.fasta").read() == ">" + fn + "\n" + _orig(fn)


def test_split_read_slices_that_do_not_tile_fall_back_to_the_original(tmp_path):
    """Parent side of a split read (cli.finish_split_reads): slices are merged only when they tile the read exactly - same event
    count and window length from every worker, each slice where shard.split_read_windows puts it, slice 0 carrying the bases; a
    missing, failed or misplaced slice gives the original basecalls and a failed-reads entry (NanoReviser.py:146-152)."""
    from nanoreviser_amd.shard import split_read_windows
    fn = sorted(os.listdir(FAST5))[0]
    rd, _ = cli.parse_read(os.path.join(FAST5, fn), "Basecall_1D_000", "BaseCalled_template")
    rt = hs.read_tensors_raw(rd)
    N, T = len(rt.feat_ev), 11
    eng = HashEngine()
    whole = eng.predict_reads_raw([rt.raw], [rt.starts], [rt.feat_ev], [rt.shift], [rt.scale])

    def parts(n, tamper=None):
        got = {}
        for k, (lo, hi) in enumerate(split_read_windows(N, T, n)):
            a1, a2 = whole[2][lo:hi - T], whole[3][lo:hi - T]
            pl = {"T": T, "n_ev": N, "lo": lo, "a1": a1, "a2": a2, "qc": None}
            if k == 0:
                pl["bases"], pl["fq"] = np.asarray(rt.bases), None
            got[k] = (pl, None)
        if tamper:
            tamper(got)
        return {fn: got}
    logs = []

    def run(tag, got, n=3):
        args = cli.get_args(["-d", FAST5 + "/", "-o", str(tmp_path / tag) + "/", "-S", "ecoli"])
        nb, failed = cli.finish_split_reads(args, {fn: n}, got, logs.append)
        return nb, failed, open(cli.out_name(args.output_dir, fn, "fasta")).read().split("\n", 1)[1]
    one = str(tmp_path / "one") + "/"
    assert cli.main(["-d", FAST5, "-o", one, "-S", "ecoli", "--thread", "1"], reviser_factory=lambda a, d: HashEngine()) == 0
    want = open(cli.out_name(one, fn, "fasta")).read().split("\n", 1)[1]
    nb, failed, text = run("ok", parts(3))
    assert failed == [] and text == want and nb == len(want)
    orig = _orig(fn)
    for tag, tamper in (("missing", lambda g: g.pop(1)),
                        ("failed", lambda g: g.__setitem__(2, (None, "engine error"))),
                        ("shifted", lambda g: g[1][0].__setitem__("lo", g[1][0]["lo"] + 1)),
                        ("short", lambda g: g[2][0].__setitem__("a1", g[2][0]["a1"][:-1])),
                        ("other_T", lambda g: g[1][0].__setitem__("T", 13)),
                        ("no_bases", lambda g: g[0][0].pop("bases"))):
        nb, failed, text = run(tag, parts(3, tamper))
        assert failed == [fn] and nb == 0 and text == orig, tag
    assert sum("writing the original basecalls" in m for m in logs) == 6


def _fake_sysfs(root, gpus):
    """gpus: [(domain, bus, numa cpulist)] -> a /sys tree with a CPU node 0 and one KFD node per GPU."""
    nodes = root / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\n")
    for i, (dom, bus, cpus) in enumerate(gpus):
        (nodes / str(i + 1)).mkdir()
        (nodes / str(i + 1) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndomain {dom}\nlocation_id {bus << 8}\n")
        dev = root / "bus" / "pci" / "devices" / ("%04x:%02x:00.0" % (dom, bus))
        dev.mkdir(parents=True)
        (dev / "local_cpulist").write_text(cpus + "\n")


def test_gpu_workers_are_pinned_to_their_gpus_numa_cores(tmp_path, monkeypatch):
    """VERDICT r04 #1c: eight workers on a two-socket node: each gets cores of ITS GPU's node, workers that share a node
    share it evenly, the slices are disjoint and stay inside what the process may use."""
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "NRV_CPU_AFFINITY"):
        monkeypatch.delenv(v, raising=False)
    _fake_sysfs(tmp_path, [(0, 0x05 + 8 * i, "0-31,64-95" if i < 4 else "32-63,96-127") for i in range(8)])
    sysfs = str(tmp_path)
    assert cli.gpu_local_cpus(0, sysfs) == list(range(0, 32)) + list(range(64, 96))
    assert cli.gpu_local_cpus(7, sysfs)[0] == 32 and cli.gpu_local_cpus(8, sysfs) is None
    sets = [cli.worker_cpus(r, 8, allowed=range(128), sysfs=sysfs) for r in range(8)]
    assert all(len(c) == 16 for c in sets) and len(set().union(*map(set, sets))) == 128
    assert all(set(c) <= set(cli.gpu_local_cpus(r, sysfs)) for r, c in enumerate(sets))
    # a cgroup that grants 16 of those cores: 2 each, still on the right node where the node has any
    few = [cli.worker_cpus(r, 8, allowed=list(range(0, 8)) + list(range(32, 40)), sysfs=sysfs) for r in range(8)]
    assert few[0] == [0, 1] and few[3] == [6, 7] and few[4] == [32, 33] and few[7] == [38, 39]
    # all workers on ONE device (rehearsal): its node's cores divided among them
    shared = [cli.worker_cpus(r, 4, devices=[0] * 4, allowed=range(128), sysfs=sysfs) for r in range(4)]
    assert [len(c) for c in shared] == [16] * 4 and not set(shared[0]) & set(shared[1])
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "6,7")                    # a re-mapped device list is honoured
    assert cli.gpu_local_cpus(0, sysfs)[0] == 32
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # no topology (this container): contiguous slices of what is allowed; too few cores or NRV_CPU_AFFINITY=0: hands off
    assert cli.worker_cpus(1, 2, allowed=range(8), sysfs=str(tmp_path / "none")) == [4, 5, 6, 7]
    assert cli.worker_cpus(0, 8, allowed=range(8), sysfs=sysfs) is None
    # two hardware threads per core (cpu k and k + 64 are siblings): workers get ONE thread of each core, so that two busy parser
    # threads never share a core (r05: the same host stage ran at 70-80 M or 115-125 M bases/s depending on that placement)
    for c in range(128):
        d = tmp_path / "devices" / "system" / "cpu" / f"cpu{c}" / "topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(f"{c % 64},{c % 64 + 64}\n")
    assert cli.primary_threads(range(128), sysfs) == list(range(64))
    assert cli.primary_threads([3, 64, 67, 70], sysfs) == [3, 64, 70]           # a core whose first thread is not ours keeps the other
    smt = [cli.worker_cpus(r, 8, allowed=range(128), sysfs=sysfs) for r in range(8)]
    assert all(len(c) == 8 and max(c) < 64 for c in smt) and len(set().union(*map(set, smt))) == 64
    assert all(set(c) <= set(cli.gpu_local_cpus(r, sysfs)) for r, c in enumerate(smt))
    monkeypatch.setenv("NRV_CPU_AFFINITY", "0")
    assert cli.worker_cpus(0, 2, allowed=range(128), sysfs=sysfs) is None


def test_dead_worker_bookkeeping_uses_reported_files_not_file_existence(tmp_path, monkeypatch):
    """Worker 1 fails its first read (original bases written, reported failed), revises the second, then dies
    in its second batch.  The parent must keep the failed read's failed_reads entry, NOT list the revised read,
    write originals for the two reads that never became final - replacing a stale output an earlier run left
    under one of those names - and leave no temporary files behind."""
    import shutil
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    names = ["r0_A", "r1_A", "r2_B", "r3_B", "r4_A", "r5_B", "r6_B", "r7_B"]
    for n in names:
        shutil.copy(src[0] if n.endswith("A") else src[1], d / (n + ".fast5"))
    monkeypatch.setattr(cli, "shard_reads", lambda sizes, world: [[0, 1, 2, 3], [4, 5, 6, 7]])
    monkeypatch.setenv("NRV_CLI_GROUPS", "8")         # device calls of two of these reads (8 x 1024 events), as the scenario needs
    out = str(tmp_path) + "/o/"
    os.makedirs(out)
    open(out + "r6_B_out.fasta", "w").write("STALE OUTPUT OF AN EARLIER RUN")
    rc = cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "1", "--batch", "1024", "-e", "bad.txt"],
                  worker_factory=fails_then_dies_factory, world=2)
    assert rc == 3
    assert sorted(open(out + "bad.txt").read().split()) == ["r4_A.fast5", "r6_B.fast5", "r7_B.fast5"]
    orig = {"A": _orig(os.path.basename(src[0])), "B": _orig(os.path.basename(src[1]))}
    for n in names:
        assert open(out + n + "_out.fasta").read() == ">" + n + ".fast5\n" + orig[n[-1]]
    assert not [f for f in os.listdir(out) if ".tmp" in f]


@pytest.mark.parametrize("fail", [False, True])
def test_pooled_workers_hand_over_bundles(tmp_path, fail):
    """With a parser pool the worker processes concatenate the reads of a task themselves (cli._load_bundle) and
    the main process issues one device call per bundle.  Same file contract; and when the bundled call fails the
    reads are retried one by one from views of the bundle, so only the failing read falls back to its original
    bases and is listed as failed."""
    import shutil
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    names = [f"b{i}_{'AB'[i % 2]}" for i in range(10)]
    for n in names:
        shutil.copy(src[0] if n.endswith("A") else src[1], d / (n + ".fast5"))
    out = str(tmp_path) + "/o/"
    marker = None
    if fail:                                                           # every call that STARTS with a read A fails
        _, _, rtA = load_read("_".join(os.path.basename(src[0]).split("_")[-3:-1]))
        marker = rtA.feat_ev[0]
    eng = EchoEngine(fail_marker=marker)
    rc = cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "2", "-e", "bad.txt"],
                  reviser_factory=lambda args, dev: eng)
    assert rc == 0
    orig = {"A": _orig(os.path.basename(src[0])), "B": _orig(os.path.basename(src[1]))}
    for n in names:
        assert open(out + n + "_out.fasta").read() == ">" + n + ".fast5\n" + orig[n[-1]]
    failed = sorted(open(out + "bad.txt").read().split())
    if fail:
        # a bundle that starts with a read A fails as a whole and is retried read by read: its A reads fail again,
        # its B reads are revised; bundles that start with a B go through untouched (the task size decides which)
        assert failed and all(f[:-6].endswith("A") for f in failed) and "b0_A.fast5" in failed
    else:
        assert failed == []
    assert eng.calls >= 2                                              # at least two bundles (+ retries)


def test_two_engines_per_device_are_two_engines(tmp_path, monkeypatch):
    """NRV_CLI_ENGINES=2: the engine factory is called twice, the two engine threads
    never share an engine (r03: the factory wrapper handed the FIRST engine out twice - two threads inside one native
    handle - and whole launch groups came back wrong), and the files are those of a one-engine run."""
    import shutil
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    names = [f"e{i}_{'AB'[i % 2]}" for i in range(12)]
    for n in names:
        shutil.copy(src[0] if n.endswith("A") else src[1], d / (n + ".fast5"))
    out = str(tmp_path) + "/o/"
    _ExclusiveEcho.made.clear()
    _ExclusiveEcho.violations.clear()
    monkeypatch.setenv("NRV_CLI_ENGINES", "2")
    monkeypatch.setenv("NRV_CLI_GROUPS", "2")          # several device calls for these twelve reads
    rc = cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "2", "--batch", "1024", "-e", "bad.txt"],
                  worker_factory=exclusive_factory, world=1)
    assert rc == 0
    assert len(_ExclusiveEcho.made) == 2 and _ExclusiveEcho.made[0] is not _ExclusiveEcho.made[1]
    assert _ExclusiveEcho.violations == []
    assert all(e.calls >= 1 for e in _ExclusiveEcho.made)               # both engines served calls
    assert open(out + "bad.txt").read().split() == []
    orig = {"A": _orig(os.path.basename(src[0])), "B": _orig(os.path.basename(src[1]))}
    for n in names:
        assert open(out + n + "_out.fasta").read() == ">" + n + ".fast5\n" + orig[n[-1]]


def test_vlen_string_fastq_does_not_discard_the_read(monkeypatch):
    """h5lite returns str for variable-length string datasets: parse_read must take both."""
    from nanoreviser_amd import h5lite
    p = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))[0]
    real = h5lite.read_fast5

    def as_str(path, g, sg):
        d = real(path, g, sg)
        d["fastq"] = bytes(d["fastq"]).decode()
        return d
    want = cli.parse_read(p, "Basecall_1D_000", "BaseCalled_template")
    monkeypatch.setattr(h5lite, "read_fast5", as_str)
    rd, fq = cli.parse_read(p, "Basecall_1D_000", "BaseCalled_template")
    assert fq == want[1] and isinstance(fq, str) and np.array_equal(rd.bases, want[0].bases)


@pytest.mark.gpu
def test_cli_end_to_end_on_gpu(tmp_path, species_models):
    from nanoreviser_amd.engine import Reviser
    out = str(tmp_path) + "/out/"
    assert cli.main(["-d", FAST5, "-o", out, "-S", "ecoli", "--gpus", "1"]) == 0
    rv = Reviser(*species_models["ecoli"])
    for fn in sorted(os.listdir(FAST5)):
        key = "_".join(fn.split("_")[-3:-1])
        _, rd, rt = load_read(key)
        p1, p2, a1, a2 = rv.predict_read(rt.sig_ev, rt.feat_ev)
        want = hs.revise_read(rd.bases, a1, a2, 11)
        text = open(out + fn.split(".")[0] + "_out.fasta").read()
        assert text == ">" + fn + "\n" + want
        orig = "".join(b.decode() for b in rd.bases.tolist())
        lab = np.array([hs.BASE_LABEL[c] for c in orig])[5:5 + len(a1)]
        assert (a1 == lab).mean() > 0.95                  # most bases are confirmed, some are revised
        assert want != orig and abs(len(want) - len(orig)) < 0.05 * len(orig)
        assert want[:5] == orig[:5] and set(want) <= set("ACGT")
    assert open(out + "failed_reads.txt").read() == ""
    rv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
def test_pipelined_device_calls_write_the_files_of_one_call_at_a_time_on_gpu(tmp_path, monkeypatch, fmt):
    """The REAL engine: 60 reads (all five fixture reads, interleaved) in small device calls, pipelined two deep
    (nrv_reads_raw_begin / _end; the default) against one call at a time (NRV_CLI_PIPELINE=0) against round 5's staged calls
    (NRV_RAW_STAGED is read once per process, so that form runs as a child): the same bytes in every file, FASTA and FASTQ
    (the qualities come from p1 / p2: every output of the call is covered)."""
    import shutil
    import subprocess
    import sys
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")) + glob.glob(os.path.join(GOLD, "fast5_more", "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    for i in range(60):
        shutil.copy(src[(i * 3) % len(src)], d / f"p{i:02d}.fast5")
    monkeypatch.setenv("NRV_CLI_GROUPS", "4")                           # ~4 reads per call: fifteen calls, ragged last groups
    outs = {}
    for tag, env in (("pipelined", "1"), ("one_at_a_time", "0")):
        monkeypatch.setenv("NRV_CLI_PIPELINE", env)
        out = str(tmp_path / tag) + "/"
        assert cli.main(["-d", str(d), "-o", out, "-S", "human", "-F", fmt, "--thread", "4", "--gpus", "1"]) == 0
        outs[tag] = {f: open(out + f, "rb").read() for f in sorted(os.listdir(out))}
    out = str(tmp_path / "staged") + "/"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "NanoReviser.py"), "-d", str(d), "-o", out, "-S", "human", "-F", fmt,
                        "--thread", "4", "--gpus", "1"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, NRV_RAW_STAGED="1", NRV_CLI_PIPELINE="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    outs["staged"] = {f: open(out + f, "rb").read() for f in sorted(os.listdir(out))}
    assert len(outs["pipelined"]) == 61 and outs["pipelined"]["failed_reads.txt"] == b""
    assert outs["pipelined"] == outs["one_at_a_time"] == outs["staged"]
    assert len({v for k, v in outs["pipelined"].items() if k.startswith("p0")}) >= 5   # (the reads do differ)


@pytest.mark.gpu
def test_script_with_parser_pool_is_quiet_and_complete(tmp_path):
    """`python NanoReviser.py` as a child process with a parser pool (>= 4 files, --thread 4): exit code 0, nothing on
    stderr (the script leaves with os._exit once main() has returned - the pool must have been shut down in order, or
    the resource tracker reports leaked semaphores), one output per read, identical to the in-process run's."""
    import shutil
    import subprocess
    import sys
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    names = [f"q{i}_{'AB'[i % 2]}" for i in range(8)]
    for n in names:
        shutil.copy(src[0] if n.endswith("A") else src[1], d / (n + ".fast5"))
    out = str(tmp_path) + "/o/"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "NanoReviser.py"), "-d", str(d), "-o", out, "-S", "ecoli",
                        "--thread", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    noise = [ln for ln in r.stderr.splitlines() if ln.strip() and "amdgpu.ids" not in ln]   # (libdrm's note on some boxes)
    assert noise == [], r.stderr
    ref = str(tmp_path) + "/ref/"
    assert cli.main(["-d", FAST5, "-o", ref, "-S", "ecoli", "--gpus", "1"]) == 0
    by_kind = {os.path.basename(src[0]).split(".")[0]: "A", os.path.basename(src[1]).split(".")[0]: "B"}
    seq = {k: open(ref + stem + "_out.fasta").read().split("\n", 1)[1] for stem, k in by_kind.items()}
    for n in names:
        assert open(out + n + "_out.fasta").read() == ">" + n + ".fast5\n" + seq[n[-1]]
    assert open(out + "failed_reads.txt").read() == ""


def test_output_directory_probe_warns_when_the_filesystem_is_too_slow(tmp_path, monkeypatch):
    """VERDICT r05 next #10a: `NanoReviser.py --gpus 8` warns when the output directory cannot take the files eight GPUs
    produce (r05: the sandbox's overlay root fed 68-78 M bases/s where tmpfs fed 120-131 M).  The probe leaves nothing
    behind, runs only for multi-GPU runs with thousands of reads (or NRV_OUTPUT_PROBE=1), and can be turned off."""
    out = str(tmp_path) + "/"
    args = cli.get_args(["-d", out, "-o", out, "-S", "ecoli"])
    rate = cli.probe_rename_rate(out, 200)
    assert rate and rate > 0 and os.listdir(out) == []
    msgs = []
    assert cli.check_output_rate(args, 1, 10 ** 6, msgs.append) is None          # one GPU: no probe
    assert cli.check_output_rate(args, 8, 100, msgs.append) is None              # a handful of reads: no probe
    monkeypatch.setattr(cli, "probe_rename_rate", lambda d, pairs=2000: 900.0)
    assert cli.check_output_rate(args, 8, 10 ** 5, msgs.append) == 900.0
    assert len(msgs) == 1 and "900 file creations" in msgs[0] and "8 GPU workers" in msgs[0]
    monkeypatch.setattr(cli, "probe_rename_rate", lambda d, pairs=2000: 1e6)
    assert cli.check_output_rate(args, 8, 10 ** 5, msgs.append) == 1e6 and len(msgs) == 1
    monkeypatch.setenv("NRV_OUTPUT_PROBE", "0")
    assert cli.check_output_rate(args, 8, 10 ** 5, msgs.append) is None


def test_device_calls_are_pipelined_two_deep_and_give_the_same_files(tmp_path, monkeypatch):
    """r06: with the native host stage the engine thread enqueues bundle k+1 (begin_packed_raw) before it collects bundle k
    (end_packed_raw).  Same files as one call at a time (NRV_CLI_PIPELINE=0), two calls in flight at most, collected in order,
    nothing else on the engine between a call's halves; a call that fails in its SECOND half is isolated read by read
    and only the failing read is written unrevised."""
    import shutil
    from nanoreviser_amd import hostlib
    from echo_engine import PipelinedEcho
    if hostlib.load() is None:
        pytest.skip("libnanorev_host.so not built")
    src = sorted(glob.glob(os.path.join(FAST5, "*.fast5")))
    d = tmp_path / "in"
    d.mkdir()
    for i in range(40):
        shutil.copy(src[i % 2], d / f"r{i:02d}.fast5")
    monkeypatch.setenv("NRV_CLI_GROUPS", "1")                           # small device calls: many bundles
    outs, engs = {}, {}
    for tag, env in (("pipelined", "1"), ("one_at_a_time", "0")):
        monkeypatch.setenv("NRV_CLI_PIPELINE", env)
        eng = engs[tag] = PipelinedEcho()
        out = str(tmp_path / tag) + "/"
        assert cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "3"], reviser_factory=lambda a, dev: eng) == 0
        outs[tag] = {f: open(out + f, "rb").read() for f in sorted(os.listdir(out))}
    assert outs["pipelined"] == outs["one_at_a_time"] and len(outs["pipelined"]) == 41
    assert engs["pipelined"].begun >= 5 and engs["pipelined"].max_in_flight == 2 and not engs["pipelined"].flight
    assert engs["one_at_a_time"].begun == 0 and engs["one_at_a_time"].calls >= 5
    assert not engs["pipelined"].violations, engs["pipelined"].violations
    # a bundle whose SECOND half fails: its reads are retried one by one, the marked read alone falls back
    _, _, rt0 = load_read("_".join(os.path.basename(src[0]).split("_")[-3:-1]))
    monkeypatch.setenv("NRV_CLI_PIPELINE", "1")
    eng = PipelinedEcho(fail_marker=rt0.feat_ev[0], fail_in_end=True)
    eng_probe = EchoEngine(fail_marker=rt0.feat_ev[0])                  # the per-read retry fails for that read too
    eng.predict_read = lambda s, f: (eng.violations.append("retry between halves") if eng.flight else None) or EchoEngine.predict_read(eng_probe, s, f)
    out = str(tmp_path / "fail") + "/"
    assert cli.main(["-d", str(d), "-o", out, "-S", "ecoli", "--thread", "3"], reviser_factory=lambda a, dev: eng) == 0
    failed = open(out + "failed_reads.txt").read().split()
    assert failed and all(int(f[1:3]) % 2 == 0 for f in failed)          # only copies of the marked read (src[0])
    assert len([f for f in os.listdir(out) if f.endswith("_out.fasta")]) == 40 and not eng.violations, eng.violations
