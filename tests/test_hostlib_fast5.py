"""The host stage in C (libnanorev_host.so: csrc/nrv_host_fast5.c, include/nanorev_host.h) against the Python host stage
that defines it (h5lite + hoststage, themselves pinned to the reference's get_read_data / signal_segmentation on the
fixture reads: tests/test_hoststage_golden.py).  CPU only.

  * nrvh_load_fast5 / nrvh_load_bundle: samples, event starts, features (bit pattern), bases, shift, scale and the Fastq
    record of every fixture read - the two committed files and the other three of the reference's own five (tests/golden/fast5_more)
  * whatever the native reader does not know is DECLINED (never guessed): a broken file, a truncated file at any length,
    a missing basecall group; the command line then runs the Python path and words the error as the reference does
  * nrvh_finish_read: merge + record + file, byte for byte cli._finish_in_worker's (hence the reference's writers') on
    random calls, both formats, T = 11 / 13, no windows at all, and on the reference-run merge vectors
  * the command line gives the same files through threads + native stage as through processes + Python stage
"""
import glob
import json
import os
import shutil

import numpy as np
import pytest

from conftest import GOLD
from nanoreviser_amd import cli, hostlib
from nanoreviser_amd import hoststage as hs
from echo_engine import EchoEngine, PackedEcho

FAST5 = sorted(glob.glob(os.path.join(GOLD, "fast5", "*.fast5")))
MORE5 = sorted(glob.glob(os.path.join(GOLD, "fast5_more", "*.fast5")))            # the other three of the reference's fixture reads (r06)
G, SG = "Basecall_1D_000", "BaseCalled_template"

pytestmark = pytest.mark.skipif(hostlib.load() is None, reason="libnanorev_host.so not built")


def _python_path(path):
    rd, fq = cli.parse_read(path, G, SG)
    return hs.read_tensors_raw(rd), fq


def _same(o, rt, fq):
    return (np.array_equal(o["raw"], rt.raw) and np.array_equal(o["starts"], rt.starts)
            and np.array_equal(o["feat"].view(np.uint32), rt.feat_ev.view(np.uint32))
            and np.array_equal(o["bases"], rt.bases) and o["shift"] == rt.shift and o["scale"] == rt.scale
            and o["fastq"] == fq)


@pytest.mark.parametrize("path", FAST5 + MORE5)
def test_native_reader_gives_the_python_host_stage_bit_for_bit(path):
    rc, o = hostlib.load_fast5(path, G, SG, True)
    assert rc == hostlib.OK, o
    rt, fq = _python_path(path)
    assert o["raw"].dtype == np.int16 and o["starts"].dtype == np.int32 and o["feat"].dtype == np.float32
    assert _same(o, rt, fq)
    rc, o2 = hostlib.load_fast5(path, G, SG, False)                   # the Fastq record only when asked for
    assert rc == hostlib.OK and o2["fastq"] is None and np.array_equal(o2["feat"], o["feat"])


def test_bundle_concatenates_the_good_reads_and_reports_the_others(tmp_path):
    bad = tmp_path / "broken.fast5"
    bad.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\x00" * 64)
    missing = str(tmp_path / "nothing_here.fast5")
    paths = [FAST5[0], str(bad), FAST5[1], missing, FAST5[0]]
    b = hostlib.load_bundle(paths, G, SG, True)
    assert list(b["status"]) == [hostlib.OK, hostlib.UNSUPPORTED, hostlib.OK, hostlib.E_IO, hostlib.OK]
    assert b["errors"][1] and b["errors"][3] and b["fastq"][1] is None
    ro = eo = 0
    for i, p in enumerate(paths):
        if b["status"][i] != hostlib.OK:
            assert not b["meta"][i].any()
            continue
        rt, fq = _python_path(p)
        rl, el, sh, sc = b["meta"][i]
        rl, el = int(rl), int(el)
        one = {"raw": b["raw"][ro:ro + rl], "starts": b["starts"][eo:eo + el], "feat": b["feat"][eo:eo + el],
               "bases": b["bases"][eo:eo + el], "shift": sh, "scale": sc, "fastq": b["fastq"][i]}
        assert _same(one, rt, fq), p
        ro, eo = ro + rl, eo + el
    assert ro == len(b["raw"]) and eo == len(b["starts"]) == len(b["feat"]) == len(b["bases"])


def test_wrong_group_and_truncated_files_are_declined_never_guessed(tmp_path):
    rc, why = hostlib.load_fast5(FAST5[0], "Basecall_1D_007", SG)
    assert rc == hostlib.UNSUPPORTED and why                           # the Python path raises the reference's message
    with pytest.raises(RuntimeError, match="No events or corrupted events"):
        cli.parse_read(FAST5[0], "Basecall_1D_007", SG)
    data = open(FAST5[0], "rb").read()
    rt, fq = _python_path(FAST5[0])
    rng = np.random.default_rng(5)
    cuts = sorted({8, 96, 2048, len(data) // 2, len(data) - 1} | {int(x) for x in rng.integers(9, len(data) - 1, 40)})
    t = tmp_path / "cut.fast5"
    for n in cuts:                                                     # any prefix of the file: a clean refusal or the right answer
        t.write_bytes(data[:n])
        rc, o = hostlib.load_fast5(str(t), G, SG, True)
        assert rc != hostlib.OK or _same(o, rt, fq), n
    flip = bytearray(data)                                             # and a few corrupted bytes in the metadata region
    for pos in rng.integers(8, 4096, 24):
        flip[int(pos)] ^= 0xFF
    t.write_bytes(bytes(flip))
    rc, o = hostlib.load_fast5(str(t), G, SG, True)
    assert rc in (hostlib.OK, hostlib.UNSUPPORTED, hostlib.E_READ)


def _member(data, name):
    """Offset of a version-1 compound member record `name` (name padded to 8, byte offset, 28 bytes of dimension fields,
    then the member's datatype: class/version, 3 flag bytes, size)."""
    key = name + b"\x00" * (8 - len(name) % 8 if len(name) % 8 else 8)
    i = data.find(key)
    assert i > 0 and data.find(key, i + 1) < 0, name
    return i + len(key)


def test_compound_datatype_from_the_file_is_checked_not_trusted(tmp_path):
    """member offsets and sizes of the Events compound come from the file.  A field outside its row,
    an integer wider than 8 bytes or of an odd width, a row of no bytes: each is DECLINED (the Python stage then words
    the error), never read out of bounds.  The mutations run in this process: a crash would end the test session loudly."""
    data = open(FAST5[0], "rb").read()
    rt, fq = _python_path(FAST5[0])
    t = tmp_path / "mut.fast5"

    def load(mut):
        t.write_bytes(bytes(mut))
        return hostlib.load_fast5(str(t), G, SG, True)

    rc, o = load(bytearray(data))
    assert rc == hostlib.OK and _same(o, rt, fq)
    mv, st, ms, mean = (_member(data, n) for n in (b"move", b"start", b"model_state", b"mean"))
    esize = int.from_bytes(data[mean - 8 - 4:mean - 8], "little")             # the compound's size: in front of its first member
    assert esize == 41 and int.from_bytes(data[mv:mv + 4], "little") == 29, esize   # mean start stdv length model_state MOVE p_model_state weights
    cases = []
    for off in (0x7FFFFF00, 0xFFFFFFF0, esize - 3, esize):                       # the field leaves its row
        m = bytearray(data); m[mv:mv + 4] = off.to_bytes(4, "little"); cases.append(("move offset %#x" % off, m))
    for size in (64, 16, 9, 3, 0, 0x80000000):                                  # not an integer load_int can hold
        m = bytearray(data); m[mv + 32 + 4:mv + 32 + 8] = size.to_bytes(4, "little"); cases.append(("move size %d" % size, m))
        m = bytearray(data); m[st + 32 + 4:st + 32 + 8] = size.to_bytes(4, "little"); cases.append(("start size %d" % size, m))
    m = bytearray(data); m[ms:ms + 4] = (esize - 2).to_bytes(4, "little"); cases.append(("model_state runs out of the row", m))
    for es in (0, 1, 8):                                                         # rows shorter than their fields, or of no bytes
        m = bytearray(data); m[mean - 12:mean - 8] = es.to_bytes(4, "little"); cases.append(("row of %d bytes" % es, m))
    for what, m in cases:
        rc, o = load(m)
        assert rc == hostlib.UNSUPPORTED, (what, rc)
    # a field moved to another place INSIDE the row is not an error the reader can see: it must simply not crash
    m = bytearray(data); m[mv:mv + 4] = (0).to_bytes(4, "little")
    assert load(m)[0] in (hostlib.OK, hostlib.UNSUPPORTED, hostlib.E_READ)


def test_byte_flip_fuzz_many_seeds_in_a_child_process(tmp_path):
    """Random corruption of the metadata region and of whole-file positions, 60 seeds x every fixture file: every outcome is a return
    code.  Run in a CHILD so that a crash of the native reader is this test's failure, not the end of the session."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from nanoreviser_amd import hostlib\n"
        "n = 0\n"
        "for path in %r:\n"
        "    data = open(path, 'rb').read()\n"
        "    for seed in range(60):\n"
        "        rng = np.random.default_rng(seed)\n"
        "        m = bytearray(data)\n"
        "        hi = 4096 if seed %% 3 else len(data)\n"
        "        for pos in rng.integers(8, hi, int(rng.integers(1, 32))):\n"
        "            m[int(pos)] = int(rng.integers(0, 256)) if seed %% 2 else m[int(pos)] ^ 0xFF\n"
        "        if seed %% 5 == 0:\n"                                          # and the datatype region in particular
        "            i = data.find(b'model_state')\n"
        "            for pos in rng.integers(i - 300, i + 300, 6):\n"
        "                m[int(pos)] = int(rng.integers(0, 256))\n"
        "        open(%r, 'wb').write(bytes(m))\n"
        "        rc, o = hostlib.load_fast5(%r, %r, %r, True)\n"
        "        assert rc in (hostlib.OK, hostlib.UNSUPPORTED, hostlib.E_READ, hostlib.E_IO), rc\n"
        "        n += 1\n"
        "print('fuzzed', n)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), FAST5, str(tmp_path / "f.fast5"), str(tmp_path / "f.fast5"), G, SG)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and f"fuzzed {60 * len(FAST5)}" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-2000:])


def test_two_reads_with_one_destination_never_share_a_temporary(tmp_path):
    """`a.x.fast5` and `a.y.fast5` both map to `a_out.fasta`; finisher THREADS of one process must not
    write through one temporary.  32 threads x 40 writes to one destination: the file is always one whole record."""
    from concurrent.futures import ThreadPoolExecutor
    dst = str(tmp_path / "a_out.fasta")
    rng = np.random.default_rng(1)
    recs = []
    for k in range(8):
        n = 2000 + 300 * k
        bases = np.array(list("ACGT"), dtype="S1")[rng.integers(0, 4, n + 12)]
        recs.append((bases, np.full(n, 5 - (k % 4), np.int8), np.full(n, 4 - (k % 4), np.int8), f"read{k}"))
    want = set()
    for b, a1, a2, nm in recs:
        hostlib.finish_read(b, a1, a2, 11, None, nm, dst, False)
        want.add(open(dst, "rb").read())

    def work(i):
        b, a1, a2, nm = recs[i % 8]
        return hostlib.finish_read(b, a1, a2, 11, None, nm, dst, False)
    with ThreadPoolExecutor(32) as ex:
        res = list(ex.map(work, range(32 * 40)))
    assert all(r is not None and r > 0 for r in res)
    assert open(dst, "rb").read() in want
    assert not [f for f in os.listdir(tmp_path) if ".tmp" in f]


class _Spec:
    def __init__(self, d, fmt):
        self.output_dir, self.output_format = d, fmt


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
@pytest.mark.parametrize("T", [11, 13])
def test_native_finisher_writes_the_python_finishers_bytes(tmp_path, fmt, T):
    rd, _ = cli.parse_read(FAST5[0], G, SG)
    rng = np.random.default_rng(T)
    N = len(rd.bases)
    for n in (N - T, 100, 1, 0):
        bases = rd.bases if n == N - T else rd.bases[:max(n + T, 5)]
        a1, a2 = rng.integers(0, 6, n).astype(np.int8), rng.integers(0, 5, n).astype(np.int8)
        qc = rng.integers(34, 74, n).astype(np.uint8) if fmt == "fastq" and n else None
        d = str(tmp_path / f"{n}") + "/"
        fn = "a read of mine_ch7.fast5"
        nb, err = cli._finish_in_worker(_Spec(d, fmt), T, fn, bases, a1, a2, qc)
        assert err is None
        want = open(cli.out_name(d, fn, fmt), "rb").read()
        os.remove(cli.out_name(d, fn, fmt))
        nb2, err2 = cli._finish_native(_Spec(d, fmt), T, fn, bases, a1, a2, qc)
        assert err2 is None and nb2 == nb
        assert open(cli.out_name(d, fn, fmt), "rb").read() == want
        assert not [f for f in os.listdir(d) if ".tmp" in f]           # the temporary was renamed into place


def test_native_finisher_on_the_reference_run_merge_vectors(tmp_path):
    """tests/golden/merge_vectors.json: outputs of the reference's own get_base_1; revise_read reproduces them modulo the
    documented decode fix, and the native merge is revise_read (T = 1: window i revises base i)."""
    vecs = json.load(open(os.path.join(GOLD, "merge_vectors.json")))["get_base_1"]
    n_checked = 0
    for v in vecs:
        bases = np.array(list(v["event_bases"]), dtype="S1")
        a1 = np.array(v["y_pre"], dtype=np.int8)
        cls2 = np.array(v["y_pre2"], dtype=np.int64) - 2               # get_base_1 takes argmax2 + 2
        if len(a1) != len(bases) or len(a1) == 0 or cls2.min() < -1 or cls2.max() > 4 or a1.min() < 0 or a1.max() > 5:
            continue
        a2 = cls2.astype(np.int8)
        dst = str(tmp_path / "v.fasta")
        hostlib.finish_read(bases, a1, a2, 1, None, "v", dst, False)
        got = open(dst).read()
        assert got == ">v\n" + hs.revise_read(bases, a1, a2, 1)
        # the reference's own result = its seed character (output_handeler.py:107; '-' is filtered) + the merge
        seed = hs.LABEL_TO_BASE[int(a1[0])]
        assert v["result"] == ("" if seed == "-" else seed) + got[3:], v
        n_checked += 1
    assert n_checked >= 20
    rng = np.random.default_rng(0)                                      # and the same identity on random calls
    for _ in range(50):
        n = int(rng.integers(1, 400))
        bases = np.array(list("ACGT"), dtype="S1")[rng.integers(0, 4, n + 12)]
        a1, a2 = rng.integers(0, 6, n).astype(np.int8), rng.integers(0, 5, n).astype(np.int8)
        dst = str(tmp_path / "r.fasta")
        hostlib.finish_read(bases, a1, a2, 11, None, "r", dst, False)
        assert open(dst).read() == ">r\n" + hs.revise_read(bases, a1, a2, 11)


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
def test_cli_threads_native_equals_processes_python(tmp_path, monkeypatch, fmt):
    d = tmp_path / "in"
    d.mkdir()
    for i in range(12):
        shutil.copy(FAST5[i % 2], d / f"r{i:02d}.fast5")
    (d / "broken.fast5").write_bytes(b"\x89HDF\r\n\x1a\n" + b"\x00" * 64)
    outs = {}
    for tag, env in (("native", {}), ("python", {"NRV_HOST_LIB": "0"}), ("native_procs", {"NRV_HOST_THREADS": "0"})):
        for k in ("NRV_HOST_LIB", "NRV_HOST_THREADS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setattr(hostlib, "_tried", False)
        monkeypatch.setattr(hostlib, "_lib", None)
        for eng_tag, eng in (("", EchoEngine), ("_packed", PackedEcho)):
            out = str(tmp_path / (tag + eng_tag)) + "/"
            assert cli.main(["-d", str(d), "-o", out, "-F", fmt, "-S", "ecoli", "--thread", "3"],
                            reviser_factory=lambda a, dev: eng()) == 0
            outs[tag + eng_tag] = {f: open(out + f, "rb").read() for f in sorted(os.listdir(out))}
    monkeypatch.setattr(hostlib, "_tried", False)
    monkeypatch.setattr(hostlib, "_lib", None)
    assert len(outs["native"]) == 13 and outs["native"]["failed_reads.txt"].split() == [b"broken.fast5"]
    # the echo engines return every read unchanged: whatever the pool (threads / processes), the reader (C / Python) and
    # the finisher (per read, per device call in one C call, Python), the files are the same bytes
    for k, v in outs.items():
        assert v == outs["native"], k
    assert len(outs) == 6
