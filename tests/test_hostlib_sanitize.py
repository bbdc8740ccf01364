"""Sanitizer gate of the native host stage.  CPU only.

csrc/nrv_host_fast5.c is a hand-written HDF5 subset reader that parses untrusted fast5 files as THREADS of the GPU worker
(the reference leaves this to h5py / libhdf5: nanorevutils/nanorev_fast5_handeler.py:39-150).  scripts/host_sanitize.sh
builds tools/hostfuzz/host_fuzz.c - which includes the two C sources - with AddressSanitizer + UndefinedBehaviorSanitizer
(-fno-sanitize-recover=all) and with ThreadSanitizer, and runs
  * the argument edge cases of every exported entry point,
  * 1000 structure-aware mutations per fixture file (all five of the reference's test reads): metadata flips, extreme
    2/4/8-byte values inside header messages, B-tree / heap nodes (cycles included), the Events compound type, ROWS of the
    Events table (the `start` values that overflowed a signed difference in round 5: nrv_host_fast5.c:576), truncations,
    chunk keys - through the image parser, the file + bundle entry points and the finishers,
  * 8 threads loading bundles and finishing reads concurrently under ThreadSanitizer.
Any report aborts the driver; the test wants exit code 0 and no report text."""
import glob
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST5 = sorted(glob.glob(os.path.join(GOLD, "fast5", "*.fast5")) + glob.glob(os.path.join(GOLD, "fast5_more", "*.fast5")))
REPORTS = ("runtime error", "ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "WARNING: ThreadSanitizer", "SANITIZE FAIL")


def _have_sanitizers():
    if shutil.which("gcc") is None:
        return False
    r = subprocess.run(["gcc", "-fsanitize=address,undefined", "-x", "c", "-", "-o", os.devnull], input="int main(void){return 0;}",
                       capture_output=True, text=True)
    return r.returncode == 0


@pytest.mark.skipif(not _have_sanitizers(), reason="gcc with libasan / libubsan is not in this image")
def test_native_host_stage_is_clean_under_asan_ubsan_tsan():
    assert len(FAST5) == 5
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "host_sanitize.sh"), "1000", "1"], capture_output=True, text=True,
                       timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "api ok" in out and "fuzzed 5000" in out and "threads ok: 8 x 6" in out and "SANITIZE OK" in out, out[-2000:]
    for word in REPORTS:
        assert word not in out, out[-4000:]


def test_event_starts_no_daq_produces_are_declined(tmp_path):
    """The round-5 finding by hand: a `start` of INT64_MIN / INT64_MAX / 2^40 + 1 in a row that counts (move != 0) made
    `st64[i+1] - st64[i]` overflow.  The native reader now declines such a file before any difference is formed; the Python
    host stage (the definition) then words the error."""
    from nanoreviser_amd import hostlib
    if hostlib.load() is None:
        pytest.skip("libnanorev_host.so not built")
    data = bytearray(open(FAST5[0], "rb").read())
    from nanoreviser_amd import h5lite
    ev = h5lite.File(FAST5[0])["Analyses/Basecall_1D_000/BaseCalled_template/Events"].read()
    row = int(np.flatnonzero(ev["move"] != 0)[7])
    first = data.find(ev[:2].tobytes())                       # the table is stored contiguously in these files
    assert first > 0
    off = first + row * ev.dtype.itemsize + ev.dtype.fields["start"][1]
    assert int.from_bytes(data[off:off + 8], "little") == int(ev["start"][row])
    t = tmp_path / "wild.fast5"
    for v in (1 << 63, (1 << 63) + 0xc04, (1 << 63) - 1, (1 << 64) - (1 << 41), (1 << 40) + 1):
        m = bytearray(data)
        m[off:off + 8] = v.to_bytes(8, "little")
        t.write_bytes(bytes(m))
        rc, why = hostlib.load_fast5(str(t), "Basecall_1D_000", "BaseCalled_template", True)
        assert rc == hostlib.UNSUPPORTED and "range" in why, (hex(v), rc, why)
    m = bytearray(data)                                        # 2^31 past the first event: int32 starts cannot hold it
    m[off:off + 8] = (int(ev["start"][0]) + (1 << 31) + 5).to_bytes(8, "little")
    t.write_bytes(bytes(m))
    rc, why = hostlib.load_fast5(str(t), "Basecall_1D_000", "BaseCalled_template", True)
    assert rc != hostlib.OK
