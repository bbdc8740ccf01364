"""The microbenchmarks the design argues from (tools/microbench/*.hip) must keep compiling for gfx950: hipcc cross-compiles
without a GPU.  They are run on the GPU box by hand (their headers say how); their recorded outputs are profiles/r04_tick_cost.txt,
profiles/r04_clock_vs_fill.txt and - every wave stamped, which corrected r04's reading - profiles/r06_clock_vs_fill.txt."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", sorted(glob.glob(os.path.join(ROOT, "tools", "microbench", "*.hip"))), ids=os.path.basename)
def test_microbenchmark_compiles_for_gfx950(src, tmp_path):
    out = tmp_path / "mb"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-o", str(out), src],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists() and out.stat().st_size > 10000
