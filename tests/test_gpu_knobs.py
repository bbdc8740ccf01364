"""The alternative kernels of the EXPERIMENTS build (tools/lstm_exp.sh knobs -> csrc/exp/
libnanorev_hip_experiments.so, -DNRV_EXPERIMENTS; the product library ships none of them) stay parity-green
behind their environment knobs: each knob is read once per process, so every case runs in its
own interpreter (GPU box only; skipped when the experiments library has not been built)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(ROOT, "nanoreviser_amd", "csrc", "exp", "libnanorev_hip_experiments.so")

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from conftest import load_read
from nanoreviser_amd import hoststage as hs
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
mg = np.load(os.path.join({root!r}, "tests", "golden", "model_goldens.npz"))
out = {{}}
for sp in ("ecoli", "human"):
    m1, m2 = load_species(sp)
    rv = Reviser(m1, m2, precision={prec!r})
    key = "ch10_read5252"
    _, _, rt = load_read(key)
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
    idx = mg[key + "/idx"]
    p1, p2, a1, a2 = rv.predict_pair(np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx]))
    out[sp] = dict(dp1=float(np.abs(p1 - mg[f"{{key}}/{{sp}}/p1"]).max()), dp2=float(np.abs(p2 - mg[f"{{key}}/{{sp}}/p2"]).max()),
                   flips=int((a1 != mg[f"{{key}}/{{sp}}/a1"]).sum() + (a2 != mg[f"{{key}}/{{sp}}/a2"]).sum()))
    rv.close()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(EXP_LIB), reason="experiments library not built (tools/lstm_exp.sh knobs)")
@pytest.mark.parametrize("knob,prec", [("NRV_L2T=0;NRV_MFMA16=0", "f16x2"), ("NRV_L2T=0;NRV_MFMA16=0;NRV_HT=0", "f16x2"),
                                       ("NRV_MFMA16=1", "f16x2"), ("NRV_L2T=0;NRV_MFMA16=2", "f16x2"), ("NRV_L2T=0", "f16x2"),
                                       ("NRV_CNN=h2", "f16x2"), ("NRV_CNN=m", "f16x2"),
                                       ("NRV_PAIR=0", "bf16x3"), ("NRV_GEO=-1,0,1,0", "f32")])
def test_alternative_kernels_match_the_goldens(knob, prec):
    env = dict(os.environ, NRV_LIB=EXP_LIB, **dict(kv.split("=") for kv in knob.split(";")))
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, prec=prec)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    for sp, d in res.items():
        assert d["flips"] == 0, (knob, sp, d)
        assert d["dp1"] <= 1e-4 and d["dp2"] <= 1e-4, (knob, sp, d)
