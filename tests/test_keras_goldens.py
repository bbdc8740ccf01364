"""Consumer of tests/golden/keras_goldens.npz - outputs of the REFERENCE's own arithmetic (keras
2.2.4 / tensorflow 1.12 running get_model1/get_model2 with the shipped weights), written by
tools/make_keras_goldens.py in an environment that has Keras.  The build image has none (SURVEY.md
F5), so until somebody runs the generator these tests SKIP and the model-graph oracle stays
"parity unpinned" (DESIGN.md 5); with the file present they are the pin: the oracle (CPU test) and
the HIP path (GPU test) against Keras itself, at north_star's bars - argmax identical, |dp| <= 1e-4."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT
from nanoreviser_amd import hoststage as hs

PATH = os.path.join(GOLD, "keras_goldens.npz")
need_file = pytest.mark.skipif(not os.path.exists(PATH),
                               reason="tests/golden/keras_goldens.npz absent: run tools/make_keras_goldens.py where "
                                      "keras 2.2.4 + tensorflow 1.12 exist (parity stays unpinned until then)")


def _windows(reads, kg, key):
    if key == "synth11":
        mg = np.load(os.path.join(GOLD, "model_goldens.npz"))
        return mg["synth11/signal"], mg["synth11/read"]
    _, _, rt = reads(key)
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
    idx = kg[f"{key}/idx"]
    return np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx])


def test_generator_is_guarded_without_keras():
    """The generator must never write a file it cannot vouch for: without Keras it exits 3."""
    if importlib.util.find_spec("keras") is not None:
        pytest.skip("keras is importable here: run tools/make_keras_goldens.py instead")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_keras_goldens.py"), "--out",
                        os.path.join(ROOT, "tests", "_never_written.npz")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "not importable" in r.stderr
    assert not os.path.exists(os.path.join(ROOT, "tests", "_never_written.npz"))


@need_file
@pytest.mark.parametrize("sp", ["ecoli", "human"])
def test_oracle_matches_keras(reads, species_models, sp):
    from oracle import nrv_oracle as O
    kg = np.load(PATH)
    assert str(kg["meta/keras_version"]).startswith("2.2"), "goldens must come from the pinned keras 2.2.x"
    m1, m2 = species_models[sp]
    for key in list(reads.keys) + ["synth11"]:
        sig, rd = _windows(reads, kg, key)
        for dt, tol in ((np.float64, 1e-4), (np.float32, 2e-4)):        # f32 vs f32: two rounding-noise floors
            p1, p2, a1, a2 = O.predict_pair(m1.tensors, m2.tensors, sig, rd, dt)
            assert np.abs(p1 - kg[f"{key}/{sp}/p1"]).max() <= tol, (key, dt)
            assert np.abs(p2 - kg[f"{key}/{sp}/p2"]).max() <= tol, (key, dt)
        assert np.array_equal(a1, kg[f"{key}/{sp}/a1"]) and np.array_equal(a2, kg[f"{key}/{sp}/a2"]), key


@need_file
@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f16x2", "bf16x3", "f32"])
@pytest.mark.parametrize("sp", ["ecoli", "human"])
def test_hip_matches_keras(reads, species_models, sp, prec):
    from nanoreviser_amd.engine import Reviser
    kg = np.load(PATH)
    rv = Reviser(*species_models[sp], precision=prec)
    for key in list(reads.keys) + ["synth11"]:
        sig, rd = _windows(reads, kg, key)
        p1, p2, a1, a2 = rv.predict_pair(sig, rd)
        assert np.abs(p1 - kg[f"{key}/{sp}/p1"]).max() <= 2e-4 and np.abs(p2 - kg[f"{key}/{sp}/p2"]).max() <= 2e-4
        assert np.array_equal(a1, kg[f"{key}/{sp}/a1"]) and np.array_equal(a2, kg[f"{key}/{sp}/a2"]), key
    rv.close()
