"""The oracle against itself and the committed goldens (CPU).  The oracle is test infrastructure;
nothing here touches the HIP path."""
import numpy as np
import pytest

from nanoreviser_amd import hoststage as hs
from nanoreviser_amd import weights as W
from oracle import c_oracle as CO
from oracle import nrv_oracle as O


def _windows(reads, key, T, idx):
    _, _, rt = reads(key)
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, T)
    return np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx])


def test_weight_files(species_models):
    # shapes of SURVEY.md 8a; the shipped files are T=11 despite the win13 names (F3)
    for sp, (m1, m2) in species_models.items():
        assert (m1.T, m1.n_class, m2.T, m2.n_class) == (11, 6, 11, 5)
        assert m1.flat().size == 595300 and m2.flat().size == 595283
        assert W.infer_T_and_classes(595300) == (11, 6) and W.infer_T_and_classes(595283) == (11, 5)
        assert m1["feature.kernel"].shape == (66, 16)
        assert m1["lstm3.fw.kernel"].shape == (192, 512)
        m13 = m1.with_window(13)
        assert m13["feature.kernel"].shape == (78, 16) and m13.flat().size == W.n_params(13, 6)
        # only tensor 56 differs
        assert all(np.array_equal(a, b) for i, (a, b) in enumerate(zip(m1.tensors, m13.tensors)) if i != 56)
    with pytest.raises(ValueError):
        W.infer_T_and_classes(12345)


def test_numpy_fp64_matches_committed_goldens(reads, model_goldens, species_models):
    key = reads.keys[0]
    idx = model_goldens[f"{key}/idx"][:96]
    sw, fw = _windows(reads, key, 11, idx)
    for sp in ("ecoli", "human"):
        m1, m2 = species_models[sp]
        p1, p2, a1, a2 = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float64)
        assert np.abs(p1 - model_goldens[f"{key}/{sp}/p1"][:96]).max() < 1e-12
        assert np.abs(p2 - model_goldens[f"{key}/{sp}/p2"][:96]).max() < 1e-12
        assert np.array_equal(a1, model_goldens[f"{key}/{sp}/a1"][:96])
        assert np.array_equal(a2, model_goldens[f"{key}/{sp}/a2"][:96])


@pytest.mark.parametrize("sp", ["ecoli", "human"])
def test_fp32_restatements_agree_with_fp64(reads, model_goldens, species_models, sp):
    """NumPy-f32 and C-f32 vs the fp64 arbiter on real windows of every fixture read.
    Tolerance on probabilities: 1e-4 for E. coli.  The human model2 is ill-conditioned on a few
    windows: BOTH independent f32 restatements sit 2.1e-4 from fp64 there (same figure as SURVEY.md
    App. B), so 1e-4 is below that model's own fp32 noise floor; 5e-4 is used for human.
    argmax identical everywhere."""
    m1, m2 = species_models[sp]
    tol = 1e-4 if sp == "ecoli" else 5e-4
    for key in reads.keys:
        idx = model_goldens[f"{key}/idx"][200:328]           # 56 from the head + 72 from the middle
        
        sw, fw = _windows(reads, key, 11, idx)
        g1, g2 = model_goldens[f"{key}/{sp}/p1"][200:328], model_goldens[f"{key}/{sp}/p2"][200:328]
        b1, b2 = model_goldens[f"{key}/{sp}/a1"][200:328], model_goldens[f"{key}/{sp}/a2"][200:328]
        p1, p2, a1, a2 = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float32)
        assert np.abs(p1 - g1).max() < tol and np.abs(p2 - g2).max() < tol
        assert np.array_equal(a1, b1) and np.array_equal(a2, b2)
        c1, ca1 = CO.predict(m1.flat(), 11, 6, sw, fw, threads=4)
        c2, ca2 = CO.predict(m2.flat(), 11, 5, sw, fw, threads=4)
        assert np.abs(c1 - g1).max() < tol and np.abs(c2 - g2).max() < tol
        assert np.array_equal(ca1, b1) and np.array_equal(ca2, b2)


def test_synthetic_goldens_T11_T13(model_goldens, species_models):
    for T in (11, 13):
        sig, rd = model_goldens[f"synth{T}/signal"][:64], model_goldens[f"synth{T}/read"][:64]
        for sp in ("ecoli", "human"):
            m1, m2 = species_models[sp]
            a, b = m1.with_window(T), m2.with_window(T)
            p1, _ = CO.predict(a.flat(), T, 6, sig, rd, threads=4)
            p2, _ = CO.predict(b.flat(), T, 5, sig, rd, threads=4)
            assert np.abs(p1 - model_goldens[f"synth{T}/{sp}/p1"][:64]).max() < 2e-4
            assert np.abs(p2 - model_goldens[f"synth{T}/{sp}/p2"][:64]).max() < 2e-4
    # the generator is deterministic for a given numpy; shapes/ranges as SURVEY.md 8d C4
    s, r = O.synth_windows(32, 13, seed=1)
    assert s.shape == (32, 13, 50) and r.shape == (32, 13, 6) and s.dtype == np.float32
    assert s.min() >= -8.4 and s.max() <= 4.8 and r[..., 3].max() <= 46.5


def test_c_oracle_read_mode_equals_window_mode(reads, species_models):
    _, _, rt = reads(reads.keys[2])
    m1, m2 = species_models["ecoli"]
    N = 80
    sw, fw = hs.sliding_windows(rt.sig_ev[:N], rt.feat_ev[:N], 11)
    pw, aw = CO.predict(m1.flat(), 11, 6, np.ascontiguousarray(sw), np.ascontiguousarray(fw), threads=4)
    pr, ar = CO.predict_read(m1.flat(), 11, 6, rt.sig_ev[:N], rt.feat_ev[:N], threads=4)
    assert pr.shape == (N - 11, 6) and np.array_equal(pw, pr) and np.array_equal(aw, ar)
    p0, a0 = CO.predict_read(m1.flat(), 11, 6, rt.sig_ev[:11], rt.feat_ev[:11])
    assert p0.shape == (0, 6) and a0.shape == (0,)
    with pytest.raises(RuntimeError):
        CO.predict(m1.flat()[:-1], 11, 6, sw[:1], fw[:1])


def test_empty_batch():
    from nanoreviser_amd.weights import load_species
    m1, m2 = load_species("ecoli")
    p1, p2, a1, a2 = O.predict_pair(m1.tensors, m2.tensors, np.zeros((0, 11, 50), np.float32),
                                    np.zeros((0, 11, 6), np.float32))
    assert p1.shape == (0, 6) and p2.shape == (0, 5) and a1.shape == (0,)
