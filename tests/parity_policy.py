"""The parity policy of tests/test_gpu_parity.py (its module docstring states it), shared by the GPU test
modules: the f32 noise floor of the oracle's own f32 implementations and the per-window check against the
fp64 arbiter.  Test infrastructure only."""
import numpy as np

BAR = 1e-4


def f32_floor(m1, m2, sig, rd, p64_1, p64_2, T=11, numpy_too=True):
    """Per-window deviation of the oracle's OWN f32 implementations from the fp64 arbiter:
    (nf1[n], nf2[n]) = max over classes and over {C port, NumPy-f32}."""
    from oracle import c_oracle as CO
    from oracle import nrv_oracle as O
    c1, _ = CO.predict(m1.flat(), T, 6, sig, rd, threads=8)
    c2, _ = CO.predict(m2.flat(), T, 5, sig, rd, threads=8)
    nf1, nf2 = np.abs(c1 - p64_1).max(-1), np.abs(c2 - p64_2).max(-1)
    if numpy_too:
        q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sig, rd, np.float32)
        nf1 = np.maximum(nf1, np.abs(q1 - p64_1).max(-1))
        nf2 = np.maximum(nf2, np.abs(q2 - p64_2).max(-1))
    return nf1, nf2


def check_vs_fp64(p, a, p64, nf, what="", max_ill=0.01, bar=BAR):
    """The policy of the module docstring for one model's outputs.  Returns a dict of what was seen."""
    p, a, p64 = np.asarray(p), np.asarray(a), np.asarray(p64)
    err = np.abs(p - p64).max(-1)
    well = nf <= bar / 2
    assert (err[well] <= bar).all(), f"{what}: well-conditioned window off by {err[well].max():.2e}"
    ill = ~well
    if ill.any():
        assert (err[ill] <= 3 * nf[ill]).all(), \
            f"{what}: ill-conditioned window off by {err[ill].max():.2e} (f32 floor there {nf[ill].max():.2e})"
    assert ill.mean() <= max_ill, f"{what}: {ill.sum()} of {len(ill)} windows ill-conditioned"
    ref = p64.argmax(-1)
    bad = np.nonzero(a != ref)[0]
    srt = np.sort(p64, -1)
    margin = srt[:, -1] - srt[:, -2]
    for i in bad:
        assert margin[i] <= 2 * max(nf[i], 1e-6), \
            f"{what}: window {i}: engine class {int(a[i])} vs {ref[i]}, fp64 margin {margin[i]:.2e}, f32 floor {nf[i]:.2e}"
    return {"max_err": float(err.max()), "max_err_well": float(err[well].max()) if well.any() else 0.0,
            "ill": int(ill.sum()), "near_ties": int(len(bad)), "n": int(len(err)),
            "max_f32_floor": float(nf.max())}
