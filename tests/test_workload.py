"""The product's workload generators (nanoreviser_amd/workload.py) against the oracle-side copy the
committed goldens were made with."""
import numpy as np

from nanoreviser_amd import workload as W


def test_synth_windows_is_the_golden_generator(model_goldens):
    from oracle import nrv_oracle as O
    for T, n in ((11, 40), (13, 17)):
        a, b = W.synth_windows(n, T, seed=7)
        c, d = O.synth_windows(n, T, seed=7)
        assert np.array_equal(a, c) and np.array_equal(b, d)
        assert a.shape == (n, T, 50) and b.shape == (n, T, 6) and a.dtype == b.dtype == np.float32


def test_synth_read_shapes_and_ranges():
    s, f = W.synth_read(5000, seed=1)
    assert s.shape == (5000, 50) and f.shape == (5000, 6)
    assert s.min() >= -8.4 - 1e-6 and s.max() <= 4.8 + 1e-6
    assert set(np.unique(np.round(f[:, 0] * 300))) <= {30, 100, 180, 250}
    assert (f[:, 3] <= 46.5).all() and (f[:, 2] >= 0).all()


def test_facade_fingerprint_sees_in_place_refills():
    """engine.Reviser._fingerprint guards the pair cache of the Keras-shaped facade (model1.predict then
    model2.predict on the same arrays): same contents -> same key, an in-place refill -> another key."""
    import numpy as np
    from nanoreviser_amd.engine import Reviser
    rng = np.random.default_rng(0)
    small = rng.normal(size=(300, 11, 50)).astype(np.float32)
    k0 = Reviser._fingerprint(small)
    assert Reviser._fingerprint(small) == k0
    small[137, 5, 49] += 1.0                       # small arrays are hashed whole: any element counts
    assert Reviser._fingerprint(small) != k0
    big = rng.normal(size=(6000, 11, 50)).astype(np.float32)      # 13 MB: sampled lines
    k1 = Reviser._fingerprint(big)
    assert Reviser._fingerprint(big) == k1
    big[...] = rng.normal(size=big.shape).astype(np.float32)
    assert Reviser._fingerprint(big) != k1
    assert Reviser._fingerprint(big.copy())[1:] == Reviser._fingerprint(big)[1:]     # only id() differs
