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
