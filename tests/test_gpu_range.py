"""Out-of-range input in every arithmetic mode (MI355X only, -m gpu).

The reference normalises samples as (raw - median) / MAD with no clipping (preprocessing.py:120-131) and runs
the graph in f32 (nanorevcnn.py:24-37): a spike sample, an open-pore stretch or a tiny MAD has a well-defined
answer.  The f32 and bf16x3 modes keep f32 buffers; the default f16x2 mode represents the signal branch as
scaled f16 pairs (|S| < 1023) and must therefore DETECT what it cannot represent and hand back the f32
kernels' result instead (include/nanorev.h nrv_saturated; nrv_cnn_r.h "Range guard").  Checked here:
every mode against the fp64 oracle at 10x / 100x / 1000x the fixture amplitude, spikes inside otherwise clean
reads (only the affected pipeline stage is re-run, and it is bit-identical to the f32 mode), int16 extremes with
a tiny MAD through the raw-read entry point, NaN / Inf samples (propagated, not masked), and the
device-pointer protocol.
"""
import numpy as np
import pytest

from nanoreviser_amd import hoststage as hs
from parity_policy import check_vs_fp64, f32_floor

pytestmark = pytest.mark.gpu

MODES = ["f16x2", "bf16x3", "f32"]


def _fixture_windows(reads, key="ch10_read5252", lo=1000, n=320, T=11):
    _, _, rt = reads(key)
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, T)
    return np.ascontiguousarray(sw[lo:lo + n]), np.ascontiguousarray(fw[lo:lo + n])


def _conv1_sample_bound(m):
    """The engine's static bound (nrv_api.hip upload_model, cnn_r_kernel's ep[24]): below it no conv1 + BatchNorm
    output can leave the f16 range of the f16x2 signal branch (|c1| x 2^6 <= 65504)."""
    w, b, g, be, mu, var = [np.asarray(x, np.float64) for x in m.tensors[:6]]
    inv = g / np.sqrt(var + 1e-3)
    sh = be - mu * inv
    return float((((1000.0 - np.abs(sh)) / np.abs(inv) - np.abs(b)) / np.abs(w[:, 0, :]).sum(0)).min())


@pytest.mark.parametrize("sp", ["ecoli", "human"])
@pytest.mark.parametrize("spike", [150.0, 300.0, 600.0, 1000.0])
def test_isolated_spikes_in_the_conv1_overflow_band(reads, species_models, sp, spike):
    """ONE isolated sample of 150 ... 1000 normalised units (an ADC-saturated spike: (32767 - median) /
    MAD) overflows conv1 + BatchNorm's f16 pair (gain 5 - 11, kept x 2^6) while S itself stays in range - the guard on
    S alone let such a window through with zeroed conv2 features.  Now a sample beyond the static bound trips the
    guard: the stage is re-run on the f32 kernels (bit-identical to the f32 mode) whenever a spike exceeds the bound,
    and every result meets the parity policy against fp64 either way."""
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models[sp]
    sw, fw = _fixture_windows(reads, n=256)
    sw = sw.copy()
    rng = np.random.default_rng(int(spike))
    hit = rng.choice(256, 24, replace=False)
    for w in hit:
        sw[w, rng.integers(11), rng.integers(50)] = np.float32(spike) * (1 if rng.integers(2) else -1)
    bound = min(_conv1_sample_bound(m1), _conv1_sample_bound(m2))
    rv = Reviser(m1, m2, precision="f16x2")
    got = rv.predict_pair(sw, fw)
    _, reruns = rv.saturated()
    rv.set_precision("f32")
    ref32 = rv.predict_pair(sw, fw)
    rv.close()
    print(f"SPIKE {sp} {spike:g}: bound {bound:.1f} reruns {reruns}")
    if spike > 1.001 * bound:
        assert reruns == 1
        for g, r in zip(got, ref32):
            assert np.array_equal(g, r)
    elif spike < 0.999 * bound:
        assert reruns == 0
    q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float64)
    nf1, nf2 = f32_floor(m1, m2, sw, fw, q1, q2)
    check_vs_fp64(got[0], got[2], q1, nf1, f"spike {spike:g} m1", max_ill=0.05)
    check_vs_fp64(got[1], got[3], q2, nf2, f"spike {spike:g} m2", max_ill=0.05)


@pytest.mark.parametrize("amp", [20.0, 40.0, 60.0])
def test_whole_windows_in_the_conv1_overflow_band(reads, species_models, amp):
    """Whole windows at 20 ... 60 x the fixture amplitude (|x| up to 170 ... 500): between the 10x case (nothing leaves
    the range) and the 100x case (S itself overflows).  Policy against fp64; re-run => the f32 mode's bits."""
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models["human"]
    sw, fw = _fixture_windows(reads, n=256)
    sig = (sw * np.float32(amp)).astype(np.float32)
    rv = Reviser(m1, m2, precision="f16x2")
    got = rv.predict_pair(sig, fw)
    _, reruns = rv.saturated()
    rv.set_precision("f32")
    ref32 = rv.predict_pair(sig, fw)
    rv.close()
    bound = min(_conv1_sample_bound(m1), _conv1_sample_bound(m2))
    if float(np.abs(sig).max()) > bound:
        assert reruns == 1                       # (below the bound S itself may still leave the range: |S| <= 99 at 1x)
    if reruns:
        for g, r in zip(got, ref32):
            assert np.array_equal(g, r)
    q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sig, fw, np.float64)
    nf1, nf2 = f32_floor(m1, m2, sig, fw, q1, q2)
    # (amplified windows through the HUMAN weights: 5 - 8 % of them are ill-conditioned for f32 arithmetic itself - the
    # two CPU f32 restatements leave half the bar there -, so the share allowed is 10 %; the engine's own error on
    # every window, well- or ill-conditioned, is held to the same policy as everywhere else)
    # ... and the bar grows with the amplitude, 1e-5 per unit of amplification (2e-4 at 20x, the amplified-input bar of
    # test_amplified_signal_vs_fp64_oracle; 6e-4 at 60x): what is returned here IS the f32 mode's result (asserted above),
    # and plain f32 matrix arithmetic on inputs of |x| <= 500 sits 1.4e-4 from fp64 on one of these windows at 40x (r04g)
    # and 3.9e-4 at 60x (r04y) - the guard's job is to hand over that result, not to beat f32.
    bar = 1e-5 * amp
    check_vs_fp64(got[0], got[2], q1, nf1, f"x{amp:g} m1", max_ill=0.10, bar=bar)
    check_vs_fp64(got[1], got[3], q2, nf2, f"x{amp:g} m2", max_ill=0.10, bar=bar)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("amp", [10.0, 100.0, 1000.0])
def test_amplified_signal_vs_fp64_oracle(reads, species_models, mode, amp):
    """Whole windows at amp x the fixture amplitude (|x| up to 8.4e3 normalised units): every mode within the
    parity policy of the fp64 oracle; the f16x2 mode must have noticed (amp >= 100) and re-run the stage."""
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models["ecoli"]
    sw, fw = _fixture_windows(reads)
    sig = (sw * np.float32(amp)).astype(np.float32)
    rv = Reviser(m1, m2, precision=mode)
    p1, p2, a1, a2 = rv.predict_pair(sig, fw)
    _, reruns = rv.saturated()
    rv.close()
    q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sig, fw, np.float64)
    nf1, nf2 = f32_floor(m1, m2, sig, fw, q1, q2)
    # At 1000x (samples of +-8e3, conv features of ~1e5) f32 arithmetic itself is the limit: the 400-term dense
    # layer cancels to ~0.1 absolute, the restatements leave BAR/2 on ~1.7 % of these windows, and whether a
    # window looks well-conditioned to the two CPU restatements is partly luck - the f32 MODE, plain f32 matrix
    # instructions, sits 1.11e-4 from fp64 on such a window (bf16x3 1.37e-4; r03a).  So the bar there is 2e-4,
    # for every mode alike; 10x and 100x hold the normal 1e-4.  Calls must still be the arbiter's.
    bar = 2e-4 if amp >= 1000 else 1e-4
    r1 = check_vs_fp64(p1, a1, q1, nf1, f"{mode} x{amp:g} m1", max_ill=0.04, bar=bar)
    r2 = check_vs_fp64(p2, a2, q2, nf2, f"{mode} x{amp:g} m2", max_ill=0.04, bar=bar)
    print(f"RANGE {mode} x{amp:g}: reruns {reruns} m1 {r1} m2 {r2}")
    if mode == "f16x2" and amp >= 100:
        assert reruns == 1                       # |S| > 1023: not representable, the stage ran on the f32 kernels
    if mode != "f16x2":
        assert reruns == 0


def test_fixture_reads_never_trip_the_guard(reads, species_models):
    """Counter is 0 on every fixture read, both species, window and read mode (the guard costs nothing there)."""
    from nanoreviser_amd.engine import Reviser
    for sp in ("ecoli", "human"):
        rv = Reviser(*species_models[sp], precision="f16x2")
        for key in reads.keys:
            _, _, rt = reads(key)
            rv.predict_read(rt.sig_ev, rt.feat_ev)
        sw, fw = _fixture_windows(reads, n=5000)
        rv.predict_pair(sw, fw)
        assert rv.saturated() == (0, 0), sp
        rv.close()


def test_spike_stage_is_rerun_and_bit_identical_to_f32_mode(reads, species_models):
    """Three pipeline stages of 4096 windows; a few spike samples (x 2000) in the middle one.  The clean stages
    keep the f16x2 bits, the spiked stage carries exactly the f32 mode's bits, and all of it meets the policy
    against fp64."""
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models["human"]
    _, _, rt = reads("ch13_read2251")
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
    sw, fw = np.ascontiguousarray(sw[:3 * 4096]).copy(), np.ascontiguousarray(fw[:3 * 4096])
    clean = sw.copy()
    rng = np.random.default_rng(7)
    hit = 4096 + rng.choice(4096, 40, replace=False)
    for w in hit:
        sw[w, rng.integers(11), rng.integers(50)] *= np.float32(2000.0)
    rv = Reviser(m1, m2, precision="f16x2")
    base = rv.predict_pair(clean, fw)
    assert rv.saturated() == (0, 0)
    got = rv.predict_pair(sw, fw)
    assert rv.saturated() == (0, 1)                            # exactly the middle stage
    rv.set_precision("f32")
    ref32 = rv.predict_pair(sw, fw)
    rv.close()
    for g, b, r in zip(got, base, ref32):
        assert np.array_equal(g[:4096], b[:4096]) and np.array_equal(g[8192:], b[8192:])
        assert np.array_equal(g[4096:8192], r[4096:8192])
    # the spiked windows against the fp64 oracle (plus neighbours)
    idx = np.unique(np.concatenate([hit, hit + 1, np.arange(4096, 4096 + 200)]))
    s, f = np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx])
    q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, s, f, np.float64)
    nf1, nf2 = f32_floor(m1, m2, s, f, q1, q2)
    check_vs_fp64(got[0][idx], got[2][idx], q1, nf1, "spikes m1", max_ill=0.05)
    check_vs_fp64(got[1][idx], got[3][idx], q2, nf2, "spikes m2", max_ill=0.05)


@pytest.mark.parametrize("mode", MODES)
def test_raw_reads_int16_extremes_and_tiny_mad(reads, species_models, mode):
    """nrv_predict_reads_raw with samples at the int16 limits and a scale (MAD) of 1: normalised samples of
    +-3e4.  Same windows through the host-cut path of the oracle (segmentation is bit-exact, tested elsewhere)."""
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models["ecoli"]
    _, rd, rt = reads("ch141_read5436")
    rr = hs.read_tensors_raw(rd)
    N = 400
    starts = rr.starts[:N].copy()
    raw = rr.raw[: int(starts[-1]) + 60].copy()
    rng = np.random.default_rng(11)
    pos = rng.choice(len(raw), 30, replace=False)
    raw[pos] = rng.choice(np.array([-32768, 32767], np.int16), 30)
    feat = rr.feat_ev[:N]
    shift, scale = float(np.median(raw)), 1.0                  # a tiny MAD: every sample is tens of units out
    rv = Reviser(m1, m2, precision=mode)
    p1, p2, a1, a2 = rv.predict_reads_raw([raw], [starts], [feat], [shift], [scale])
    sig_ev = rv.segment_reads([raw], [starts], [shift], [scale])
    _, reruns = rv.saturated()
    rv.close()
    assert np.abs(sig_ev).max() > 3e4
    assert reruns == (1 if mode == "f16x2" else 0)
    sw, fw = hs.sliding_windows(sig_ev, feat, 11)
    sw, fw = np.ascontiguousarray(sw), np.ascontiguousarray(fw)
    q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float64)
    nf1, nf2 = f32_floor(m1, m2, sw, fw, q1, q2)
    check_vs_fp64(p1, a1, q1, nf1, f"{mode} raw extremes m1", max_ill=0.05)
    check_vs_fp64(p2, a2, q2, nf2, f"{mode} raw extremes m2", max_ill=0.05)


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_nan_and_inf_samples_propagate_like_the_oracle(reads, species_models, bad):
    """A NaN / Inf sample (or event feature) is not masked into a finite value: exactly the windows the fp64
    oracle poisons come out NaN with call 0 (NumPy's argmax of NaNs) in every mode, the others are untouched,
    and the f16x2 result is the f32 kernels' bit for bit (the guard re-ran the stage)."""
    import warnings
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models["ecoli"]
    sw, fw = _fixture_windows(reads, n=256)
    sw, fw = sw.copy(), fw.copy()
    clean = sw.copy()
    sw[17, 3, 20] = bad
    sw[200, 10, 49] = bad
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        q1, q2, b1, b2 = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float64)
    poisoned = ~np.isfinite(q1).all(-1)
    assert set(np.nonzero(poisoned)[0]) == {17, 200} and (~np.isfinite(q2).all(-1) == poisoned).all()
    outs = {}
    for mode in MODES:
        rv = Reviser(m1, m2, precision=mode)
        base = rv.predict_pair(clean, fw)
        outs[mode] = rv.predict_pair(sw, fw)
        assert rv.saturated()[1] == (1 if mode == "f16x2" else 0)
        p1, p2, a1, a2 = outs[mode]
        assert (np.isnan(p1).all(-1) == poisoned).all() and (np.isnan(p2).all(-1) == poisoned).all(), mode
        assert (a1[poisoned] == b1[poisoned]).all() and (a2[poisoned] == b2[poisoned]).all()
        if mode != "f16x2":                                    # (there the whole stage moved to the f32 kernels)
            for x, y in zip(outs[mode], base):
                assert np.array_equal(x[~poisoned], y[~poisoned])
        # a NaN event feature takes the other route (lstm1, no guard involved) and must poison its window too
        f2 = fw.copy()
        f2[5, 2, 1] = np.nan
        r1, r2, ra1, _ = rv.predict_pair(clean, f2)
        assert np.isnan(r1[5]).all() and np.isnan(r2[5]).all() and ra1[5] == 0
        assert np.isfinite(r1[np.arange(256) != 5]).all()
        rv.close()
    for x, y in zip(outs["f16x2"], outs["f32"]):
        assert np.array_equal(x, y, equal_nan=True)            # the re-run IS the f32 kernels


def test_device_pointer_protocol(reads, species_models):
    """Device-pointer calls are asynchronous: the guard is a pending count read by nrv_saturated, and
    Reviser.predict_device_checked repeats the call in f32.  Clean input: pending stays 0."""
    import torch
    from nanoreviser_amd.engine import Reviser
    m1, m2 = species_models["ecoli"]
    sw, fw = _fixture_windows(reads, n=600)
    spiked = sw.copy()
    spiked[300, 5, 25] = 1.0e5
    rv = Reviser(m1, m2, precision="f16x2", batch=256)
    f32 = Reviser(m1, m2, precision="f32", batch=256)

    def run(s, checked):
        d_s, d_f = torch.from_numpy(s).cuda(), torch.from_numpy(fw).cuda()
        n = len(s)
        o = (torch.empty(n, 6, device="cuda"), torch.empty(n, 5, device="cuda"),
             torch.empty(n, dtype=torch.int8, device="cuda"), torch.empty(n, dtype=torch.int8, device="cuda"))
        torch.cuda.synchronize()
        ptr = (d_s.data_ptr(), d_f.data_ptr(), n) + tuple(x.data_ptr() for x in o)
        if checked is None:
            f32.predict_device(*ptr)
            f32.sync()
            return o, None
        if checked:
            return o, rv.predict_device_checked(*ptr)
        rv.predict_device(*ptr)
        return o, rv.saturated()[0]

    _, pend = run(sw, False)
    assert pend == 0
    _, pend = run(spiked, False)
    assert pend > 0
    assert rv.saturated()[0] == 0                              # reading clears it
    o, redone = run(spiked, True)
    assert redone is True and rv.precision == "f16x2"
    ref, _ = run(spiked, None)
    for x, y in zip(o, ref):
        assert torch.equal(x, y)
    o, redone = run(sw, True)
    assert redone is False
    rv.close(); f32.close()
