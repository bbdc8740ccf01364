"""Parity tests proper: the HIP path, called through the C-ABI (ctypes -> libnanorev_hip.so),
against the oracle and the committed goldens.  MI355X only (-m gpu).

Bars (BASELINE.json north_star): per-base argmax identical; softmax probabilities within 1e-4 (fp32).
How they are asserted, per window, against the fp64 arbiter AND both f32 restatements of the oracle:

  * BAR = 1e-4 for both species.  A window is WELL-CONDITIONED when the oracle's own two f32
    implementations (NumPy-f32 and the C port) both stay within BAR/2 of fp64 on it; on those the
    engine must be within BAR of fp64, no exceptions.
  * On the remaining, ILL-CONDITIONED windows fp32 arithmetic itself cannot hold the bar (the human
    model2 has windows where both f32 restatements sit 2.1e-4 from fp64: SURVEY.md App. B,
    tests/test_oracle.py); there the engine must be no worse than 3x the f32 restatements' own
    deviation, and such windows must be rare (<= 1 %).  Their count and the maxima are printed and
    recorded in DESIGN.md 5.
  * argmax must equal the fp64 arbiter's.  The only tolerated difference is a NEAR-TIE: the arbiter's
    top-2 margin on that window is below twice the measured f32 deviation of that window (floor
    1e-6), i.e. fp32 arithmetic does not determine the call.  Near-ties are counted and the count is
    asserted (0 on every real read; the synthetic T=13 set holds one window with margin 1.5e-6).
"""
import numpy as np
import pytest

from nanoreviser_amd import hoststage as hs

pytestmark = pytest.mark.gpu

from parity_policy import BAR, check_vs_fp64, f32_floor

MODES = ["f16x2", "bf16x3", "f32"]


@pytest.fixture(scope="module", params=MODES, autouse=True)
def precision(request):
    """Every test of this module runs once per matrix-arithmetic mode (include/nanorev.h
    nrv_set_precision); NRV_PRECISION is what a new handle starts in.  Same bars for both."""
    import os
    old = os.environ.get("NRV_PRECISION")
    os.environ["NRV_PRECISION"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("NRV_PRECISION", None)
    else:
        os.environ["NRV_PRECISION"] = old


@pytest.fixture(scope="module")
def engines(species_models, precision):
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from nanoreviser_amd.engine import Reviser
    revs = {sp: Reviser(*species_models[sp]) for sp in species_models}
    for rv in revs.values():
        assert rv.backend == "hip" and rv.precision == precision
    yield revs
    for rv in revs.values():
        rv.close()


def _windows(reads, key, T=11):
    _, _, rt = reads(key)
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, T)
    return rt, sw, fw


@pytest.mark.parametrize("sp", ["ecoli", "human"])
def test_fixture_reads_vs_committed_fp64_goldens(engines, reads, model_goldens, species_models, sp, precision):
    rv = engines[sp]
    m1, m2 = species_models[sp]
    tot = {"max_err": 0.0, "max_err_well": 0.0, "ill": 0, "near_ties": 0, "n": 0, "max_f32_floor": 0.0}
    for key in reads.keys:
        _, sw, fw = _windows(reads, key)
        idx = model_goldens[f"{key}/idx"]
        sig, rd = np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx])
        g1, g2 = model_goldens[f"{key}/{sp}/p1"], model_goldens[f"{key}/{sp}/p2"]
        p1, p2, a1, a2 = rv.predict_pair(sig, rd)
        nf1, nf2 = f32_floor(m1, m2, sig, rd, g1, g2)
        for r in (check_vs_fp64(p1, a1, g1, nf1, f"{sp} {key} m1"), check_vs_fp64(p2, a2, g2, nf2, f"{sp} {key} m2")):
            for k in tot:
                tot[k] = max(tot[k], r[k]) if k.startswith("max") else tot[k] + r[k]
    assert tot["near_ties"] == 0                           # real reads: every call is the arbiter's
    print(f"PARITY {precision} {sp}: {tot}")


@pytest.mark.parametrize("sp", ["ecoli", "human"])
def test_synthetic_goldens_T11_and_T13(species_models, model_goldens, sp, precision):
    from nanoreviser_amd.engine import Reviser
    m1, m2 = species_models[sp]
    ties = 0
    for T in (11, 13):
        a, b = m1.with_window(T), m2.with_window(T)
        rv = Reviser(a, b)
        sig, rd = model_goldens[f"synth{T}/signal"], model_goldens[f"synth{T}/read"]
        g1, g2 = model_goldens[f"synth{T}/{sp}/p1"], model_goldens[f"synth{T}/{sp}/p2"]
        p1, p2, a1, a2 = rv.predict_pair(sig, rd)
        nf1, nf2 = f32_floor(a, b, sig, rd, g1, g2, T)
        ties += check_vs_fp64(p1, a1, g1, nf1, f"synth{T} {sp} m1")["near_ties"]
        ties += check_vs_fp64(p2, a2, g2, nf2, f"synth{T} {sp} m2")["near_ties"]
        rv.close()
    assert ties <= 1                                       # synth13 holds one window with an fp64 margin of 1.5e-6
    print(f"PARITY {precision} {sp} synthetic: near-ties {ties}")


@pytest.mark.parametrize("sp", ["ecoli", "human"])
def test_whole_reads_vs_both_f32_restatements(engines, reads, species_models, sp, precision):
    """Every window of two whole fixture reads (15.4 k windows) per species: the engine against the
    reference-shaped f32 arithmetic (C port; NumPy-f32 on a slice).  Two correct f32 evaluations of
    the same graph differ by at most the sum of their rounding noises: <= BAR on every window where
    the graph is well-conditioned, identical calls outside fp32 near-ties."""
    from oracle import c_oracle as CO
    from oracle import nrv_oracle as O
    rv = engines[sp]
    m1, m2 = species_models[sp]
    worst_c, worst_np, n_over, n_mis, n_all = 0.0, 0.0, 0, 0, 0
    for key in ("ch117_read6465", "ch13_read2251"):
        _, sw, fw = _windows(reads, key)
        sw, fw = np.ascontiguousarray(sw), np.ascontiguousarray(fw)
        p1, p2, a1, a2 = rv.predict_pair(sw, fw)
        c1, ca1 = CO.predict(m1.flat(), 11, 6, sw, fw, threads=8)
        c2, ca2 = CO.predict(m2.flat(), 11, 5, sw, fw, threads=8)
        d = np.maximum(np.abs(p1 - c1).max(-1), np.abs(p2 - c2).max(-1))
        worst_c = max(worst_c, float(d.max()))
        n_over += int((d > BAR).sum())
        n_all += len(d)
        for a, ca, c in ((a1, ca1, c1), (a2, ca2, c2)):
            for i in np.nonzero(a != ca)[0]:
                srt = np.sort(c[i])
                assert srt[-1] - srt[-2] <= 2 * BAR, (key, int(i))    # only where f32 itself cannot decide
                n_mis += 1
        sl = slice(1000, 1400)
        q1, q2, _, _ = O.predict_pair(m1.tensors, m2.tensors, sw[sl], fw[sl], np.float32)
        worst_np = max(worst_np, float(np.abs(p1[sl] - q1).max()), float(np.abs(p2[sl] - q2).max()))
    print(f"PARITY {precision} {sp} whole reads: max|dp| vs C-f32 {worst_c:.2e} ({n_over} of {n_all} windows > 1e-4), "
          f"vs NumPy-f32 {worst_np:.2e}, argmax differences vs C-f32 {n_mis}")
    # Two f32 evaluations of one graph differ by at most the sum of their own deviations from fp64, so a
    # handful of windows in 20 000 sit a little over 1e-4 against ANOTHER f32 evaluation (README says so next
    # to "within 1e-4").  The bars are the maxima measured over the three modes on MI355X (round 3, r03c:
    # E. coli 1.07e-4, human 1.20e-4 against the C port; 4.2e-5 against NumPy-f32 on the slice) + 10 %; the
    # kernels are deterministic, so a change here means the arithmetic changed and the numbers are re-measured.
    assert n_over <= 1 and n_mis == 0
    assert worst_c <= (1.18e-4 if sp == "ecoli" else 1.32e-4)
    assert worst_np <= 4.7e-5


@pytest.mark.parametrize("sp,batch", [("ecoli", 512), ("human", 4096)])
def test_config_batch_sizes_C2_C3(reads, species_models, sp, batch):
    """BASELINE configs[1] (ecoli, batch=512 windows) and configs[2] (human, batch=4096) on whole
    fixture reads through the host entry points: same bits as any other grouping, checked against the
    C-f32 oracle."""
    from nanoreviser_amd.engine import Reviser
    from oracle import c_oracle as CO
    m1, m2 = species_models[sp]
    rt, sw, fw = _windows(reads, "ch141_read5436")
    sw, fw = np.ascontiguousarray(sw), np.ascontiguousarray(fw)
    rv = Reviser(m1, m2, batch=batch)
    assert rv.batch == batch
    p1, p2, a1, a2 = rv.predict_pair(sw, fw)
    r = rv.predict_read(rt.sig_ev, rt.feat_ev)
    rv.close()
    rv = Reviser(m1, m2, batch=1000)
    o = rv.predict_pair(sw, fw)
    rv.close()
    for x, y, z in zip((p1, p2, a1, a2), r, o):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    c1, ca1 = CO.predict(m1.flat(), 11, 6, sw, fw, threads=8)
    c2, ca2 = CO.predict(m2.flat(), 11, 5, sw, fw, threads=8)
    e1, e2, flips = float(np.abs(p1 - c1).max()), float(np.abs(p2 - c2).max()), int((a1 != ca1).sum() + (a2 != ca2).sum())
    print(f"MEASURED C2C3 {sp} batch {batch}: max|dp| vs C-f32 m1 {e1:.3e} m2 {e2:.3e}, argmax differences {flips}")
    # measured maxima over the three modes (r03c): E. coli 2.4e-5, human 9.1e-5; + 10 %
    assert max(e1, e2) <= (2.7e-5 if sp == "ecoli" else 1.0e-4)
    assert flips == 0


def test_whole_read_live_oracle_and_read_mode(engines, reads, species_models):
    """A full fixture read (6.6 k windows): window mode vs the live C oracle (f32) and NumPy fp64
    on a slice; read mode (device-formed windows, per-event CNN) bit-identical to window mode."""
    from oracle import c_oracle as CO
    from oracle import nrv_oracle as O
    rv = engines["ecoli"]
    m1, m2 = species_models["ecoli"]
    rt, sw, fw = _windows(reads, "ch10_read5252")
    sw, fw = np.ascontiguousarray(sw), np.ascontiguousarray(fw)
    p1, p2, a1, a2 = rv.predict_pair(sw, fw)
    r1, r2, ra1, ra2 = rv.predict_read(rt.sig_ev, rt.feat_ev)
    assert r1.shape == p1.shape == (len(rt.feat_ev) - 11, 6)
    assert np.array_equal(p1, r1) and np.array_equal(p2, r2)
    assert np.array_equal(a1, ra1) and np.array_equal(a2, ra2)
    c1, ca1 = CO.predict(m1.flat(), 11, 6, sw, fw, threads=8)
    c2, ca2 = CO.predict(m2.flat(), 11, 5, sw, fw, threads=8)
    assert np.array_equal(a1, ca1) and np.array_equal(a2, ca2)
    assert np.abs(p1 - c1).max() <= 1e-4 and np.abs(p2 - c2).max() <= 1e-4
    sl = slice(3000, 3300)
    q1, q2, b1, b2 = O.predict_pair(m1.tensors, m2.tensors, sw[sl], fw[sl], np.float64)
    assert np.abs(p1[sl] - q1).max() <= 1e-4 and np.abs(p2[sl] - q2).max() <= 1e-4
    assert np.array_equal(a1[sl], b1) and np.array_equal(a2[sl], b2)
    # domain property: calls agree with the basecalls at the window centre
    _, rd, _ = reads("ch10_read5252")
    lab = np.array([hs.BASE_LABEL[b.decode()] for b in rd.bases])[5:5 + len(a1)]
    assert (a1 == lab).mean() > 0.95 and (a2 + 1 == lab).mean() > 0.97


def test_ragged_and_empty_inputs(engines, reads, species_models):
    from oracle import c_oracle as CO
    rv = engines["ecoli"]
    m1, m2 = species_models["ecoli"]
    rt, sw, fw = _windows(reads, "ch141_read5436")
    ref1, refa1 = CO.predict(m1.flat(), 11, 6, np.ascontiguousarray(sw[:130]), np.ascontiguousarray(fw[:130]), threads=8)
    base = rv.predict_pair(np.ascontiguousarray(sw[:130]), np.ascontiguousarray(fw[:130]))
    assert np.abs(base[0] - ref1).max() <= 1e-4 and np.array_equal(base[2], refa1)
    for n in (0, 1, 2, 31, 32, 33, 63, 64, 65, 127, 129):
        p1, p2, a1, a2 = rv.predict_pair(np.ascontiguousarray(sw[:n]), np.ascontiguousarray(fw[:n]))
        assert p1.shape == (n, 6) and p2.shape == (n, 5) and a1.shape == (n,) and a2.shape == (n,)
        # rows are independent: a prefix gives the same bits as the larger call
        assert np.array_equal(p1, base[0][:n]) and np.array_equal(p2, base[1][:n])
        assert np.array_equal(a1, base[2][:n]) and np.array_equal(a2, base[3][:n])
    # read mode: N <= T yields no windows; N = T+1 yields one
    for N in (0, 5, 11):
        p1, p2, a1, a2 = rv.predict_read(rt.sig_ev[:N], rt.feat_ev[:N])
        assert p1.shape == (0, 6) and a2.shape == (0,)
    p1, _, a1, _ = rv.predict_read(rt.sig_ev[:12], rt.feat_ev[:12])
    assert p1.shape == (1, 6) and np.array_equal(p1[0], base[0][0]) and a1[0] == base[2][0]


def test_batch_grouping_is_invisible(reads, species_models):
    """Keras predict(batch_size=...) must not change results: groups of 64/96/4096 windows."""
    from nanoreviser_amd.engine import Reviser
    m1, m2 = species_models["human"]
    rt, sw, fw = _windows(reads, "ch13_read2251")
    sw, fw = np.ascontiguousarray(sw[:1000]), np.ascontiguousarray(fw[:1000])
    outs = []
    for b in (4096, 96, 64):
        rv = Reviser(m1, m2, batch=b)
        assert rv.batch == b
        outs.append(rv.predict_pair(sw, fw) + rv.predict_read(rt.sig_ev[:1011], rt.feat_ev[:1011]))
        rv.close()
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert np.array_equal(x, y)
    # Keras-shaped facade: model1.predict / model2.predict
    rv = Reviser(m1, m2)
    assert np.array_equal(rv.model1.predict([sw[..., None], fw]), outs[0][0])
    assert np.array_equal(rv.model2.predict([sw[..., None], fw], batch_size=512), outs[0][1])
    # the facade's pair cache must not serve stale results after an in-place refill of the same arrays
    s4, f4 = sw[..., None].copy(), fw.copy()
    first = rv.model1.predict([s4, f4])
    s4[:] = s4[::-1].copy(); f4[:] = f4[::-1].copy()
    assert np.array_equal(rv.model2.predict([s4, f4]), outs[0][1][::-1])
    assert np.array_equal(rv.model1.predict([s4, f4]), first[::-1])
    rv.close()


def test_sigmoid_variant_matches_oracle(reads, species_models):
    """recurrent_act=1 (Keras >= 2.3 default, enviroment/NanoReviser_macOS.yaml) is a different
    function; the engine implements it too and it must match the oracle's sigmoid variant."""
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models["ecoli"]
    _, sw, fw = _windows(reads, "ch117_read6465")
    sw, fw = np.ascontiguousarray(sw[:200]), np.ascontiguousarray(fw[:200])
    rv = Reviser(m1, m2, recurrent_activation="sigmoid")
    p1, p2, a1, a2 = rv.predict_pair(sw, fw)
    q1, q2, b1, b2 = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float64, recurrent_act="sigmoid")
    assert np.abs(p1 - q1).max() <= 1e-4 and np.abs(p2 - q2).max() <= 1e-4
    assert np.array_equal(a1, b1) and np.array_equal(a2, b2)
    h1 = O.forward(m1.tensors, sw, fw, np.float64)
    assert np.abs(p1 - h1).max() > 1e-3          # and it is NOT the hard_sigmoid function
    rv.close()


def test_extreme_inputs_stay_finite(engines):
    """Saturating gates, zero signal, huge features: no NaN/Inf, valid distributions."""
    from oracle import nrv_oracle as O
    rv = engines["ecoli"]
    sig, rd = O.synth_windows(96, 11, seed=3)
    sig[:32] = 0.0
    rd[32:64] *= 50.0
    sig[64:] = np.where(np.arange(50) % 2 == 0, 4.8, -8.4)
    p1, p2, a1, a2 = rv.predict_pair(sig, rd)
    for p in (p1, p2):
        assert np.isfinite(p).all() and (p >= 0).all()
        assert np.abs(p.sum(-1) - 1).max() < 1e-5
    assert np.array_equal(a1, p1.argmax(-1)) and np.array_equal(a2, p2.argmax(-1))


@pytest.mark.parametrize("sp", ["ecoli", "human"])
def test_full_size_properties_device_api(species_models, sp):
    """BASELINE sizes through the device-pointer entry points: 1 M synthetic 13-event windows in
    groups of 4096 (configs[3]) and one 200 k-event read (configs[4]).  Size-independent properties:
    determinism, permutation equivariance (rows independent), batch-grouping invariance,
    probabilities sum to 1, argmax == argmax(prob), read mode == window mode."""
    import torch
    from nanoreviser_amd.engine import Reviser
    from oracle import c_oracle as CO
    m1, m2 = species_models[sp]
    T, n = 13, (1 << 20) if sp == "ecoli" else (1 << 17)     # configs[3] is the E. coli one
    rv = Reviser(m1.with_window(T), m2.with_window(T), batch=4096)
    # device-pointer calls are asynchronous on the handle's stream, which is ordered after the
    # default stream torch produces these inputs on
    g = torch.Generator(device="cuda").manual_seed(1234)
    sig = (torch.randn(n, T, 50, device="cuda", generator=g) * 1.36 - 0.10).clamp_(-8.4, 4.8)
    feat = torch.rand(n, T, 6, device="cuda", generator=g)
    feat[..., 4] = feat[..., 4] * 80 + 70
    feat[..., 5] = feat[..., 5] * 10 + 1

    def run(s, f, rvx):
        k = s.shape[0]
        p1 = torch.empty(k, 6, device="cuda"); p2 = torch.empty(k, 5, device="cuda")
        a1 = torch.empty(k, dtype=torch.int8, device="cuda"); a2 = torch.empty(k, dtype=torch.int8, device="cuda")
        rvx.predict_device(s.data_ptr(), f.data_ptr(), k, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
        rvx.sync()
        torch.cuda.synchronize()
        return p1, p2, a1, a2

    o = run(sig, feat, rv)
    o2 = run(sig, feat, rv)
    for x, y in zip(o, o2):
        assert torch.equal(x, y)                                   # deterministic
    for p in o[:2]:
        assert torch.isfinite(p).all() and (p.sum(-1) - 1).abs().max() < 1e-5
    assert torch.equal(o[2].long(), o[0].argmax(-1)) and torch.equal(o[3].long(), o[1].argmax(-1))
    perm = torch.randperm(n, device="cuda", generator=g)
    op = run(sig[perm].contiguous(), feat[perm].contiguous(), rv)
    for x, y in zip(o, op):
        assert torch.equal(x[perm], y)                             # rows are independent
    rv.set_batch(1000)                                             # ragged groups
    sub = slice(0, 50_000)
    ob = run(sig[sub].contiguous(), feat[sub].contiguous(), rv)
    for x, y in zip(o, ob):
        assert torch.equal(x[sub], y)
    rv.close()

    # one long read, streamed in device-formed window groups
    N = 200_000
    rv = Reviser(m1.with_window(T), m2.with_window(T), batch=4096)
    sig_ev = (torch.randn(N, 50, device="cuda", generator=g) * 1.36 - 0.10).clamp_(-8.4, 4.8)
    feat_ev = torch.rand(N, 6, device="cuda", generator=g)
    k = N - T
    p1 = torch.empty(k, 6, device="cuda"); p2 = torch.empty(k, 5, device="cuda")
    a1 = torch.empty(k, dtype=torch.int8, device="cuda"); a2 = torch.empty(k, dtype=torch.int8, device="cuda")
    rv.predict_read_device(sig_ev.data_ptr(), feat_ev.data_ptr(), N, p1.data_ptr(), p2.data_ptr(),
                           a1.data_ptr(), a2.data_ptr())
    rv.sync()
    idx = torch.arange(0, 20_000, device="cuda")[:, None] + torch.arange(T, device="cuda")[None, :]
    for base in (0, 100_000, k - 20_000):
        w = run(sig_ev[idx + base].contiguous(), feat_ev[idx + base].contiguous(), rv)
        assert torch.equal(w[0], p1[base:base + 20_000]) and torch.equal(w[1], p2[base:base + 20_000])
        assert torch.equal(w[2], a1[base:base + 20_000]) and torch.equal(w[3], a2[base:base + 20_000])
    # and the streamed read against the C-f32 oracle's own read mode on a slice from its middle
    lo, cnt = 150_000, 3000
    se, fe = sig_ev[lo:lo + cnt + T].cpu().numpy(), feat_ev[lo:lo + cnt + T].cpu().numpy()
    a, b = m1.with_window(T), m2.with_window(T)
    c1, ca1 = CO.predict_read(a.flat(), T, 6, se, fe, threads=8)
    c2, ca2 = CO.predict_read(b.flat(), T, 5, se, fe, threads=8)
    e1 = float(np.abs(p1[lo:lo + cnt].cpu().numpy() - c1).max())
    e2 = float(np.abs(p2[lo:lo + cnt].cpu().numpy() - c2).max())
    flips = int((a1[lo:lo + cnt].cpu().numpy() != ca1).sum() + (a2[lo:lo + cnt].cpu().numpy() != ca2).sum())
    print(f"MEASURED C5 {sp}: max|dp| vs C-f32 read mode m1 {e1:.3e} m2 {e2:.3e}, argmax differences {flips}")
    # measured maxima over the three modes (r03c): E. coli 1.07e-5, human 4.43e-5; + 10 %
    assert max(e1, e2) <= (1.2e-5 if sp == "ecoli" else 4.9e-5)
    assert flips == 0
    rv.close()


def test_small_groups_coalesced_or_on_lanes_bit_identical(species_models, monkeypatch):
    """Launch groups below 4096 windows: coalesced into 4096-window launches (the default since r04, nrv_api.hip
    group_windows), or - NRV_COALESCE=0 - spread over stream lanes with their own activation buffers (for_groups),
    or - NRV_LANES=0 on top - in order on one stream.  Window and read mode, ragged tail, repeated calls: every
    form bit-identical to groups of 4096 on one stream; nrv_get_batch keeps reporting what the caller set."""
    import torch
    from nanoreviser_amd.engine import Reviser
    m1, m2 = species_models["ecoli"]
    T, n, N = 11, 10_037, 30_011
    g = torch.Generator(device="cuda").manual_seed(77)
    sig = (torch.randn(n, T, 50, device="cuda", generator=g) * 1.36 - 0.10).clamp_(-8.4, 4.8)
    feat = torch.rand(n, T, 6, device="cuda", generator=g)
    sig_ev = (torch.randn(N, 50, device="cuda", generator=g) * 1.36 - 0.10).clamp_(-8.4, 4.8)
    feat_ev = torch.rand(N, 6, device="cuda", generator=g)

    def outs(k):
        # sentinels, not torch.empty: a row the engine does not write must not pass because the allocator handed back the
        # block an earlier, identical run had filled (r06: one unexplained failure of this test in ~12 runs of the suite)
        return (torch.full((k, 6), float("nan"), device="cuda"), torch.full((k, 5), float("nan"), device="cuda"),
                torch.full((k,), -7, dtype=torch.int8, device="cuda"), torch.full((k,), -7, dtype=torch.int8, device="cuda"))

    def run(rv):
        w, r = outs(n), outs(N - T)
        torch.cuda.synchronize()
        rv.predict_device(sig.data_ptr(), feat.data_ptr(), n, *[x.data_ptr() for x in w])
        rv.predict_read_device(sig_ev.data_ptr(), feat_ev.data_ptr(), N, *[x.data_ptr() for x in r])
        rv.sync()
        torch.cuda.synchronize()
        return w + r

    def _where(x, y):
        bad = (x != y) if x.dim() == 1 else (x != y).any(1)
        idx = torch.nonzero(bad).flatten()
        return f"{tuple(x.shape)}: {idx.numel()} rows differ, first {idx[:8].tolist()}, last {idx[-4:].tolist()}; x {x[idx[:2]].tolist()} y {y[idx[:2]].tolist()}"

    ref_rv = Reviser(m1, m2, batch=4096)
    ref = run(ref_rv)
    ref_rv.close()
    assert not any(torch.isnan(x).any() for x in ref[:2] + ref[4:6]) and all((x != -7).all() for x in ref[2:4] + ref[6:8])
    for batch in (512, 1000):                         # coalesced
        rv = Reviser(m1, m2, batch=batch)
        assert rv.batch == batch
        for x, y in zip(ref, run(rv)):
            assert torch.equal(x, y), (batch, _where(x, y))
        rv.close()
    monkeypatch.setenv("NRV_COALESCE", "0")
    for batch in (512, 992, 1000, 2048):              # lanes; 1000: not whole row tiles, stays on one stream
        rv = Reviser(m1, m2, batch=batch)
        for _ in range(2):
            for x, y in zip(ref, run(rv)):
                assert torch.equal(x, y), (batch, _where(x, y))
        rv.close()
    monkeypatch.setenv("NRV_LANES", "0")
    rv = Reviser(m1, m2, batch=512)
    for x, y in zip(ref, run(rv)):
        assert torch.equal(x, y)
    rv.close()


@pytest.mark.parametrize("T", [1, 2, 5, 16, 32])
def test_other_window_lengths_vs_oracle(species_models, T):
    """The engine is parametric in T (1..32); only `feature.kernel` depends on T, so
    every T runs the shipped weights + the seeded synthetic feature kernel, checked against the
    oracle built from the same tensors.  Covers the step-loop edges (T=1: no recurrent product at
    all; T=2: one; T=32: the largest head buffers)."""
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models["ecoli"]
    a, b = m1.with_window(T), m2.with_window(T)
    sig, rd = O.synth_windows(70, T, seed=100 + T)
    rv = Reviser(a, b)
    p1, p2, a1, a2 = rv.predict_pair(sig, rd)
    q1, q2, _, _ = O.predict_pair(a.tensors, b.tensors, sig, rd, np.float64)
    nf1, nf2 = f32_floor(a, b, sig, rd, q1, q2, T)
    check_vs_fp64(p1, a1, q1, nf1, f"T={T} m1")
    check_vs_fp64(p2, a2, q2, nf2, f"T={T} m2")
    # read mode at this T
    N = 70 + T
    rng = np.random.default_rng(T)
    sig_ev = np.clip(rng.normal(-0.1, 1.36, (N, 50)), -8.4, 4.8).astype(np.float32)
    feat_ev = np.abs(rng.normal(0.5, 0.3, (N, 6))).astype(np.float32)
    r = rv.predict_read(sig_ev, feat_ev)
    sw, fw = hs.sliding_windows(sig_ev, feat_ev, T)
    w = rv.predict_pair(np.ascontiguousarray(sw), np.ascontiguousarray(fw))
    assert r[0].shape == (N - T, 6)
    for x, y in zip(r, w):
        assert np.array_equal(x, y)
    rv.close()
    with pytest.raises(Exception):
        Reviser(m1.with_window(33), m2.with_window(33))      # NRV_E_INVALID: T out of range


def test_set_batch_grows_workspace_and_multiple_handles(species_models):
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    m1, m2 = species_models["ecoli"]
    sig, rd = O.synth_windows(9000, 11, seed=5)
    rv_small = Reviser(m1, m2, batch=256)
    rv_big = Reviser(m1, m2)                      # two live handles on one GPU
    a = rv_small.predict_pair(sig, rd)
    rv_small.set_batch(8192)                      # grows the workspace in place
    b = rv_small.predict_pair(sig, rd)
    c = rv_big.predict_pair(sig, rd, batch_size=3000)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    rv_small.close(); rv_big.close()
    rv_small.close()                              # idempotent


def test_device_segmentation_is_bit_exact_and_raw_reads_match(engines, reads):
    """SURVEY 8f-1 on the device: the signal windows cut by segment_kernel from int16 samples +
    event starts are BIT-identical to the host stage (itself pinned to the reference's
    signal_segmentation by sha256, tests/test_hoststage_golden.py), for every fixture read alone and
    for all five in one call; predictions through nrv_predict_reads_raw are identical to
    nrv_predict_read on the host-cut windows, including the windows straddling two reads."""
    rv = engines["ecoli"]
    raws = []
    for key in reads.keys:
        g, rd, rt = reads(key)
        rr = hs.read_tensors_raw(rd)
        assert rr.raw.dtype == np.int16 and rr.starts.dtype == np.int32
        assert np.array_equal(rr.feat_ev, rt.feat_ev) and rr.shift == rt.shift and rr.scale == rt.scale
        sig = rv.segment_reads([rr.raw], [rr.starts], [rr.shift], [rr.scale])
        assert sig.dtype == np.float32 and sig.shape == rt.sig_ev.shape
        assert np.array_equal(sig.view(np.uint32), rt.sig_ev.view(np.uint32)), key
        raws.append((rr, rt))
    # one call, five reads: per-read shift/scale and clipping at each read's own ends
    sig_all = rv.segment_reads([r.raw for r, _ in raws], [r.starts for r, _ in raws],
                               [r.shift for r, _ in raws], [r.scale for r, _ in raws])
    host_all = np.concatenate([t.sig_ev for _, t in raws])
    assert np.array_equal(sig_all.view(np.uint32), host_all.view(np.uint32))
    feat_all = np.concatenate([t.feat_ev for _, t in raws])
    want = rv.predict_read(host_all, feat_all)
    got = rv.predict_reads_raw([r.raw for r, _ in raws], [r.starts for r, _ in raws],
                               [r.feat_ev for r, _ in raws], [r.shift for r, _ in raws],
                               [r.scale for r, _ in raws])
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # edges: starts beyond the samples, a one-sample read, an empty call
    raw = np.arange(-5, 35, dtype=np.int16)
    st = np.array([0, 3, 20, 39, 60, 200], np.int32)
    dev = rv.segment_reads([raw], [st], [1.5], [2.25])
    assert np.array_equal(dev.view(np.uint32), hs.segment_windows_f32(raw, st, 1.5, 2.25).view(np.uint32))
    one = rv.segment_reads([raw[:1]], [st[:2]], [0.0], [1.0])
    assert np.array_equal(one.view(np.uint32), hs.segment_windows_f32(raw[:1], st[:2], 0.0, 1.0).view(np.uint32))
    assert rv.segment_reads([], [], [], []).shape == (0, 50)
    e = rv.predict_reads_raw([raw], [st[:3]], [np.zeros((3, 6), np.float32)], [0.0], [1.0])
    assert e[0].shape == (0, 6) and e[2].shape == (0,)                 # N <= T: no windows


def test_raw_read_descriptors_are_validated(engines):
    from nanoreviser_amd.engine import NrvError, _ReadDesc
    import ctypes as C
    rv = engines["ecoli"]
    raw = np.zeros(100, np.int16)
    st = np.zeros(20, np.int32)
    out = np.empty((20, 50), np.float32)
    for bad in (_ReadDesc(0, 101, 0, 20, 0.0, 1.0),      # samples beyond the array
                _ReadDesc(0, 100, 1, 19, 0.0, 1.0),      # events do not start at 0
                _ReadDesc(0, 100, 0, 19, 0.0, 1.0)):     # events do not cover N
        rc = rv._lib.nrv_segment_reads(rv._h, raw.ctypes.data_as(C.POINTER(C.c_int16)), 100,
                                       st.ctypes.data_as(C.POINTER(C.c_int32)), 20, C.byref(bad), 1,
                                       out.ctypes.data_as(C.POINTER(C.c_float)))
        assert rc == -1
    with pytest.raises(NrvError):
        rv._check(rc)


def test_random_shapes_vs_f32_oracle(species_models, precision):
    """Seeded random window lengths (1..32), batch sizes (1..700), launch-group sizes and both
    recurrent activations against the NumPy f32 oracle (scripts/gpu_fuzz.py runs the open-ended
    version).  Both sides are f32 paths: the bound is the measured maximum of their difference + 10 %."""
    from nanoreviser_amd.engine import Reviser
    from oracle import nrv_oracle as O
    rng = np.random.default_rng(20261)
    m1, m2 = species_models["ecoli"]
    worst = 0.0
    for _ in range(10):
        T = int(rng.integers(1, 33))
        n = int(rng.integers(1, 700))
        batch = int(rng.choice([64, 96, 500, 4096]))
        act = str(rng.choice(["hard_sigmoid", "sigmoid"]))
        a, b = m1.with_window(T), m2.with_window(T)
        sig, rd = O.synth_windows(n, T, seed=int(rng.integers(1 << 30)))
        rv = Reviser(a, b, recurrent_activation=act, batch=batch)
        assert rv.precision == precision
        p1, p2, a1, a2 = rv.predict_pair(sig, rd)
        q1, q2, b1, b2 = O.predict_pair(a.tensors, b.tensors, sig, rd, np.float32, recurrent_act=act)
        worst = max(worst, float(np.abs(p1 - q1).max()), float(np.abs(p2 - q2).max()))
        # measured maximum over the three modes (r03c): 2.95e-5; + 10 %
        assert np.abs(p1 - q1).max() <= 3.3e-5 and np.abs(p2 - q2).max() <= 3.3e-5, (T, n, batch, act)
        for arr, brr, q in ((a1, b1, q1), (a2, b2, q2)):
            for i in np.nonzero(arr != brr)[0]:
                assert q[i, brr[i]] - q[i, arr[i]] <= 6.6e-5, (T, n, batch, act, int(i))
        rv.close()
    print(f"MEASURED random shapes {precision}: max|dp| vs NumPy-f32 {worst:.3e}")
