"""The host-pointer pipeline of nrv_predict over MANY stages (MI355X only, -m gpu): three input staging sets in rotation, two
output sets, uploads one stage ahead, one download per stage (csrc/nrv_api.hip predict_host, r05).  Results do not depend on
the grouping, so a call of many stages (ramped: 1, 1, 2, 4, 4 ... launch groups, r06) must give, window for window, the bits of one-stage calls on the same windows -
with the caller's arrays registered in place AND through the bounce buffers (NRV_HOST_REGISTER=0) - and a stage that trips
the f16x2 range guard late in the call (its inputs must still be in their staging set) must come back with the f32 kernels'
bits while its neighbours keep theirs."""
import os

import numpy as np
import pytest

from nanoreviser_amd import hoststage as hs

pytestmark = pytest.mark.gpu


def _stages(n, group=4096):
    """The window-mode stage schedule of predict_host (r06): 1, 1, 2, then NRV_WINDOW_STAGE_MAX (default 4) launch groups per
    stage -> [(lo, hi)] in windows.  (The variable is read once per process by the library: set it for the whole test run.)"""
    mx = max(1, min(8, int(os.environ.get("NRV_WINDOW_STAGE_MAX", "4"))))
    out, s, k = [], 0, 0
    while s < n:
        g = 1 if (k < 2 or mx == 1) else (2 if k == 2 else mx)
        g = min(g, mx)
        out.append((s, min(s + g * group, n)))
        s, k = out[-1][1], k + 1
    return out


def _seven_stages(reads, T=11):
    _, _, rt = reads("ch13_read2251")
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, T)
    n = 13 * 4096 + 1234                                           # fourteen launch groups, the last one ragged: six stages (1, 1, 2, 4, 4, 2)
    idx = np.arange(n) % (len(fw) - 7)                             # the read's windows, wrapped around
    idx = (idx * 7919) % (len(fw) - 7)                             # ... and shuffled: every stage holds different windows
    return np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx])


@pytest.mark.parametrize("register", ["1", "0"])
def test_seven_stage_call_equals_one_stage_calls(reads, species_models, register, monkeypatch):
    from nanoreviser_amd.engine import Reviser
    monkeypatch.setenv("NRV_HOST_REGISTER", register)              # read at nrv_create
    m1, m2 = species_models["ecoli"]
    sw, fw = _seven_stages(reads)
    rv = Reviser(m1, m2, precision="f16x2")
    whole = rv.predict_pair(sw, fw)
    for s in (0, 3, 5, 9, 13):                                     # launch groups in stages 0, 2, 3 (first reuse of input set 0), 4, the ragged last
        lo, hi = s * 4096, min((s + 1) * 4096, len(fw))
        part = rv.predict_pair(np.ascontiguousarray(sw[lo:hi]), np.ascontiguousarray(fw[lo:hi]))
        for w, p in zip(whole, part):
            assert np.array_equal(w[lo:hi].view(np.uint8), p.view(np.uint8)), (register, s)
    again = rv.predict_pair(sw, fw)                                # and the call is repeatable on one handle
    assert all(np.array_equal(a, b) for a, b in zip(whole, again))
    assert rv.saturated() == (0, 0)
    rv.close()


def test_late_stage_range_rerun_keeps_its_inputs(reads, species_models):
    from nanoreviser_amd.engine import Reviser
    m1, m2 = species_models["human"]
    sw, fw = _seven_stages(reads)
    clean = sw.copy()
    rng = np.random.default_rng(11)
    for stage in (4, 13):                                          # spikes in launch group 4 (stage 3) and in the last one
        lo = stage * 4096
        for w in lo + rng.choice(min(4096, len(fw) - lo), 25, replace=False):
            sw[w, rng.integers(11), rng.integers(50)] *= np.float32(2000.0)
    rv = Reviser(m1, m2, precision="f16x2")
    base = rv.predict_pair(clean, fw)
    assert rv.saturated() == (0, 0)
    got = rv.predict_pair(sw, fw)
    # a STAGE is re-run whole (with the ramped schedule a stage is up to four launch groups)
    st = _stages(len(fw))
    hit = [i for i, (lo, hi) in enumerate(st) if any(lo <= g * 4096 < hi for g in (4, 13))]
    assert len(hit) == 2
    assert rv.saturated() == (0, len(hit))                         # exactly the spiked stages were re-run
    rv.set_precision("f32")
    ref32 = rv.predict_pair(sw, fw)
    rv.close()
    for g, b, r in zip(got, base, ref32):
        for i, (lo, hi) in enumerate(st):
            want = r if i in hit else b
            assert np.array_equal(g[lo:hi].view(np.uint8), want[lo:hi].view(np.uint8)), (i, lo, hi)


def test_raw_read_call_of_five_stages_equals_shorter_calls(species_models):
    """Read mode (device-formed windows, 16384 windows per stage): the five fixture reads twice in one call = five stages; window i
    of the concatenation depends on events i .. i + T - 1 only, so both halves must carry the bits of the five-read call."""
    import bench
    from nanoreviser_amd.engine import Reviser
    m1, m2 = species_models["human"]
    reads = bench.fixture_reads()

    def pack(rs):
        return [[r.raw for r in rs], [r.starts for r in rs], [r.feat_ev for r in rs], [r.shift for r in rs], [r.scale for r in rs]]
    rv = Reviser(m1, m2, precision="f16x2")
    five = rv.predict_reads_raw(*pack(reads))
    ten = rv.predict_reads_raw(*pack(reads + reads))
    N5 = sum(len(r.feat_ev) for r in reads)
    n5 = N5 - rv.T
    assert len(ten[2]) == 2 * N5 - rv.T and len(five[2]) == n5 and n5 > 2 * 16384
    for t, f in zip(ten, five):
        assert np.array_equal(t[:n5].view(np.uint8), f.view(np.uint8))
        assert np.array_equal(t[N5:N5 + n5].view(np.uint8), f.view(np.uint8))
    assert rv.saturated() == (0, 0)
    rv.close()


def test_two_raw_read_calls_in_flight_equal_the_calls_one_by_one(species_models):
    """nrv_reads_raw_begin / _end (r06): a raw-read call goes up in one transfer, all its stages are enqueued at once, and a
    second call may be enqueued before the first is collected.  Three different bundles, two in flight at any time, must come
    back with the bits of the same bundles run one by one - and with the bits of the per-event entry point (nrv_predict_read:
    host-side arrays through the staged pipeline) on the device's own segmentation of the same reads.  A third call in
    flight is refused, a ticket is good for one _end."""
    import bench
    from nanoreviser_amd.engine import Reviser
    m1, m2 = species_models["human"]
    reads = bench.fixture_reads()

    def pack(rs):
        return [[r.raw for r in rs], [r.starts for r in rs], [r.feat_ev for r in rs], [r.shift for r in rs], [r.scale for r in rs]]
    rv = Reviser(m1, m2, precision="f16x2")
    bundles = [reads[0:3], reads[2:5], reads + reads[:2]]
    one_by_one = [[x.copy() for x in rv.predict_reads_raw(*pack(b))] for b in bundles]
    pk = [rv.pack_reads_raw(*pack(b), rv.T) for b in bundles]
    tA = rv.begin_packed_raw(pk[0])
    tB = rv.begin_packed_raw(pk[1])
    with pytest.raises(RuntimeError, match="two calls are in flight"):
        rv.begin_packed_raw(rv.pack_reads_raw(*pack(bundles[2]), rv.T))
    outA = rv.end_packed_raw(tA)
    tC = rv.begin_packed_raw(pk[2])
    outB = rv.end_packed_raw(tB)
    outC = rv.end_packed_raw(tC)
    with pytest.raises(RuntimeError, match="no such call in flight"):
        rv.end_packed_raw(tC)
    for got, want in zip((outA, outB, outC), one_by_one):
        for g, w in zip(got, want):
            assert g.shape == w.shape and np.array_equal(g.view(np.uint8), w.view(np.uint8))
    # the per-event entry point on the same reads' device-cut windows: another pipeline, the same kernels on the same windows
    b = bundles[1]
    sig_ev = rv.segment_reads(*[pack(b)[i] for i in (0, 1, 3, 4)])
    feat = np.concatenate([r.feat_ev for r in b])
    per_event = rv.predict_read(sig_ev, feat)
    for g, w in zip(one_by_one[1], per_event):
        assert np.array_equal(g.view(np.uint8), w.view(np.uint8))
    assert rv.saturated() == (0, 0)
    # an empty call and a call shorter than a window go through both halves too
    short = rv.pack_reads_raw([reads[0].raw[:200]], [reads[0].starts[:5]], [reads[0].feat_ev[:5]], [reads[0].shift], [reads[0].scale], rv.T)
    assert all(len(x) == 0 for x in rv.end_packed_raw(rv.begin_packed_raw(short)))
    rv.close()
