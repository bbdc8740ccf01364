"""Host stage against vectors produced by RUNNING the reference's own
functions (tools/make_goldens.py): bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLD
from nanoreviser_amd import hoststage as hs


def test_collapse_events_matches_get_read_data(reads):
    # nanorev_fast5_handeler.py:39-150
    for key in reads.keys:
        g, rd, _ = reads(key)
        assert rd.abs_event_start == int(g["rd_abs_event_start"])
        assert np.array_equal(rd.start, g["rd_start"])
        assert np.array_equal(rd.length, g["rd_length"])
        assert np.array_equal(rd.bases, g["rd_bases"])
        assert np.array_equal(rd.ab_mean, g["rd_ab_mean"])
        assert np.array_equal(rd.ab_std, g["rd_ab_std"])


def test_signal_segmentation_bit_exact(reads):
    # preprocessing.py:85-170; full (N,50) f64 matrix compared through its sha256
    for key in reads.keys:
        g, rd, _ = reads(key)
        win, m, s, shift, scale = hs.signal_segmentation(
            rd.signal[rd.abs_event_start:], rd.start, int(rd.length[-1]))
        assert shift == float(g["seg_shift"]) and scale == float(g["seg_scale"])
        assert np.array_equal(m, g["seg_mean"])
        assert np.array_equal(s, g["seg_std"])
        assert win.shape == (len(rd.bases), 50)
        assert np.array_equal(win[g["seg_sig_rows"]], g["seg_sig_vals"])
        digest = hashlib.sha256(np.ascontiguousarray(win).tobytes()).hexdigest().encode()
        assert digest == bytes(g["seg_sig_sha256"])


def test_signal_segmentation_edge_padding():
    # reads shorter than a window, starts at both edges: symmetric zero pad, odd pad -> extra left
    raw = (np.arange(30) * 7 % 23 + 400).astype(np.int16)
    starts = np.array([0, 3, 11, 27])
    win, m, s, shift, scale = hs.signal_segmentation(raw, starts, 3)
    assert win.shape == (4, 50)
    norm = (raw.astype(float) - shift) / scale
    # base at st=3: samples [0, 28) -> 28 long, pad 22 -> 11 / 11
    assert np.array_equal(win[1, 11:39], norm[0:28]) and not win[1, :11].any() and not win[1, 39:].any()
    # base at st=11: samples [0, 30) -> pad 20 -> 10/10
    assert np.array_equal(win[2, 10:40], norm) and not win[2, :10].any()
    # base at st=27: samples [2, 30) = 28 -> 11/11 ; st=0: [0,25) = 25 -> pad 25 odd -> 13 left, 12 right
    assert np.array_equal(win[3, 11:39], norm[2:30])
    assert np.array_equal(win[0, 13:38], norm[0:25]) and not win[0, :13].any() and not win[0, 38:].any()
    assert m[0] == raw[0:3].mean() and s[3] == raw[27:30].astype(float).std()


def test_feature_rows_layout(reads):
    # nanorevtrainutils.py:162-169 column order; NanoReviser.py:124-125 scaling
    g, rd, rt = reads(reads.keys[0])
    assert rt.feat_ev.dtype == np.float32 and rt.feat_ev.shape == (len(rd.bases), 6)
    col = {"A": 250, "G": 180, "T": 100, "C": 30}
    exp0 = np.array([col[b.decode()] for b in rd.bases]) / 300.0
    assert np.array_equal(rt.feat_ev[:, 0], exp0.astype(np.float32))
    assert np.array_equal(rt.feat_ev[:, 1], (g["seg_mean"] / float(g["seg_shift"])).astype(np.float32))
    assert np.array_equal(rt.feat_ev[:, 2], (g["seg_std"] / float(g["seg_scale"])).astype(np.float32))
    assert np.array_equal(rt.feat_ev[:, 3], (g["rd_length"] / 10.0).astype(np.float32))
    assert np.array_equal(rt.feat_ev[:, 4], g["rd_ab_mean"])
    assert np.array_equal(rt.feat_ev[:, 5], g["rd_ab_std"])


def test_sliding_windows_match_training_builder(reads):
    # nanorevtrainutils.py:198-209: x[i:i+T], i in range(len(x) - T)
    _, _, rt = reads(reads.keys[1])
    T = 11
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, T)
    N = len(rt.feat_ev)
    assert sw.shape == (N - T, T, 50) and fw.shape == (N - T, T, 6)
    for i in (0, 1, 17, N - T - 1):
        assert np.array_equal(sw[i], rt.sig_ev[i:i + T]) and np.array_equal(fw[i], rt.feat_ev[i:i + T])
    s0, f0 = hs.sliding_windows(rt.sig_ev[:T], rt.feat_ev[:T], T)
    assert s0.shape == (0, T, 50) and f0.shape == (0, T, 6)


def test_trim_fastq(reads):
    # nanorev_fast5_handeler.py:152-171
    for key in reads.keys:
        g, _, _ = reads(key)
        b, q = hs.trim_fastq(bytes(g["fastq"]).decode("utf8"))
        assert b == bytes(g["fq_bases"]).decode() and q == bytes(g["fq_qual"]).decode()
    with pytest.raises(NotImplementedError):
        hs.trim_fastq("@x\nACGT\n+\n!!!!")


@pytest.fixture(scope="module")
def merge_vectors():
    return json.load(open(os.path.join(GOLD, "merge_vectors.json")))


def test_get_base_1_matches_reference(merge_vectors):
    # output_handeler.py:104-122 run on 41 vectors incl. zip truncation and the seeded first char
    for c in merge_vectors["get_base_1"]:
        assert hs.get_base_1(list(c["event_bases"]), c["y_pre"], c["y_pre2"]) == c["result"]


def test_writers_match_reference(merge_vectors):
    # output_handeler.py:26-62 (FASTQ keeps the reference's missing newline before '+')
    for w in merge_vectors["writers"]:
        assert hs.fasta_record(w["fast5_fn"], list(w["bases"])) == w["fasta"]
        assert hs.fastq_record(w["fast5_fn"], list(w["bases"]), list(w["qul"])) == w["fastq"]


def test_label_tables(merge_vectors):
    lab = merge_vectors["labels"]
    for b, v in lab["base_color"].items():
        assert hs.BASE_COLOR.get(b, 0) == v
    for b, v in lab["base_label"].items():
        assert hs.BASE_LABEL.get(b, 0) == v
    assert {str(k): v for k, v in hs.LABEL_TO_BASE.items()} == lab["label_to_base"]


def test_revise_read_rules():
    # window i revises base i+(T-1)//2; edges are kept
    bases = np.array(list("ACGTACGTACGTACG"), dtype="S1")     # N=15, T=11 -> 4 windows, offset 5
    lab = {"A": 5, "G": 4, "T": 3, "C": 2}
    T = 11
    same1 = [lab[b] for b in "CGTA"]                            # bases 5..8 = C G T A
    same2 = [v - 1 for v in same1]
    assert hs.revise_read(bases, same1, same2, T) == "ACGTACGTACGTACG"
    # substitution where both agree
    a1, a2 = list(same1), list(same2)
    a1[1], a2[1] = lab["T"], lab["T"] - 1
    assert hs.revise_read(bases, a1, a2, T) == "ACGTAC" + "T" + "TACGTACG"
    # deletion after base 5: m1 'D' (0), m2 says the missing base
    a1, a2 = list(same1), list(same2)
    a1[0], a2[0] = 0, lab["A"] - 1
    assert hs.revise_read(bases, a1, a2, T) == "ACGTA" + "CA" + "GTACGTACG"
    # insertion: both '-' drops the base
    a1, a2 = list(same1), list(same2)
    a1[2], a2[2] = 1, 0
    assert hs.revise_read(bases, a1, a2, T) == "ACGTACG" + "ACGTACG"
    # disagreement keeps the original
    a1, a2 = list(same1), list(same2)
    a1[3], a2[3] = lab["C"], lab["G"] - 1
    assert hs.revise_read(bases, a1, a2, T) == "ACGTACGTACGTACG"
    assert hs.revise_read(bases[:5], [], [], T) == "ACGTA"


def test_vectorised_merge_equals_rule_by_rule_loop():
    """hoststage.merge_calls/expand_calls (NumPy) against a plain per-window loop over the rules."""
    def loop(bases, a1, a2, T):
        b = [x.decode() for x in bases.tolist()]
        off, n, out = (T - 1) // 2, len(a1), []
        out += b[:off]
        for i in range(n):
            x, y = hs.LABEL_TO_BASE[int(a1[i])], hs.LABEL_TO_BASE[int(a2[i]) + 1]
            if x == y and x in "ATCG":
                out.append(x)
            elif x == "D" and y in "ATCG":
                out += [b[i + off], y]
            elif x == "-" and y == "-":
                continue
            else:
                out.append(b[i + off])
        return "".join(out + b[off + n:])

    rng = np.random.default_rng(0)
    for N in (12, 14, 30, 500, 5000):
        for T in (11, 13):
            if N <= T:
                continue
            bases = np.array(list("ACGT"), dtype="S1")[rng.integers(0, 4, N)]
            a1 = rng.integers(0, 6, N - T).astype(np.int8)
            a2 = rng.integers(0, 5, N - T).astype(np.int8)
            assert hs.revise_read(bases, a1, a2, T) == loop(bases, a1, a2, T)


def test_f32_window_fast_path_equals_cast_of_reference_matrix(reads):
    """read_tensors feeds the device from segment_windows_f32 (int16 -> f32 through a value table);
    it must be bit-identical to float32(reference f64 window matrix), the cast Keras does at feed."""
    for key in reads.keys:
        _, rd, rt = reads(key)
        sig = rd.signal[rd.abs_event_start:]
        win, m, s, shift, scale = hs.signal_segmentation(sig, rd.start, int(rd.length[-1]))
        assert rt.sig_ev.dtype == np.float32 and np.array_equal(rt.sig_ev, win.astype(np.float32))
        m2, s2, sh2, sc2 = hs.event_stats(sig, rd.start, int(rd.length[-1]))
        assert np.array_equal(m, m2) and np.array_equal(s, s2) and (sh2, sc2) == (shift, scale)
    raw = (np.arange(30) * 7 % 23 + 400).astype(np.int16)
    st = np.array([0, 3, 11, 27])
    w64, _, _, shift, scale = hs.signal_segmentation(raw, st, 3)
    assert np.array_equal(hs.segment_windows_f32(raw, st, shift, scale), w64.astype(np.float32))
    assert np.array_equal(hs.segment_windows_f32(raw.astype(np.float64), st, shift, scale), w64.astype(np.float32))


def test_median_mad_of_integer_samples_is_numpy_median_exactly():
    """hoststage.median_mad takes shift / scale of int16 samples from a histogram; they must be the SAME f64 numbers
    np.median gives on the f64 copy (preprocessing.py:100-101), for odd and even counts, half-integer shifts, the
    extremes of the type - and anything that is not a small integer type goes through np.median itself."""
    rng = np.random.default_rng(7)
    for trial in range(400):
        n = int(rng.integers(1, 60)) if trial % 2 else int(rng.integers(1000, 20000))
        lo = int(rng.integers(-32768, 32700))
        hi = int(rng.integers(lo + 1, min(lo + 1 + int(rng.integers(1, 2000)), 32768)))
        raw = rng.integers(lo, hi, n).astype(np.int16)
        r64 = raw.astype(np.float64)
        shift = np.median(r64)
        scale = np.median(np.abs(r64 - shift))
        got = hs.median_mad(raw)
        assert got[0] == shift and got[1] == scale and isinstance(got[0], np.float64), (n, lo, hi)
    for raw in (np.array([-32768, 32767], np.int16), np.array([5], np.int16), np.arange(256, dtype=np.uint8)):
        r64 = raw.astype(np.float64)
        assert hs.median_mad(raw) == (np.median(r64), np.median(np.abs(r64 - np.median(r64))))
    x = rng.normal(size=101)                      # not integer samples: the plain path
    assert hs.median_mad(x) == (np.median(x), np.median(np.abs(x - np.median(x))))


def _per_event_numpy(raw, st, last):
    r64 = np.asarray(raw).astype(np.float64)
    n = len(st)
    m, s = np.empty(n), np.empty(n)
    for i in range(n):
        seg = r64[st[i]:(st[i + 1] if i + 1 < n else st[i] + last)]
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                m[i], s[i] = (seg.mean(), seg.std()) if len(seg) else (np.nan, np.nan)
    return m, s


def test_native_event_stats_is_numpy_bit_for_bit():
    """libnanorev_host.so (include/nanorev_host.h): per-event mean / std with NumPy's own summation order - every
    branch of it (fewer than 8 values, the eight running sums up to 128, the halving above) - equals np.mean / np.std
    of the slice exactly, as do the grouped NumPy formulation it replaces and the clipped / empty ranges."""
    from nanoreviser_amd import hostlib
    import __graft_entry__ as g
    g.build_host()
    hostlib._tried = False
    assert hostlib.load() is not None, "libnanorev_host.so must build with gcc"
    rng = np.random.default_rng(11)
    n_events = 0
    for trial in range(60):
        nev = int(rng.integers(1, 120))
        lens = rng.integers(1, [12, 40, 140, 300, 1200][trial % 5], nev)
        st = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        L = max(int(lens.sum()) - int(rng.integers(0, 3)) * (trial % 2), 1)      # sometimes the last event is clipped
        lo = int(rng.integers(-32768, 32000))
        raw = rng.integers(lo, min(lo + 1 + int(rng.integers(1, 4000)), 32768), L).astype(np.int16)
        got = hostlib.event_stats(raw, st, int(lens[-1]))
        ref = _per_event_numpy(raw, st, int(lens[-1]))
        assert np.array_equal(got[0], ref[0], equal_nan=True) and np.array_equal(got[1], ref[1], equal_nan=True)
        en = np.append(st[1:], st[-1] + int(lens[-1]))
        old = hs._event_mean_std(raw.astype(np.float64), st, en)
        assert np.array_equal(got[0], old[0], equal_nan=True) and np.array_equal(got[1], old[1], equal_nan=True)
        n_events += nev
    assert n_events > 2000
    # an event wholly past the samples: NaN, like the empty slice
    m, s = hostlib.event_stats(np.array([1, 2, 3], np.int16), np.array([0, 2, 5]), 4)
    assert m[0] == 1.5 and m[1] == 3.0 and np.isnan(m[2]) and np.isnan(s[2])
    # what the helper does not take goes back to NumPy: float samples, starts beyond int32
    assert hostlib.event_stats(np.array([1.0, 2.0]), np.array([0]), 2) is None
    assert hostlib.event_stats(np.array([1, 2], np.int16), np.array([0, 2 ** 31]), 2) is None


def test_event_stats_same_with_and_without_the_native_helper(monkeypatch):
    from nanoreviser_amd import hostlib
    rng = np.random.default_rng(12)
    lens = rng.integers(3, 60, 500)
    st = np.concatenate([[0], np.cumsum(lens)[:-1]])
    raw = rng.integers(300, 900, int(lens.sum())).astype(np.int16)
    hostlib._tried = False
    a = hs.event_stats(raw, st, int(lens[-1]))
    monkeypatch.setenv("NRV_HOST_LIB", "0")
    hostlib._tried, hostlib._lib = False, None
    b = hs.event_stats(raw, st, int(lens[-1]))
    hostlib._tried = False
    assert all(np.array_equal(x, y) for x, y in zip(a[:2], b[:2])) and a[2:] == b[2:]
