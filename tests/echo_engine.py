"""Stand-in engines for the command-line tests (host logic only, no compute).  Module-level so that
spawned worker processes can unpickle the factories."""
import os

import numpy as np


class EchoEngine:
    """Stand-in for engine.Reviser in host-logic tests: 'predicts' exactly the original base at
    every window centre (model1 label, model2 label-1), so revise_read must return the input."""
    T = 11

    def __init__(self, fail_marker=None):
        self.fail_marker, self.calls = fail_marker, 0

    def predict_read(self, sig_ev, feat_ev):
        self.calls += 1
        # fails on every call whose first event is the marked read's first event: the batched call
        # (marked read first) AND the per-read retry of that read, but not the other reads' retries
        if self.fail_marker is not None and np.array_equal(feat_ev[0], self.fail_marker):
            raise RuntimeError("injected engine failure")
        assert sig_ev.dtype == np.float32 and sig_ev.shape[1] == 50 and feat_ev.shape[1] == 6
        n = len(feat_ev) - self.T
        col = np.rint(feat_ev[:, 0] * 300).astype(int)                 # 250/180/100/30 -> A/G/T/C
        lab = np.select([col == 250, col == 180, col == 100, col == 30], [5, 4, 3, 2])
        a1 = lab[5:5 + n].astype(np.int8)
        p1 = np.eye(6, dtype=np.float32)[a1] * 0.9 + 0.1 / 6
        p2 = np.eye(5, dtype=np.float32)[a1 - 1] * 0.9 + 0.1 / 5
        return p1, p2, a1, (a1 - 1).astype(np.int8)

    def close(self):
        pass

    def predict_reads_raw(self, raws, starts, feats, shifts, scales):
        """What the CLI's workers hand over (engine.Reviser.predict_reads_raw): cut the windows with
        the host stage here and go through the same echo."""
        from nanoreviser_amd import hoststage as hs
        for r, s in zip(raws, starts):
            assert r.dtype == np.int16 and s.dtype == np.int32
        sig = np.concatenate([hs.segment_windows_f32(r, s, sh, sc) for r, s, sh, sc in zip(raws, starts, shifts, scales)])
        return self.predict_read(sig, np.concatenate(feats))


class PackedEcho(EchoEngine):
    """EchoEngine with the packed raw-read surface of engine.Reviser (pack_bundle / run_packed_raw): what the command
    line's native host stage drives - one call per bundle of reads, outputs for the concatenated event range."""

    @staticmethod
    def pack_bundle(raw, starts, feat, meta, T):
        from nanoreviser_amd.engine import Reviser
        return Reviser.pack_bundle(raw, starts, feat, meta, T)

    def run_packed_raw(self, packed):
        raw, st, feat, descs, nr, N, (p1, p2, a1, a2) = packed
        self.calls += 1
        if self.fail_marker is not None and np.array_equal(feat[0], self.fail_marker):
            raise RuntimeError("injected engine failure")
        n = max(N - self.T, 0)
        col = np.rint(feat[:, 0] * 300).astype(int)
        lab = np.select([col == 250, col == 180, col == 100, col == 30], [5, 4, 3, 2])
        a1[:] = lab[5:5 + n]
        a2[:] = a1 - 1
        p1[:] = np.eye(6, dtype=np.float32)[a1] * 0.9 + 0.1 / 6
        p2[:] = np.eye(5, dtype=np.float32)[a2] * 0.9 + 0.1 / 5
        return p1, p2, a1, a2


class PipelinedEcho(PackedEcho):
    """PackedEcho with the two-halves surface of engine.Reviser (begin_packed_raw / end_packed_raw, r06): what the command
    line pipelines two deep.  Keeps the contract of the native handle: at most two calls in flight, a ticket is good once,
    nothing else may run on the engine between a call's halves.  `fail_in_end`: the call whose first event matches
    fail_marker fails in its SECOND half (the first half only enqueues)."""

    def __init__(self, fail_marker=None, fail_in_end=False):
        super().__init__(None if fail_in_end else fail_marker)
        self.end_marker = fail_marker if fail_in_end else None
        self.flight, self.max_in_flight, self.begun, self.violations = {}, 0, 0, []

    def begin_packed_raw(self, packed):
        if len(self.flight) >= 2:
            raise RuntimeError("two calls are in flight already")
        out = PackedEcho.run_packed_raw(self, packed)      # raises here for a fail_marker of the first half
        self.begun += 1
        self.flight[self.begun] = (packed[2][0].copy() if len(packed[2]) else None, out)
        self.max_in_flight = max(self.max_in_flight, len(self.flight))
        return self.begun, out

    def end_packed_raw(self, ticket):
        t, out = ticket
        first, out2 = self.flight.pop(t)                   # KeyError: a ticket used twice
        if t != min([t] + list(self.flight)):
            self.violations.append(("out of order", t))
        if self.end_marker is not None and first is not None and np.array_equal(first, self.end_marker):
            raise RuntimeError("injected engine failure in the second half")
        return out

    def run_packed_raw(self, packed):
        if self.flight:
            self.violations.append(("synchronous call between the halves of another", len(self.flight)))
        return super().run_packed_raw(packed)

    def predict_read(self, sig_ev, feat_ev):
        if self.flight:
            self.violations.append(("per-read call between the halves of another", len(self.flight)))
        return super().predict_read(sig_ev, feat_ev)


def echo_factory(args, device):
    return EchoEngine()


class HashEngine(EchoEngine):
    """Calls that depend on the window's CENTRE EVENT alone (a hash of its feature row), all classes of both models
    included: deletions, insertions, disagreements.  Whatever way a read is cut into device calls, its revised text is
    the same - unless a slice is lost, doubled, shifted or out of order, which the echo (always the original base) hides."""

    def predict_read(self, sig_ev, feat_ev):
        self.calls += 1
        n = max(len(feat_ev) - self.T, 0)
        h = (np.ascontiguousarray(feat_ev[:, 1:4], np.float32).view(np.uint32).astype(np.uint64) *
             np.array([2654435761, 40503, 2246822519], np.uint64)).sum(1)[5:5 + n]
        a1, a2 = ((h >> np.uint64(7)) % np.uint64(6)).astype(np.int8), ((h >> np.uint64(13)) % np.uint64(5)).astype(np.int8)
        p1 = np.eye(6, dtype=np.float32)[a1] * ((h % np.uint64(89)).astype(np.float32)[:, None] / 100 + 0.1)
        p2 = np.eye(5, dtype=np.float32)[a2] * ((h % np.uint64(83)).astype(np.float32)[:, None] / 100 + 0.1)
        return p1, p2, a1, a2


def hash_factory(args, device):
    return HashEngine()


class _HashDiesOnSlice(HashEngine):
    def predict_reads_raw(self, raws, starts, feats, shifts, scales):
        if len(raws) == 1 and len(starts[0]) < 6000:         # a slice of a split read (whole fixture reads are longer)
            os._exit(134)
        return super().predict_reads_raw(raws, starts, feats, shifts, scales)


def hash_dies_on_slice_factory(args, device):
    return _HashDiesOnSlice() if device == 1 else HashEngine()


def dying_factory(args, device):
    """Worker on 'GPU' 1 dies before it has an engine."""
    if device == 1:
        os._exit(1)
    return EchoEngine()


class _DiesInPredict(EchoEngine):
    def predict_read(self, sig_ev, feat_ev):
        os._exit(134)                      # what an abort() inside the native library looks like


def dying_midway_factory(args, device):
    return _DiesInPredict() if device == 1 else EchoEngine()


def broken_factory(args, device):
    if device == 1:
        raise RuntimeError("libnanorev_hip: error -3: no HIP device")
    return EchoEngine()


class _FailsFirstReadThenDies(EchoEngine):
    """Fails every call that starts with the first read it ever saw (so the batched call and that read's
    retry fail, the other read of the batch is revised), then dies hard on its 4th call - the second
    batch - after giving the queue's feeder thread time to flush the per-file records."""
    def predict_read(self, sig_ev, feat_ev):
        import time
        if self.fail_marker is None:
            self.fail_marker = feat_ev[0].copy()
        if self.calls >= 3:
            time.sleep(1.0)
            os._exit(134)
        return super().predict_read(sig_ev, feat_ev)


def fails_then_dies_factory(args, device):
    return _FailsFirstReadThenDies() if device == 1 else EchoEngine()


def shared_device_factory(args, device):
    """The REAL engine for every worker rank, all on device 0 (a 1-GPU box rehearsing the N-worker path)."""
    from nanoreviser_amd import cli
    return cli._default_factory(args, 0)


class _ExclusiveEcho(EchoEngine):
    """An engine is one native handle: never inside two calls at once.  Counts the engines made and the violations."""
    made = []
    violations = []

    def __init__(self):
        super().__init__()
        import threading
        self._in_call = threading.Lock()
        _ExclusiveEcho.made.append(self)

    def predict_read(self, sig_ev, feat_ev):
        import time
        if not self._in_call.acquire(blocking=False):
            _ExclusiveEcho.violations.append(id(self))
            raise RuntimeError("engine entered by two threads at once")
        try:
            time.sleep(0.02)                               # long enough for the calls of two engine threads to overlap
            return super().predict_read(sig_ev, feat_ev)
        finally:
            self._in_call.release()


def exclusive_factory(args, device):
    return _ExclusiveEcho()
