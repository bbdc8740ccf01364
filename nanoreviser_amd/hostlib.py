"""libnanorev_host.so: the host stage's native helpers (include/nanorev_host.h; plain C, no HIP, no GPU).

Built by `__graft_entry__.build()` next to the engine library.  The host stage uses it when it is there and runs the
NumPy formulation of the same arithmetic when it is not - the two give the same numbers bit for bit
(tests/test_hoststage_golden.py), so this is a speed path, not a fallback of the device path."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnanorev_host.so")
SYMBOLS = ["nrvh_abi_version", "nrvh_event_stats", "nrvh_load_fast5", "nrvh_free_read", "nrvh_load_bundle",
           "nrvh_free_bundle", "nrvh_finish_read", "nrvh_finish_bundle"]
_lib = None
_tried = False


class _NativeRead(C.Structure):               # include/nanorev_host.h: nrvh_read
    _fields_ = [("n_raw", C.c_int64), ("n_ev", C.c_int64), ("raw", C.POINTER(C.c_int16)), ("starts", C.POINTER(C.c_int32)),
                ("feat", C.POINTER(C.c_float)), ("bases", C.POINTER(C.c_char)), ("shift", C.c_double), ("scale", C.c_double),
                ("fastq", C.POINTER(C.c_char)), ("fastq_len", C.c_int64)]


class _NativeBundle(C.Structure):             # include/nanorev_host.h: nrvh_bundle
    _fields_ = [("n_files", C.c_int32), ("n_ok", C.c_int32), ("n_raw", C.c_int64), ("n_ev", C.c_int64),
                ("raw", C.POINTER(C.c_int16)), ("starts", C.POINTER(C.c_int32)), ("feat", C.POINTER(C.c_float)),
                ("bases", C.POINTER(C.c_char)), ("meta", C.POINTER(C.c_double)), ("status", C.POINTER(C.c_int32)),
                ("fastq", C.POINTER(C.c_char)), ("fastq_off", C.POINTER(C.c_int64)), ("errors", C.POINTER(C.c_char))]


OK, UNSUPPORTED, E_READ, E_IO, E_ARG = 0, 1, 2, 3, 4
ERR_LEN = 96


def load() -> Optional[C.CDLL]:
    """The library, or None when it has not been built (NRV_HOST_LIB=0 disables it: the NumPy path runs)."""
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("NRV_HOST_LIB", "1") == "0" or not os.path.exists(LIB_PATH):
        return None
    try:
        lib = C.CDLL(LIB_PATH)
        lib.nrvh_abi_version.restype = C.c_int
        if lib.nrvh_abi_version() != 2:
            return None
        lib.nrvh_load_fast5.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_NativeRead), C.c_char_p, C.c_int]
        lib.nrvh_load_fast5.restype = C.c_int
        lib.nrvh_free_read.argtypes = [C.POINTER(_NativeRead)]
        lib.nrvh_free_read.restype = None
        lib.nrvh_load_bundle.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_NativeBundle)]
        lib.nrvh_load_bundle.restype = C.c_int
        lib.nrvh_free_bundle.argtypes = [C.POINTER(_NativeBundle)]
        lib.nrvh_free_bundle.restype = None
        lib.nrvh_finish_read.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]
        lib.nrvh_finish_read.restype = C.c_int
        lib.nrvh_finish_bundle.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                           C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int, C.c_void_p, C.c_void_p]
        lib.nrvh_finish_bundle.restype = C.c_int
        lib.nrvh_event_stats.argtypes = [C.POINTER(C.c_int16), C.c_int64, C.POINTER(C.c_int32), C.c_int64, C.c_int32,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.nrvh_event_stats.restype = C.c_int
        _lib = lib
    except OSError:
        _lib = None
    return _lib


def event_stats(raw_signal, starts, last_dur):
    """(mean[N], std[N]) in f64 of raw[start_i:start_{i+1}] per event (np.mean / np.std semantics), or None when the
    helper does not apply: library absent, samples not a <= 16-bit integer type, starts beyond int32."""
    lib = load()
    raw = np.asarray(raw_signal)
    st = np.asarray(starts)
    if lib is None or raw.ndim != 1 or st.ndim != 1 or raw.dtype.kind not in "iu" or raw.dtype.itemsize > 2:
        return None
    if raw.dtype != np.int16:
        if raw.size and (int(raw.max()) > 32767 or int(raw.min()) < -32768):      # uint16 above the int16 range
            return None
        raw = raw.astype(np.int16)
    n = st.shape[0]
    if n and (int(st.max()) + int(last_dur) >= 2 ** 31 or int(st.min()) < 0 or int(last_dur) < 0):
        return None
    raw = np.ascontiguousarray(raw)
    st32 = np.ascontiguousarray(st, dtype=np.int32)
    mean = np.empty(n, np.float64)
    std = np.empty(n, np.float64)
    rc = lib.nrvh_event_stats(raw.ctypes.data_as(C.POINTER(C.c_int16)), raw.size,
                              st32.ctypes.data_as(C.POINTER(C.c_int32)), n, int(last_dur),
                              mean.ctypes.data_as(C.POINTER(C.c_double)), std.ctypes.data_as(C.POINTER(C.c_double)))
    if rc != 0:
        return None
    return mean, std


def load_fast5(path: str, group: str, subgroup: str, want_fastq: bool = True):
    """One fast5 file through the native host stage (csrc/nrv_host_fast5.c; the GIL is released for the whole call).
    Returns (code, payload): code OK -> payload = dict(raw int16[L], starts int32[N], feat float32[N,6], bases S1[N],
    shift, scale, fastq str | None); any other code -> payload = the reader's reason, and the caller runs the Python
    host stage (h5lite + hoststage), which is the definition of every number here.  (None, ...) without the library."""
    lib = load()
    if lib is None:
        return None, "libnanorev_host.so not built"
    r = _NativeRead()
    err = C.create_string_buffer(160)
    rc = lib.nrvh_load_fast5(os.fsencode(path), group.encode(), subgroup.encode(), 1 if want_fastq else 0, C.byref(r), err, 160)
    if rc != OK:
        return rc, err.value.decode("utf8", "replace")
    try:
        n, L = int(r.n_ev), int(r.n_raw)
        out = {"raw": _arr(r.raw, L, np.int16), "starts": _arr(r.starts, n, np.int32),
               "feat": _arr(r.feat, n * 6, np.float32).reshape(n, 6), "bases": _arr(r.bases, n, "S1"),
               "shift": float(r.shift), "scale": float(r.scale),
               "fastq": C.string_at(r.fastq, int(r.fastq_len)).decode("utf8", "replace") if r.fastq else None}
    finally:
        lib.nrvh_free_read(C.byref(r))
    return OK, out


def _arr(ptr, count, dtype):
    """One copy out of a C buffer: a read-only array over the bytes."""
    return np.frombuffer(C.string_at(ptr, count * np.dtype(dtype).itemsize), dtype=dtype) if count else np.zeros(0, dtype)


def load_bundle(paths, group: str, subgroup: str, want_fastq: bool = True):
    """Several fast5 files in ONE native call (GIL released throughout): dict(status int32[n], errors [str], raw, starts,
    feat (E,6), bases S1[E], meta float64[n,4] = (raw_len, ev_len, shift, scale) per file, fastq [str | None]) over the
    reads whose status is OK, concatenated in file order; None without the library."""
    lib = load()
    if lib is None:
        return None
    n = len(paths)
    arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    mem = _BundleMem(lib)
    b = mem.b
    rc = lib.nrvh_load_bundle(arr, n, group.encode(), subgroup.encode(), 1 if want_fastq else 0, C.byref(b))
    if rc != OK:
        return None
    mem.live = True
    E = int(b.n_ev)
    status = _arr(b.status, n, np.int32)
    errs = C.string_at(b.errors, n * ERR_LEN) if (status != OK).any() else b""
    fastq = [None] * n
    if want_fastq:
        foff = _arr(b.fastq_off, n + 1, np.int64)
        ftxt = C.string_at(b.fastq, int(foff[n])) if int(foff[n]) else b""
        for i in range(n):
            if status[i] == OK and foff[i] >= 0:
                j = i + 1
                while j < n and foff[j] < 0:
                    j += 1
                fastq[i] = ftxt[int(foff[i]):int(foff[j])].decode("utf8", "replace")
    # the big arrays are NOT copied: they are views of the C buffers, which live until the last view is gone
    return {"status": status,
            "errors": [errs[i * ERR_LEN:(i + 1) * ERR_LEN].split(b"\0")[0].decode("utf8", "replace") if errs else "" for i in range(n)],
            "raw": mem.view(b.raw, int(b.n_raw), np.int16), "starts": mem.view(b.starts, E, np.int32),
            "feat": mem.view(b.feat, E * 6, np.float32).reshape(E, 6), "bases": mem.view(b.bases, E, "S1"),
            "meta": _arr(b.meta, n * 4, np.float64).reshape(n, 4), "fastq": fastq}


class _BundleMem:
    """Owner of one nrvh_bundle's C buffers: released when the last array that views them has gone."""

    def __init__(self, lib):
        self.lib, self.b, self.live = lib, _NativeBundle(), False

    def view(self, ptr, count, dtype):
        if not count:
            return np.zeros(0, dtype)
        nbytes = count * np.dtype(dtype).itemsize
        buf = (C.c_char * nbytes).from_address(C.addressof(ptr.contents))
        buf._owner = self                               # numpy keeps `buf` as the array's base, `buf` keeps us
        a = np.frombuffer(buf, dtype=dtype)
        a.flags.writeable = False
        return a

    def __del__(self):
        if self.live:
            self.live = False
            self.lib.nrvh_free_bundle(C.byref(self.b))


def finish_read(bases, a1, a2, T: int, qc, name: str, dst: str, fastq: bool):
    """The calls of one read -> its revised record -> the file `dst` (written atomically), in one native call.
    bases: S1[N]; a1, a2: int8[n]; qc: uint8[n] Phred characters or None.  Returns the length of the revised sequence;
    raises OSError / ValueError on failure; None without the library."""
    lib = load()
    if lib is None:
        return None
    bb = np.ascontiguousarray(bases, dtype="S1")
    x1, x2 = np.ascontiguousarray(a1, dtype=np.int8), np.ascontiguousarray(a2, dtype=np.int8)
    q = np.ascontiguousarray(qc, dtype=np.uint8) if qc is not None else None
    if len(x1) != len(x2) or (q is not None and len(q) != len(x1)):
        raise ValueError("finish_read: calls of different lengths")
    nw = C.c_int64(0)
    rc = lib.nrvh_finish_read(bb.tobytes(), len(bb), x1.ctypes.data, x2.ctypes.data, len(x1), int(T),
                              q.ctypes.data if q is not None else None, name.encode("utf8"), os.fsencode(dst),
                              1 if fastq else 0, C.byref(nw))
    if rc == E_ARG:
        raise ValueError("finish_read: bad arguments (more windows than bases?)")
    if rc != OK:
        raise OSError(f"finish_read: cannot write {dst}")
    return int(nw.value)


def finish_bundle(bases, ev_len, a1, a2, T: int, qc, names, dsts, fastq: bool):
    """`finish_read` for all reads of one device call in ONE native call (bases S1[E] concatenated, ev_len int64[R],
    a1 / a2 int8[E - T] and qc uint8[E - T] | None as the call returned them).  Returns (n_written int64[R],
    status int32[R]); None without the library."""
    lib = load()
    if lib is None:
        return None
    bb = np.ascontiguousarray(bases, dtype="S1")
    el = np.ascontiguousarray(ev_len, dtype=np.int64)
    x1, x2 = np.ascontiguousarray(a1, dtype=np.int8), np.ascontiguousarray(a2, dtype=np.int8)
    q = np.ascontiguousarray(qc, dtype=np.uint8) if qc is not None else None
    R = len(el)
    if int(el.sum()) != len(bb) or len(x1) != len(x2) or (q is not None and len(q) != len(x1)) or len(names) != R or len(dsts) != R:
        raise ValueError("finish_bundle: arrays do not match the read table")
    nm = (C.c_char_p * R)(*[n.encode("utf8") for n in names])
    ds = (C.c_char_p * R)(*[os.fsencode(d) for d in dsts])
    nw, st = np.zeros(R, np.int64), np.zeros(R, np.int32)
    rc = lib.nrvh_finish_bundle(bb.ctypes.data, el.ctypes.data, R, x1.ctypes.data, x2.ctypes.data, len(x1), int(T),
                                q.ctypes.data if q is not None else None, nm, ds, 1 if fastq else 0, nw.ctypes.data, st.ctypes.data)
    if rc != OK:
        raise ValueError("finish_bundle: bad arguments")
    return nw, st
