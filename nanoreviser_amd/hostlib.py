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
SYMBOLS = ["nrvh_abi_version", "nrvh_event_stats"]
_lib = None
_tried = False


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
        if lib.nrvh_abi_version() != 1:
            return None
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
