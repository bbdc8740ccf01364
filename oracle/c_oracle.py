"""ORACLE - TEST INFRASTRUCTURE ONLY.  ctypes loader of oracle/_build/libnrv_oracle.so
(the plain-C f32 restatement, oracle/nrv_oracle.c).  Used by tests/ and by bench.py's
`cpu_baseline` leg; never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_DIR, "_build", "libnrv_oracle.so")
_lib = None


def load(build=True):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB) and build:
        subprocess.run(["make", "-C", _DIR], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(LIB)
    fp, i8p = C.POINTER(C.c_float), C.POINTER(C.c_int8)
    for name in ("nrvo_predict", "nrvo_predict_read"):
        f = getattr(lib, name)
        f.argtypes = [fp, C.c_int64, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int64, fp, i8p, C.c_int]
        f.restype = C.c_int
    lib.nrvo_n_params.argtypes = [C.c_int, C.c_int]
    lib.nrvo_n_params.restype = C.c_int64
    _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def predict(flat_weights, T, n_class, signal, read, act=0, threads=1):
    lib = load()
    w = np.ascontiguousarray(flat_weights, np.float32)
    read = np.ascontiguousarray(read, np.float32)
    n = read.shape[0]
    signal = np.ascontiguousarray(signal, np.float32).reshape(n, T, 50)
    prob = np.empty((n, n_class), np.float32)
    am = np.empty(n, np.int8)
    rc = lib.nrvo_predict(_fp(w), w.size, T, n_class, act, _fp(signal), _fp(read), n, _fp(prob),
                          am.ctypes.data_as(C.POINTER(C.c_int8)), threads)
    if rc:
        raise RuntimeError(f"nrvo_predict rc={rc}")
    return prob, am


def predict_read(flat_weights, T, n_class, sig_ev, feat_ev, act=0, threads=1):
    lib = load()
    w = np.ascontiguousarray(flat_weights, np.float32)
    sig_ev = np.ascontiguousarray(sig_ev, np.float32)
    feat_ev = np.ascontiguousarray(feat_ev, np.float32)
    N = feat_ev.shape[0]
    n = max(N - T, 0)
    prob = np.empty((n, n_class), np.float32)
    am = np.empty(n, np.int8)
    rc = lib.nrvo_predict_read(_fp(w), w.size, T, n_class, act, _fp(sig_ev), _fp(feat_ev), N, _fp(prob),
                               am.ctypes.data_as(C.POINTER(C.c_int8)), threads)
    if rc:
        raise RuntimeError(f"nrvo_predict_read rc={rc}")
    return prob, am
