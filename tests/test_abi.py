"""C-ABI surface (CPU box: no compute calls).  The library must load, export every symbol that
include/nanorev.h declares, reject bad arguments with the documented codes, and FAIL LOUDLY when
no GPU is present - there is no CPU fallback to hide behind."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from nanoreviser_amd import engine
from nanoreviser_amd.weights import load_species

HEADER = os.path.join(ROOT, "include", "nanorev.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nrv_[a-z_0-9]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(engine.LIB_PATH):
        import __graft_entry__ as g
        g.build_hip()
    return engine.load_library()


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nanorev.h but not exported"
    assert sorted(engine.SYMBOLS) == names


def test_product_library_has_no_diagnostic_entry_points(lib):
    """The stamp / stage-selector entry points (nrv_exp_*) exist only in diagnostic builds (-DNRV_STAMP=1, tools/lstm_exp.sh)."""
    out = os.popen(f"nm -D --defined-only {engine.LIB_PATH}").read()
    exported = sorted(set(re.findall(r"\b(nrv_[a-z_0-9]+)\b", out)))
    assert exported, "nm found no nrv_* symbol"
    assert not [n for n in exported if n.startswith("nrv_exp_")], exported
    assert set(exported) == set(_declared()), sorted(set(exported) ^ set(_declared()))


def test_host_helper_header_symbols_exported():
    """include/nanorev_host.h <-> libnanorev_host.so (plain C, gcc): every declared nrvh_* symbol is exported."""
    import __graft_entry__ as g
    from nanoreviser_amd import hostlib
    g.build_host()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nanorev_host.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(nrvh_[a-z_0-9]+)\s*\(", txt)))
    assert names == sorted(hostlib.SYMBOLS) and len(names) >= 2
    lib = C.CDLL(hostlib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nanorev_host.h but not exported"
    assert "libamdhip64" not in os.popen(f"readelf -d {hostlib.LIB_PATH}").read()      # host only


def test_no_oracle_or_cpu_path_in_product():
    """The product package must not import or link anything under oracle/."""
    pkg = os.path.join(ROOT, "nanoreviser_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src, fn
                assert "nrv_oracle" not in src and "libnrv_oracle" not in src, fn
    out = os.popen(f"readelf -d {engine.LIB_PATH}").read()
    assert "nrv_oracle" not in out and "libamdhip64" in out


def _weights(T=11):
    m1, m2 = load_species("ecoli")
    f1, f2 = m1.with_window(T).flat(), m2.with_window(T).flat()
    w1 = engine._Weights(f1.ctypes.data_as(C.POINTER(C.c_float)), f1.size)
    w2 = engine._Weights(f2.ctypes.data_as(C.POINTER(C.c_float)), f2.size)
    return (f1, f2), w1, w2


def test_create_argument_errors(lib):
    keep, w1, w2 = _weights()
    h = C.c_void_p()
    assert lib.nrv_create(C.byref(w1), C.byref(w2), 11, 0, 0, None) == -1
    assert lib.nrv_create(C.byref(w1), C.byref(w2), 0, 0, 0, C.byref(h)) == -1
    assert lib.nrv_create(C.byref(w1), C.byref(w2), 99, 0, 0, C.byref(h)) == -1
    assert lib.nrv_create(C.byref(w1), C.byref(w2), 11, 0, 7, C.byref(h)) == -1
    assert b"recurrent_act" in lib.nrv_last_error(None)
    # blob length does not match the graph at this T / swapped models -> NRV_E_WEIGHTS
    assert lib.nrv_create(C.byref(w1), C.byref(w2), 13, 0, 0, C.byref(h)) == -2
    assert lib.nrv_create(C.byref(w2), C.byref(w1), 11, 0, 0, C.byref(h)) == -2
    assert b"model1" in lib.nrv_last_error(None)
    assert not h.value
    assert lib.nrv_backend(None) == 1
    assert lib.nrv_kernel_name(3).startswith(b"lstm3")
    lib.nrv_destroy(None)            # no-op


def test_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    keep, w1, w2 = _weights()
    h = C.c_void_p()
    rc = lib.nrv_create(C.byref(w1), C.byref(w2), 11, 0, 0, C.byref(h))
    assert rc in (-3, -4) and not h.value
    assert b"no CPU fallback" in lib.nrv_last_error(None) or rc == -4
    m1, m2 = load_species("ecoli")
    with pytest.raises(engine.NrvError):
        engine.Reviser(m1, m2)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(OSError):
        engine.load_library(str(tmp_path / "libnanorev_hip.so"))


def test_wrapper_validates_shapes():
    m1, m2 = load_species("ecoli")
    with pytest.raises(ValueError):
        engine.Reviser(m1, m2.with_window(13))
    with pytest.raises(ValueError):
        engine.Reviser(m1, m2, recurrent_activation="relu")
    with pytest.raises(ValueError):
        engine._as_f32(np.zeros((4, 11, 5)), (11, 6))
