"""Weight files of the reviser models.

The reference keeps its trained parameters in Keras `save_weights` HDF5 files
`./model/<species>/<species>_win13_50ep_model{1,2}.h5` (path convention:
NanoReviser.py:191-193; writer: NanoReviser_train.py:175-176,203-204).  Keras
loads them *positionally* (root attr `layer_names` x per-layer `weight_names`),
and so do we: a model is a list of 60 f32 tensors whose order and shapes are
fixed by the graph of output_handeler.py:206-307.

Two on-disk forms are accepted:
  * `<stem>.f32` (+ optional `<stem>.json` manifest) - the flat little-endian f32
    concatenation written by tools/convert_weights.py;
  * `<stem>.h5` - read with the minimal HDF5 reader in `h5lite.py` (no h5py).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

N_TENSORS = 60
SIGNAL_LEN = 50          # output_handeler.py:202  SIGNEL_LEN
VEC_LEN = 6              # output_handeler.py:203
N_CLASS_M1 = 6           # output_handeler.py:200/237
N_CLASS_M2 = 5           # output_handeler.py:289

ROLES: List[str] = (
    ["conv1.kernel", "conv1.bias", "bn1.gamma", "bn1.beta", "bn1.mean", "bn1.var",
     "conv2.kernel", "conv2.bias", "bn2.gamma", "bn2.beta", "bn2.mean", "bn2.var"]
    + [f"lstm1.{d}.{w}" for d in ("fw", "bw") for w in ("kernel", "recurrent", "bias")]
    + ["bn_l1.gamma", "bn_l1.beta", "bn_l1.mean", "bn_l1.var"]
    + [f"lstm2.{d}.{w}" for d in ("fw", "bw") for w in ("kernel", "recurrent", "bias")]
    + ["bn_l2.gamma", "bn_l2.beta", "bn_l2.mean", "bn_l2.var"]
    + ["sig_dense.kernel", "sig_dense.bias"]
    + [f"lstm3.{d}.{w}" for d in ("fw", "bw") for w in ("kernel", "recurrent", "bias")]
    + ["bn_l3.gamma", "bn_l3.beta", "bn_l3.mean", "bn_l3.var"]
    + [f"lstm4.{d}.{w}" for d in ("fw", "bw") for w in ("kernel", "recurrent", "bias")]
    + ["dense1.kernel", "dense1.bias", "dense2.kernel", "dense2.bias",
       "main_out.kernel", "main_out.bias", "feature.kernel", "feature.bias",
       "final_out.kernel", "final_out.bias"]
)


def tensor_shapes(T: int, n_class: int) -> List[tuple]:
    """Shapes of the 60 positional tensors for window length T."""
    def bn(c):
        return [(c,)] * 4

    def bilstm(d, h):
        return [(d, 4 * h), (h, 4 * h), (4 * h,)] * 2

    s = [(3, 1, 8), (8,)] + bn(8) + [(3, 8, 8), (8,)] + bn(8)
    s += bilstm(6, 16) + bn(32)
    s += bilstm(32, 64) + bn(128)
    s += [(400, 64), (64,)]
    s += bilstm(192, 128) + bn(256)
    s += bilstm(256, 64)
    s += [(128, 128), (128,), (128, 32), (32,), (32, 6), (6,)]
    s += [(6 * T, 16), (16,), (16, n_class), (n_class,)]
    assert len(s) == N_TENSORS
    return s


def n_params(T: int, n_class: int) -> int:
    return int(sum(int(np.prod(s)) for s in tensor_shapes(T, n_class)))


@dataclass
class ModelWeights:
    """One model (model1 or model2): 60 f32 tensors in Keras positional order."""
    tensors: List[np.ndarray]
    T: int
    n_class: int
    source: str = ""

    def __post_init__(self):
        want = tensor_shapes(self.T, self.n_class)
        if len(self.tensors) != N_TENSORS:
            raise ValueError(f"expected {N_TENSORS} tensors, got {len(self.tensors)}")
        for i, (t, s) in enumerate(zip(self.tensors, want)):
            if tuple(t.shape) != tuple(s):
                raise ValueError(f"tensor {i} ({ROLES[i]}): shape {t.shape} != {s}")

    def __getitem__(self, role) -> np.ndarray:
        if isinstance(role, str):
            return self.tensors[ROLES.index(role)]
        return self.tensors[role]

    def flat(self) -> np.ndarray:
        """The flat f32 blob handed to the C-ABI (`nrv_weights.data`)."""
        return np.ascontiguousarray(
            np.concatenate([t.astype("<f4", copy=False).ravel() for t in self.tensors]))

    def with_window(self, T: int, seed: int = 20260) -> "ModelWeights":
        """Same model at another window length.

        Only `feature.kernel` (6T,16) depends on T.  The shipped files
        are T=11; BASELINE.json's "13-event window" configs need a (78,16) kernel that
        does not exist, so a seeded synthetic N(0, 0.25^2) one is substituted
        (SURVEY.md 8d, C4).  Results at T != shipped T are throughput-only.
        """
        if T == self.T:
            return self
        rng = np.random.default_rng(seed + T)
        ts = list(self.tensors)
        ts[56] = (rng.standard_normal((6 * T, 16)) * 0.25).astype(np.float32)
        return ModelWeights(ts, T, self.n_class, self.source + f"+synthetic_feature_T{T}")


def _split_flat(flat: np.ndarray, T: int, n_class: int) -> List[np.ndarray]:
    out, off = [], 0
    for s in tensor_shapes(T, n_class):
        n = int(np.prod(s))
        out.append(flat[off:off + n].reshape(s).copy())
        off += n
    if off != flat.size:
        raise ValueError("flat blob size mismatch")
    return out


def infer_T_and_classes(n_f32: int):
    """Recover (T, n_class) from a blob length: only tensors 56 and 58/59 vary."""
    for n_class in (N_CLASS_M1, N_CLASS_M2):
        base = n_params(0, n_class)
        rest = n_f32 - base
        if rest > 0 and rest % (6 * 16) == 0:
            T = rest // 96
            if n_params(T, n_class) == n_f32 and 1 <= T <= 64:
                return T, n_class
    raise ValueError(f"blob of {n_f32} f32 is not a NanoReviser model")


def load_f32(path: str) -> ModelWeights:
    flat = np.fromfile(path, dtype="<f4")
    T, n_class = infer_T_and_classes(flat.size)
    man = os.path.splitext(path)[0] + ".json"
    if os.path.exists(man):
        with open(man) as f:
            m = json.load(f)
        if m["n_f32"] != flat.size:
            raise ValueError(f"{path}: manifest/blob size mismatch")
    return ModelWeights(_split_flat(flat, T, n_class), T, n_class, os.path.basename(path))


def load_h5(path: str) -> ModelWeights:
    from . import h5lite
    ts = h5lite.read_keras_weights(path)
    n_class = int(ts[59].shape[0])
    T = int(ts[56].shape[0]) // 6
    return ModelWeights([np.asarray(t, dtype=np.float32) for t in ts], T, n_class,
                        os.path.basename(path))


def load_model(path: str) -> ModelWeights:
    """Load `<stem>.h5` or `<stem>.f32`; if `path` is missing try the other extension."""
    stem, ext = os.path.splitext(path)
    cands = [path] + [stem + e for e in (".f32", ".h5") if stem + e != path]
    for p in cands:
        if os.path.exists(p):
            return load_h5(p) if p.endswith(".h5") else load_f32(p)
    raise FileNotFoundError(f"model file: {path} (also tried .f32/.h5)")


def default_model_dir() -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "model")


def species_paths(species: str, model_dir: str | None = None) -> Sequence[str]:
    """NanoReviser.py:191-193 path convention."""
    d = model_dir or default_model_dir()
    return (os.path.join(d, species, f"{species}_win13_50ep_model1.h5"),
            os.path.join(d, species, f"{species}_win13_50ep_model2.h5"))


def load_species(species: str, model_dir: str | None = None):
    p1, p2 = species_paths(species, model_dir)
    return load_model(p1), load_model(p2)
