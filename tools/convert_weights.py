#!/opt/conda/bin/python3.9
"""Convert the reference's Keras `save_weights` HDF5 files into flat f32 blobs.

Runs in the build container only (needs h5py, which this image's default
python3 lacks; /opt/conda/bin/python3.9 has it).  Output, per weight file:

  model/<species>/<species>_win13_50ep_model{1,2}.f32    60 tensors, C-order f32,
                                                          concatenated POSITIONALLY
  model/<species>/<species>_win13_50ep_model{1,2}.json   manifest: per tensor role,
                                                          original name, shape, offset,
                                                          sha256; plus file-level attrs

The tensor order is the order Keras' `load_weights` uses: root attr
`layer_names`, then each layer group's `weight_names` (layer *names* differ
between the four files, so only position is meaningful).
Reference writer: NanoReviser_train.py:175-176,203-204 (`save_weights`).
"""
import hashlib
import json
import os
import sys

import h5py
import numpy as np

# role of each positional tensor
ROLES = (
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
assert len(ROLES) == 60


def _s(x):
    return x.decode() if isinstance(x, bytes) else str(x)


def convert(h5_path, out_stem):
    f = h5py.File(h5_path, "r")
    tensors = []
    blob = bytearray()
    idx = 0
    for lname in [_s(n) for n in f.attrs["layer_names"]]:
        g = f[lname]
        for wname in [_s(n) for n in g.attrs["weight_names"]]:
            a = np.ascontiguousarray(g[wname][()], dtype="<f4")
            tensors.append({
                "index": idx, "role": ROLES[idx], "h5_name": wname,
                "shape": list(a.shape), "offset_f32": len(blob) // 4,
                "sha256": hashlib.sha256(a.tobytes()).hexdigest(),
            })
            blob += a.tobytes()
            idx += 1
    assert idx == 60, idx
    man = {
        "source": os.path.basename(h5_path),
        "keras_version": _s(f.attrs["keras_version"]),
        "backend": _s(f.attrs["backend"]),
        "n_tensors": idx, "n_f32": len(blob) // 4,
        "blob_sha256": hashlib.sha256(bytes(blob)).hexdigest(),
        "tensors": tensors,
    }
    with open(out_stem + ".f32", "wb") as o:
        o.write(bytes(blob))
    with open(out_stem + ".json", "w") as o:
        json.dump(man, o, indent=1)
    return man


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sp in ("ecoli", "human"):
        for m in (1, 2):
            stem = f"{sp}_win13_50ep_model{m}"
            man = convert(os.path.join(ref, "model", sp, stem + ".h5"),
                          os.path.join(root, "model", sp, stem))
            T = man["tensors"][56]["shape"][0] // 6
            print(stem, man["n_f32"], "f32; window T =", T, "classes =", man["tensors"][59]["shape"][0])
