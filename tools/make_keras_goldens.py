#!/usr/bin/env python3
"""Pin the model-graph oracle to the reference's own arithmetic: Keras goldens.

The reference's hot path is `get_model1()/get_model2()` (nanorevutils/output_handeler.py:206-255,
258-307; CNN block nanorevutils/nanorevcnn.py:17-38) executed by keras 2.2.4 on tensorflow 1.12
(enviroment/NanoReviser_cpu.yaml:57-60).  Neither package exists in the build image, so
tests/golden/model_goldens.npz holds the ORACLE's outputs and parity is "unpinned".  This script is
the missing link, ready to run the moment such an environment exists:

    conda env create -f /root/reference/enviroment/NanoReviser_cpu.yaml     # keras 2.2.4 + tf 1.12
    python tools/make_keras_goldens.py [--reference /root/reference]

It builds the two predict models, loads the shipped weights POSITIONALLY (`load_weights`), runs
`model.predict([signal (n,T,50,1), read (n,T,6)])` on exactly the windows of the committed oracle
goldens (same fixture reads, same indices, same synthetic T=11 windows) for both species, and writes
tests/golden/keras_goldens.npz with the same key layout as model_goldens.npz (+ `meta/*`).
tests/test_keras_goldens.py consumes the file when it is present: the oracle (CPU) and the HIP path
(GPU) are then checked against Keras itself.

Where the graph comes from, in order of preference:
  1. the reference's OWN `get_model1/get_model2/Conv1d_BN/identity_Block` function definitions,
     AST-extracted from --reference (the module cannot be imported: it pulls in albacore at the top,
     nanorev_fast5_handeler.py:23-36) and executed with `SENT_LEN` set to the window the weight files
     were trained with (T = 11: `feature/kernel` is (66,16); the source says 13 and
     `load_weights` would fail on it);
  2. when --reference is absent, this file's restatement of the same layer list (`build_model`).
Only data is written; no reference source is copied into the repo.
"""
import argparse
import ast
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(G, "keras_goldens.npz")


def have_keras():
    return importlib.util.find_spec("keras") is not None and (
        importlib.util.find_spec("tensorflow") is not None or importlib.util.find_spec("theano") is not None)


def build_model(T, n_class):
    """Layer list of output_handeler.py:206-255 (n_class 6) / :258-307 (n_class 5), predict model only."""
    from keras.layers import (Add, BatchNormalization, Bidirectional, Conv1D, Dense, Dropout, Flatten, Input, LSTM,
                              TimeDistributed, concatenate)
    from keras.models import Model

    def conv_bn(x):                                        # nanorevcnn.py:17-26
        x = TimeDistributed(Conv1D(8, 3, padding="same", strides=1, activation="relu"))(x)
        return TimeDistributed(BatchNormalization())(x)

    signal_input = Input(shape=(T, 50, 1), dtype="float", name="signal_input")
    x = conv_bn(conv_bn(signal_input))                     # nanorevcnn.py:29-38
    x = Add()([x, signal_input])
    x = Dropout(0.2)(x)
    s = TimeDistributed(Flatten())(x)
    s = TimeDistributed(Dense(64))(s)
    read_input = Input(shape=(T, 6), dtype="float", name="read_input")
    r = Bidirectional(LSTM(16, return_sequences=True, activation="tanh"))(read_input)
    r = BatchNormalization()(r)
    r = Bidirectional(LSTM(64, return_sequences=True, activation="tanh"))(r)
    r = BatchNormalization()(r)
    t = concatenate([r, s], axis=-1)
    t = Bidirectional(LSTM(128, return_sequences=True, activation="tanh"))(t)
    t = BatchNormalization()(t)
    t = Bidirectional(LSTM(64, return_sequences=True, activation="tanh"))(t)
    o = Dense(128, activation="relu")(t)
    o = Dense(32, activation="relu")(o)
    o = Dense(6, activation="relu", name="main_out")(o)
    f = Dense(16, activation="relu", name="feature")(Flatten()(o))
    p = Dense(n_class, activation="softmax", name="final_out")(f)
    return Model(inputs=[signal_input, read_input], outputs=[p])


def reference_builders(ref_root, T):
    """get_model1 / get_model2 as the reference defines them, with SENT_LEN = T."""
    ns = {}
    exec("from keras.models import Sequential, Model\n"
         "from keras.layers import Input, Embedding, Activation, BatchNormalization, Dropout\n"
         "from keras.layers import Dense, Lambda, Flatten, RepeatVector, Permute, Multiply, Dot, concatenate, Add\n"
         "from keras.layers import Bidirectional, LSTM\n"
         "from keras.layers import TimeDistributed, Conv1D, MaxPool1D\n"
         "import keras.backend as K\n", ns)
    for rel, names in (("nanorevutils/nanorevcnn.py", ("Conv1d_BN", "identity_Block")),
                       ("nanorevutils/output_handeler.py", ("get_model1", "get_model2"))):
        tree = ast.parse(open(os.path.join(ref_root, rel)).read())
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                exec(compile(ast.Module([node], []), rel, "exec"), ns)
    ns.update(NB_CLASS=6, SENT_LEN=T, SIGNEL_LEN=50, VEC_LEN=6)      # output_handeler.py:200-203, T from the weights
    return ns["get_model1"], ns["get_model2"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference", help="checkout of pkubioinformatics/nanoreviser")
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    if not have_keras():
        print("keras (+ tensorflow) is not importable in this interpreter: nothing written.\n"
              "Create the reference's environment (enviroment/NanoReviser_cpu.yaml: keras 2.2.4, tensorflow 1.12) "
              "and run this script again.", file=sys.stderr)
        return 3
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "-1")                   # the reference runs on CPU (NanoReviser.py:37-38)
    import keras
    sys.path.insert(0, ROOT)
    from nanoreviser_amd import hoststage as hs
    from nanoreviser_amd.weights import species_paths

    mg = np.load(os.path.join(G, "model_goldens.npz"))
    index = json.load(open(os.path.join(G, "reads", "index.json")))
    T = 11
    use_ref = os.path.exists(os.path.join(args.reference, "nanorevutils", "output_handeler.py"))
    out = {"meta/keras_version": np.array(keras.__version__), "meta/backend": np.array(keras.backend.backend()),
           "meta/graph_source": np.array("reference functions (AST-extracted)" if use_ref else "tools/make_keras_goldens.build_model"),
           "meta/T": np.array(T)}
    try:
        import tensorflow as tf
        out["meta/tensorflow_version"] = np.array(tf.__version__)
    except Exception:
        pass
    if keras.__version__.split(".")[:2] != ["2", "2"]:
        print(f"warning: keras {keras.__version__} is not the pinned 2.2.4: the LSTM recurrent_activation default "
              "became 'sigmoid' in 2.3 (SURVEY.md F4) - these goldens would pin a DIFFERENT function", file=sys.stderr)
    windows = {}
    for ent in index:
        key = ent["key"]
        g = np.load(os.path.join(G, "reads", key + ".npz"))
        rd = hs.collapse_events(g["ev_start"], g["ev_mean"], g["ev_stdv"], g["ev_model_state"], g["ev_move"],
                                g["raw_signal"])
        rt = hs.read_tensors(rd)
        sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, T)
        idx = mg[f"{key}/idx"]
        windows[key] = (np.ascontiguousarray(sw[idx])[..., None], np.ascontiguousarray(fw[idx]))
        out[f"{key}/idx"] = idx
    windows["synth11"] = (mg["synth11/signal"][..., None], mg["synth11/read"])
    for sp in ("ecoli", "human"):
        paths = [p if p.endswith(".h5") else os.path.splitext(p)[0] + ".h5" for p in species_paths(sp)]
        if use_ref:
            b1, b2 = reference_builders(args.reference, T)
            models = (b1(), b2())
        else:
            models = (build_model(T, 6), build_model(T, 5))
        for m, p in zip(models, paths):
            m.load_weights(p)                                             # positional (no by_name), as Keras does
        for key, (sig, rd_) in windows.items():
            for i, m in enumerate(models, 1):
                p = m.predict([sig, rd_], batch_size=512)
                out[f"{key}/{sp}/p{i}"] = p.astype(np.float32)
                out[f"{key}/{sp}/a{i}"] = p.argmax(-1).astype(np.int8)
            print(key, sp, "done", flush=True)
        keras.backend.clear_session()
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
