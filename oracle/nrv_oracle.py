"""ORACLE - TEST INFRASTRUCTURE ONLY.  NumPy restatement of the reviser graph.

Nothing under oracle/ is part of the product.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it, and only as the checker / the
reported CPU baseline.  The product path (nanoreviser_amd + libnanorev_hip.so) never
routes through this file and has no CPU fallback.

PARITY STATUS: **parity unpinned** for the model graph.  The arithmetic of the
reference's hot path lives in third-party keras 2.2.4 / tensorflow 1.12
(enviroment/NanoReviser_cpu.yaml:57-60; .h5 root attr keras_version='2.2.4'),
neither of which is vendored under /root/reference nor installable here, and the
reference ships no golden outputs for `predict`.  This file restates
the published Keras-2.2.4 layer semantics at the reference's own call sites:

  graph      nanorevutils/output_handeler.py:206-255 (model1), :258-307 (model2)
             == nanorevutils/lstmmodel.py:32-133
  CNN block  nanorevutils/nanorevcnn.py:17-38
  inputs     nanorevutils/nanorevtrainutils.py:162-169,198-209

and is linked to the authors' trained behaviour only empirically (tests/
test_oracle_properties.py: the shipped E. coli weights reproduce the Albacore base
at the window centre >= 97 %, and every semantic ablation collapses that).

Semantics implemented:
  * TimeDistributed = reshape (B,T,..)->(B*T,..).
  * Conv1D k=3 'same': out[p,o] = b[o] + sum_{k,c} x[p+k-1,c] W[k,c,o], zero pad; ReLU
    inside the conv layer, BatchNorm after it (nanorevcnn.py:24-25).
  * BatchNorm inference: x*inv + (beta - mean*inv), inv = gamma/sqrt(var+1e-3).
  * Add([x, inpt]) broadcasts (..,50,1) over the 8 channels (nanorevcnn.py:37).
  * Flatten is channels-last row-major (index p*8+o; head: t*6+k).
  * LSTM (Keras 2.2.4): z = x W + h U + b, gate order i,f,c,o; i,f,o = hard_sigmoid
    = clip(0.2 z + 0.5, 0, 1); g = tanh; c' = f c + i g; h' = o tanh(c'); h0=c0=0.
  * Bidirectional concat [fw, bw]; the backward output is re-reversed to input order.
  * concatenate([read_rnn2, signal_x_out]) (output_handeler.py:222).
  * softmax max-subtracted; argmax ties -> lowest index.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-3  # keras.layers.BatchNormalization default epsilon


def hard_sigmoid(z):
    return np.clip(z * z.dtype.type(0.2) + z.dtype.type(0.5), 0, 1)


def sigmoid(z):
    return 1 / (1 + np.exp(-z))


def _bn(x, g, b, m, v):
    dt = x.dtype.type
    inv = g / np.sqrt(v + dt(BN_EPS))
    return x * inv + (b - m * inv)


def _conv1d_same_relu(x, W, b):
    """x (E,50,Cin), W (3,Cin,Cout): nanorevcnn.py:24 Conv1D(k=3,'same',relu)."""
    E, P, Cin = x.shape
    xp = np.zeros((E, P + 2, Cin), x.dtype)
    xp[:, 1:-1] = x
    y = np.zeros((E, P, W.shape[2]), x.dtype) + b
    for k in range(3):
        y = y + xp[:, k:k + P] @ W[k]
    return np.maximum(y, 0)


def signal_branch(w, sig):
    """sig (E,50) -> (E,64).  output_handeler.py:209-215, nanorevcnn.py:29-38."""
    x = sig[:, :, None]
    y = _bn(_conv1d_same_relu(x, w[0], w[1]), w[2], w[3], w[4], w[5])
    y = _bn(_conv1d_same_relu(y, w[6], w[7]), w[8], w[9], w[10], w[11])
    y = y + x                      # Add(): broadcast over the 8 channels
    flat = y.reshape(y.shape[0], 400)   # index p*8+o
    return flat @ w[32] + w[33]    # Dense(64), no activation


def lstm_dir(x, W, U, b, reverse, act):
    """x (B,T,D) -> (B,T,H) in input time order.  Keras LSTM, gate order i,f,c,o."""
    B, T, _ = x.shape
    H = U.shape[0]
    h = np.zeros((B, H), x.dtype)
    c = np.zeros((B, H), x.dtype)
    out = np.zeros((B, T, H), x.dtype)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        z = x[:, t] @ W + h @ U + b
        i = act(z[:, :H])
        f = act(z[:, H:2 * H])
        g = np.tanh(z[:, 2 * H:3 * H])
        o = act(z[:, 3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        out[:, t] = h
    return out


def bilstm(x, w6, act):
    fw = lstm_dir(x, w6[0], w6[1], w6[2], False, act)
    bw = lstm_dir(x, w6[3], w6[4], w6[5], True, act)
    return np.concatenate([fw, bw], axis=-1)


def forward(weights, signal, read, dtype=np.float32, recurrent_act="hard_sigmoid",
            return_logits=False, sig_out=None):
    """One model.  signal (B,T,50[,1]), read (B,T,6) -> softmax (B,C).

    `weights`: the 60 positional tensors.  dtype float32 = the reference's floatx;
    float64 = the high-precision arbiter used to judge fp32 implementations.
    `sig_out` (B,T,64) may be passed to skip the signal branch (per-event dedup tests).
    """
    dt = np.dtype(dtype)
    w = [np.asarray(t, dtype=dt) for t in weights]
    act = hard_sigmoid if recurrent_act == "hard_sigmoid" else sigmoid
    read = np.asarray(read, dtype=np.float32).astype(dt)   # cast-at-feed is f32 in Keras
    B, T, _ = read.shape
    if sig_out is None:
        signal = np.asarray(signal, dtype=np.float32).astype(dt).reshape(B, T, 50)
        s = signal_branch(w, signal.reshape(B * T, 50)).reshape(B, T, 64)
    else:
        s = np.asarray(sig_out, dtype=dt)
    r = _bn(bilstm(read, w[12:18], act), *w[18:22])
    r = _bn(bilstm(r, w[22:28], act), *w[28:32])
    x = np.concatenate([r, s], axis=-1)                    # [read 128 | signal 64]
    x = _bn(bilstm(x, w[34:40], act), *w[40:44])
    x = bilstm(x, w[44:50], act)
    x = np.maximum(x @ w[50] + w[51], 0)
    x = np.maximum(x @ w[52] + w[53], 0)
    x = np.maximum(x @ w[54] + w[55], 0)                   # main_out, ReLU
    flat = x.reshape(B, T * 6)                             # index t*6+k
    feat = np.maximum(flat @ w[56] + w[57], 0)
    logits = feat @ w[58] + w[59]
    if return_logits:
        return logits
    e = np.exp(logits - logits.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def predict_pair(w1, w2, signal, read, dtype=np.float32, recurrent_act="hard_sigmoid",
                 chunk=2048):
    """Both models on the same windows -> (p1 (B,6), p2 (B,5), a1, a2)."""
    B = np.asarray(read).shape[0]
    p1s, p2s = [], []
    for s in range(0, B, chunk):
        sl = slice(s, min(B, s + chunk))
        p1s.append(forward(w1, signal[sl], read[sl], dtype, recurrent_act))
        p2s.append(forward(w2, signal[sl], read[sl], dtype, recurrent_act))
    if not p1s:
        return (np.zeros((0, 6), dtype), np.zeros((0, 5), dtype),
                np.zeros(0, np.int8), np.zeros(0, np.int8))
    p1, p2 = np.concatenate(p1s), np.concatenate(p2s)
    return p1, p2, p1.argmax(-1).astype(np.int8), p2.argmax(-1).astype(np.int8)


def windows_from_events(sig_ev, feat_ev, T):
    """Sliding windows x[i:i+T], i in [0, N-T)  (nanorevtrainutils.py:198-209)."""
    N = len(feat_ev)
    n = max(N - T, 0)
    idx = np.arange(n)[:, None] + np.arange(T)[None, :]
    return np.asarray(sig_ev)[idx], np.asarray(feat_ev)[idx]


def synth_windows(n, T, seed=20260):
    """Synthetic independent windows matched to the fixture statistics.

    Returns signal (n,T,50) f32, read (n,T,6) f32.
    """
    rng = np.random.default_rng(seed)
    sig = np.clip(rng.normal(-0.10, 1.36, (n, T, 50)), -8.4, 4.8)
    color = rng.choice(np.array([30., 100., 180., 250.]), (n, T)) / 300.0
    smean = rng.normal(0.992, 0.100, (n, T))
    sstd = np.abs(rng.normal(0.0, 0.65, (n, T)))
    ln = np.minimum(2 + rng.geometric(0.15, (n, T)), 465) / 10.0
    abm = rng.normal(110.6, 20.8, (n, T))
    abs_ = rng.lognormal(np.log(4.4), 0.8, (n, T))
    read = np.stack([color, smean, sstd, ln, abm, abs_], axis=-1)
    return sig.astype(np.float32), read.astype(np.float32)
