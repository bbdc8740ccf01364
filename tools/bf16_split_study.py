#!/usr/bin/env python3
"""Offline study for SURVEY.md 8f-4: would a split-bf16 MFMA path hold the parity bars?

Emulates, in NumPy, the LSTM/dense contractions with each f32 operand split into n bf16 terms
(x = hi + mid + lo, round-to-nearest-even) and the products hi*hi, hi*mid, ... accumulated in
f32 (what v_mfma_f32_32x32x16_bf16 does), everything else in f32, and compares the softmax outputs
with the fp64 oracle on real fixture windows.  Variants: 'x3' = 3 products (hi*hi, hi*mid, mid*hi),
'x6' = 6 products (all with i+j <= 2).  Nothing here is used by the product.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_read                      # noqa: E402
from nanoreviser_amd import hoststage as hs         # noqa: E402
from nanoreviser_amd.weights import load_species    # noqa: E402
from oracle import nrv_oracle as O                  # noqa: E402


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split(x, n):
    parts, rem = [], np.asarray(x, np.float32)
    for _ in range(n):
        p = bf16_round(rem)
        parts.append(p)
        rem = (rem - p).astype(np.float32)
    return parts


def split_f16(x, n, ftz=False):
    """n f16 terms (11-bit significands); ftz=True zeroes f16 subnormals (what a flushing pipe would do)."""
    parts, rem = [], np.asarray(x, np.float32)
    for _ in range(n):
        p = rem.astype(np.float16).astype(np.float32)
        if ftz:
            p = np.where(np.abs(p) < 2.0 ** -14, np.float32(0), p)
        parts.append(p)
        rem = (rem - p).astype(np.float32)
    return parts


def make_matmul(mode):
    if mode == "f32":
        return lambda a, b: a @ b
    if mode in ("h3", "h4", "h3z", "h3s", "h4s", "h3sz"):
        pairs = [(0, 0), (0, 1), (1, 0)] + ([(1, 1)] if mode.startswith("h4") else [])
        scaled = "s" in mode[2:]

        def pow2_scale(x):
            """power of two that brings max|x| into [2^13, 2^14) (exact; f16 max is 65504)"""
            m = float(np.abs(x).max())
            return np.float32(2.0 ** (13 - np.floor(np.log2(m)))) if m > 0 else np.float32(1)

        def mmh(a, b):
            if scaled:
                sa, sb = pow2_scale(a), pow2_scale(b)
                a, b = a * sa, b * sb
            A, B = split_f16(a, 2, mode.endswith("z")), split_f16(b, 2, mode.endswith("z"))
            out = np.zeros((a.shape[0], b.shape[1]), np.float32)
            for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):
                out = out + (A[i] @ B[j])
            if scaled:
                out = out * (np.float32(1) / (sa * sb))
            return out
        return mmh
    nterm = 2 if mode == "x3" else 3
    pairs = [(i, j) for i in range(nterm) for j in range(nterm) if i + j <= nterm - 1]

    def mm(a, b):
        A, B = split(a, nterm), split(b, nterm)
        out = np.zeros((a.shape[0], b.shape[1]), np.float32)
        for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):     # small terms first
            out = out + (A[i].astype(np.float32) @ B[j].astype(np.float32))
        return out
    return mm


def forward_with(weights, signal, read, mm):
    """oracle.forward with every LSTM / dense contraction routed through `mm` (f32 elsewhere)."""
    w = [np.asarray(t, np.float32) for t in weights]
    act = O.hard_sigmoid
    read = np.asarray(read, np.float32)
    B, T, _ = read.shape
    s = O.signal_branch(w, np.asarray(signal, np.float32).reshape(B * T, 50)).reshape(B, T, 64)

    def lstm_dir(x, W, U, b, reverse):
        H = U.shape[0]
        h = np.zeros((B, H), np.float32)
        c = np.zeros((B, H), np.float32)
        out = np.zeros((B, T, H), np.float32)
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            z = mm(x[:, t], W) + mm(h, U) + b
            i, f, g, o = act(z[:, :H]), act(z[:, H:2 * H]), np.tanh(z[:, 2 * H:3 * H]), act(z[:, 3 * H:])
            c = f * c + i * g
            h = o * np.tanh(c)
            out[:, t] = h
        return out

    def bil(x, w6):
        return np.concatenate([lstm_dir(x, w6[0], w6[1], w6[2], False), lstm_dir(x, w6[3], w6[4], w6[5], True)], -1)

    r = O._bn(bil(read, w[12:18]), *w[18:22])
    r = O._bn(bil(r, w[22:28]), *w[28:32])
    x = np.concatenate([r, s], -1)
    x = O._bn(bil(x, w[34:40]), *w[40:44])
    x = bil(x, w[44:50])
    x2 = x.reshape(B * T, -1)
    x2 = np.maximum(mm(x2, w[50]) + w[51], 0)
    x2 = np.maximum(mm(x2, w[52]) + w[53], 0)
    x2 = np.maximum(x2 @ w[54] + w[55], 0)
    flat = x2.reshape(B, T * 6)
    feat = np.maximum(flat @ w[56] + w[57], 0)
    logits = feat @ w[58] + w[59]
    e = np.exp(logits - logits.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def main():
    mg = np.load(os.path.join(ROOT, "tests", "golden", "model_goldens.npz"))
    for sp in ("ecoli", "human"):
        m1, m2 = load_species(sp)
        for key in ("ch10_read5252", "ch10_read6297"):
            _, _, rt = load_read(key)
            sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
            idx = mg[f"{key}/idx"][:256]
            sw, fw = np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx])
            for mode in (sys.argv[1:] or ("f32", "x3", "x6", "h3", "h4", "h3z")):
                mm = make_matmul(mode)
                res = []
                for m, nm in ((m1, "p1"), (m2, "p2")):
                    p = forward_with(m.tensors, sw, fw, mm)
                    g = mg[f"{key}/{sp}/{nm}"][:256]
                    res.append((float(np.abs(p - g).max()), int((p.argmax(-1) != g.argmax(-1)).sum())))
                print(f"{sp:6s} {key:15s} {mode:4s}  max|dp| m1 {res[0][0]:.2e} (flips {res[0][1]})   "
                      f"m2 {res[1][0]:.2e} (flips {res[1][1]})", flush=True)


if __name__ == "__main__":
    main()
