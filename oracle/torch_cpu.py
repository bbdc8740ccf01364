"""BLAS-grade CPU evaluation of the reviser graph on PyTorch-CPU (oneDNN / MKL GEMMs), float32.

TEST INFRASTRUCTURE / REPORTED BASELINE ONLY - like everything under oracle/ this is never imported by the
product (nanoreviser_amd/), only by tests/ and by bench.py's `cpu_baseline` leg.

Why it exists: the reference runs its graph through Keras on TensorFlow-MKL with the CPU forced
(/root/reference/NanoReviser.py:37-38); neither is installable here, and oracle/nrv_oracle.c - a scalar,
unblocked loop - is far below what such a stack delivers.  This module times the SAME graph
(/root/reference/nanorevutils/nanorevcnn.py:17-38, output_handeler.py:206-307; Keras-2.2.4 semantics of
SURVEY.md Appendix A) with every contraction on the host's BLAS: one GEMM per layer for the convolutions (unfolded),
the dense layers and the input projections of the four Bi-LSTM layers (all T steps at once), and one
(B x H) x (H x 4H) GEMM per recurrent step.  It is checked against the fp64 oracle before anything is timed.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _bn(x, g, b, m, v):                      # Keras BatchNormalization(epsilon=1e-3) at inference, channels last
    return (x - m) * (g / torch.sqrt(v + 1e-3)) + b


def _lstm(x, W, U, b, reverse):
    """Keras 2.2.4 LSTM, gates i,f,c,o along 4H, hard_sigmoid recurrent activation, zero initial state.
    The input projection of all T steps is ONE GEMM; the recurrence is one GEMM per step."""
    B, T, D = x.shape
    H = U.shape[0]
    zx = (x.reshape(B * T, D) @ W + b).reshape(B, T, 4 * H)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = x.new_empty(B, T, H)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        z = torch.addmm(zx[:, t], h, U)
        zi, zf, zc, zo = z.split(H, dim=1)
        i = torch.clamp(0.2 * zi + 0.5, 0.0, 1.0)
        f = torch.clamp(0.2 * zf + 0.5, 0.0, 1.0)
        o = torch.clamp(0.2 * zo + 0.5, 0.0, 1.0)
        c = f * c + i * torch.tanh(zc)
        h = o * torch.tanh(c)
        out[:, t] = h                          # the backward direction's outputs land in time order
    return out


def _bilstm(x, w):
    return torch.cat([_lstm(x, w[0], w[1], w[2], False), _lstm(x, w[3], w[4], w[5], True)], dim=-1)


class TorchCpuModel:
    """One of the two models with its weights as torch tensors (float32 by default)."""

    def __init__(self, tensors, dtype=torch.float32):
        self.w = [torch.from_numpy(np.ascontiguousarray(np.asarray(t))).to(dtype) for t in tensors]
        self.dtype = dtype
        # conv kernels as torch wants them: (out, in, k) from Keras (k, in, out)
        self.k1 = self.w[0].permute(2, 1, 0).contiguous()
        self.k2 = self.w[6].permute(2, 1, 0).contiguous()

    @torch.no_grad()
    def forward(self, signal, read):
        w = self.w
        sig = torch.from_numpy(np.ascontiguousarray(signal)).to(self.dtype)   # (B,T,50)
        x = torch.from_numpy(np.ascontiguousarray(read)).to(self.dtype)       # (B,T,6)
        B, T, _ = x.shape
        s = sig.reshape(B * T, 1, 50)
        y = F.conv1d(s, self.k1, w[1], padding=1)
        y = _bn(F.relu(y).permute(0, 2, 1), w[2], w[3], w[4], w[5])
        y = F.conv1d(y.permute(0, 2, 1), self.k2, w[7], padding=1)
        y = _bn(F.relu(y).permute(0, 2, 1), w[8], w[9], w[10], w[11])
        y = y + s.permute(0, 2, 1)                                       # residual, broadcast over the 8 channels
        sx = (y.reshape(B * T, 400) @ w[32] + w[33]).reshape(B, T, 64)
        r = _bn(_bilstm(x, w[12:18]), *w[18:22])
        r = _bn(_bilstm(r, w[22:28]), *w[28:32])
        t = torch.cat([r, sx], dim=-1)                                   # [read 128 | signal 64]
        t = _bn(_bilstm(t, w[34:40]), *w[40:44])
        t = _bilstm(t, w[44:50])
        t = F.relu(t.reshape(B * T, -1) @ w[50] + w[51])
        t = F.relu(t @ w[52] + w[53])
        t = F.relu(t @ w[54] + w[55]).reshape(B, T * 6)
        f = F.relu(t @ w[56] + w[57])
        return torch.softmax(f @ w[58] + w[59], dim=-1).numpy()


def predict_pair(m1: "TorchCpuModel", m2: "TorchCpuModel", signal, read):
    p1, p2 = m1.forward(signal, read), m2.forward(signal, read)
    return p1, p2, p1.argmax(-1).astype(np.int8), p2.argmax(-1).astype(np.int8)
