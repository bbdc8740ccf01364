"""A third, independent restatement of the graph - on torch's own conv1d / linear primitives in
fp64, with a hand-written Keras-2.2.4 LSTM cell - against the NumPy fp64 oracle.

There is no executable Keras here (oracle header: "parity unpinned"), so what can be done is to make
sure the oracle is not one author's single reading of the graph: this file shares no code with
oracle/ (different conv implementation and padding mechanism, different flatten path, torch's BLAS)
and follows the reference directly: nanorevcnn.py:17-38, output_handeler.py:206-255 / 258-307,
Keras LSTM semantics of SURVEY.md Appendix A."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from nanoreviser_amd import hoststage as hs
from oracle import nrv_oracle as O


def _bn(x, g, b, m, v):                      # channels last, Keras BatchNormalization(eps=1e-3) at inference
    return (x - m) / torch.sqrt(v + 1e-3) * g + b


def _lstm(x, W, U, b, reverse):
    """Keras 2.2.4 LSTM: gates i,f,c,o along 4H; recurrent_activation hard_sigmoid; zero state."""
    B, T, _ = x.shape
    H = U.shape[0]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = [None] * T
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        z = x[:, t] @ W + h @ U + b
        zi, zf, zc, zo = z.split(H, dim=1)
        hsig = lambda a: torch.clamp(0.2 * a + 0.5, 0.0, 1.0)        # noqa: E731
        c = hsig(zf) * c + hsig(zi) * torch.tanh(zc)
        h = hsig(zo) * torch.tanh(c)
        out[t] = h                                                    # backward outputs land in time order
    return torch.stack(out, dim=1)


def _bilstm(x, w):
    return torch.cat([_lstm(x, w[0], w[1], w[2], False), _lstm(x, w[3], w[4], w[5], True)], dim=-1)


def torch_forward(weights, signal, read):
    w = [torch.from_numpy(np.asarray(t, np.float64)) for t in weights]
    sig = torch.from_numpy(np.asarray(signal, np.float64))            # (B,T,50)
    x = torch.from_numpy(np.asarray(read, np.float64))                # (B,T,6)
    B, T, _ = x.shape
    s = sig.reshape(B * T, 1, 50)                                     # (N, C=1, L=50)
    # Conv1d_BN: conv('same') -> relu -> BN, twice; kernels are Keras (k, in, out)
    y = F.conv1d(s, w[0].permute(2, 1, 0), w[1], padding=1)
    y = _bn(F.relu(y).permute(0, 2, 1), w[2], w[3], w[4], w[5])       # (N,50,8)
    y = F.conv1d(y.permute(0, 2, 1), w[6].permute(2, 1, 0), w[7], padding=1)
    y = _bn(F.relu(y).permute(0, 2, 1), w[8], w[9], w[10], w[11])
    y = y + s.permute(0, 2, 1)                                        # residual, broadcast over 8 channels
    sx = (y.reshape(B * T, 400) @ w[32] + w[33]).reshape(B, T, 64)    # flatten index p*8+o, linear
    r = _bn(_bilstm(x, w[12:18]), *w[18:22])
    r = _bn(_bilstm(r, w[22:28]), *w[28:32])
    t = torch.cat([r, sx], dim=-1)                                    # [read 128 | signal 64]
    t = _bn(_bilstm(t, w[34:40]), *w[40:44])
    t = _bilstm(t, w[44:50])
    t = F.relu(t @ w[50] + w[51])
    t = F.relu(t @ w[52] + w[53])
    t = F.relu(t @ w[54] + w[55])                                     # (B,T,6)
    f = F.relu(t.reshape(B, T * 6) @ w[56] + w[57])
    return torch.softmax(f @ w[58] + w[59], dim=-1).numpy()


@pytest.mark.parametrize("sp", ["ecoli", "human"])
def test_torch_fp64_restatement_agrees_with_numpy_oracle(reads, species_models, sp):
    _, _, rt = reads("ch141_read5436")
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
    sw, fw = np.ascontiguousarray(sw[100:148]), np.ascontiguousarray(fw[100:148])
    for m in species_models[sp]:
        want = O.forward(m.tensors, sw, fw, np.float64)
        got = torch_forward(m.tensors, sw, fw)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-9
        assert np.array_equal(got.argmax(-1), want.argmax(-1))


def test_torch_restatement_other_window_length(species_models):
    m1, _ = species_models["ecoli"]
    m = m1.with_window(5)
    sig, rd = O.synth_windows(16, 5)
    assert np.abs(torch_forward(m.tensors, sig, rd) - O.forward(m.tensors, sig, rd, np.float64)).max() < 1e-9
