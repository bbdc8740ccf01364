"""A NON-BUILDER implementation in the evidence chain: torch.nn.LSTM(bidirectional=True) - PyTorch's own
recurrent machinery, written by neither the reference's authors nor this repo - against the oracle's four
Bi-LSTM layers (output_handeler.py:218-225) with the SHIPPED weights re-laid from Keras to torch layout:

    Keras kernel (D, 4H) / recurrent (H, 4H), gate order i,f,c,o, one bias (4H)
    torch weight_ih (4H, D) / weight_hh (4H, H), gate order i,f,g,o, bias_ih + bias_hh

Same gate order, transposed matrices, bias in bias_ih (bias_hh = 0); `_reverse` = Keras' backward layer, whose
outputs torch - like Keras - returns in input time order, concatenated [fw, bw].  torch.nn.LSTM has sigmoid
gates, so the comparison is against the oracle's recurrent_act="sigmoid" variant (Keras >= 2.3 behaviour);
hard_sigmoid vs sigmoid is a one-line difference inside the same cell (oracle/nrv_oracle.py lstm_dir), and the
engine's sigmoid variant is checked against that same oracle on the GPU (test_gpu_parity.py
test_sigmoid_variant_matches_oracle).  What this pins: gate order and slicing, the direction handling and
re-reversal, the concat order, zero initial state, and the weight indices 12-17 / 22-27 / 34-39 / 44-49.
CPU only, fp64, < 1e-6 (measured ~1e-15)."""
import numpy as np
import pytest
import torch

from oracle import nrv_oracle as O

LAYERS = [(12, 6, 16), (22, 32, 64), (34, 192, 128), (44, 256, 64)]     # (first weight index, D, H)


def torch_bilstm(w6, D, H):
    m = torch.nn.LSTM(input_size=D, hidden_size=H, num_layers=1, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for sfx, (W, U, b) in (("", w6[0:3]), ("_reverse", w6[3:6])):
            assert W.shape == (D, 4 * H) and U.shape == (H, 4 * H) and b.shape == (4 * H,)
            getattr(m, "weight_ih_l0" + sfx).copy_(torch.from_numpy(np.ascontiguousarray(W.T, dtype=np.float64)))
            getattr(m, "weight_hh_l0" + sfx).copy_(torch.from_numpy(np.ascontiguousarray(U.T, dtype=np.float64)))
            getattr(m, "bias_ih_l0" + sfx).copy_(torch.from_numpy(b.astype(np.float64)))
            getattr(m, "bias_hh_l0" + sfx).zero_()
    return m


@pytest.mark.parametrize("sp", ["ecoli", "human"])
@pytest.mark.parametrize("which", [0, 1])
def test_four_bilstm_layers_match_torch_nn_lstm(species_models, sp, which):
    w = [np.asarray(t, np.float64) for t in species_models[sp][which].tensors]
    rng = np.random.default_rng(100 + which)
    for base, D, H in LAYERS:
        x = rng.normal(0, 1.0, (48, 11, D))
        want = O.bilstm(x, w[base:base + 6], O.sigmoid)                 # the oracle's layer, sigmoid gates
        with torch.no_grad():
            got, _ = torch_bilstm(w[base:base + 6], D, H)(torch.from_numpy(x))
        d = float(np.abs(got.numpy() - want).max())
        assert got.shape == (48, 11, 2 * H) and d < 1e-6, (sp, which, base, d)
        # and torch agrees that the direction handling matters: feeding the backward half un-reversed differs
        fw_only = O.lstm_dir(x, *w[base:base + 3], False, O.sigmoid)
        assert np.abs(got.numpy()[..., :H] - fw_only).max() < 1e-6
        wrong_bw = O.lstm_dir(x, *w[base + 3:base + 6], False, O.sigmoid)
        assert np.abs(got.numpy()[..., H:] - wrong_bw).max() > 1e-3


def test_stacked_read_branch_through_torch(species_models):
    """The two read-branch layers chained with their BatchNorms (output_handeler.py:218-221) on real event
    features: torch's stack vs the oracle's, so the BatchNorm-between-layers wiring is covered too."""
    from conftest import load_read
    from nanoreviser_amd import hoststage as hs
    w = [np.asarray(t, np.float64) for t in species_models["ecoli"][0].tensors]
    _, _, rt = load_read("ch10_read5252")
    _, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
    x = np.ascontiguousarray(fw[2000:2064]).astype(np.float64)
    want = O._bn(O.bilstm(O._bn(O.bilstm(x, w[12:18], O.sigmoid), *w[18:22]), w[22:28], O.sigmoid), *w[28:32])

    def bn(t, g, b, m, v):
        g, b, m, v = (torch.from_numpy(a) for a in (g, b, m, v))
        return torch.nn.functional.batch_norm(t.transpose(1, 2), m, v, g, b, training=False, eps=1e-3).transpose(1, 2)

    with torch.no_grad():
        h, _ = torch_bilstm(w[12:18], 6, 16)(torch.from_numpy(x))
        h, _ = torch_bilstm(w[22:28], 32, 64)(bn(h, *w[18:22]))
        got = bn(h, *w[28:32]).numpy()
    assert np.abs(got - want).max() < 1e-6
