"""No result may depend on what the engine's workspace held before (MI355X only, -m gpu).  Device memory is filled with NaN
bit patterns / FLT_MAX, handed back to the driver, and the engine - whose hipMalloc'd workspace then lies in that memory - runs
ragged launch groups (coalesced, on one stream, on stream lanes) in every arithmetic mode: the outputs are the bits of a run on
a clean device.  Written in round 6 after one unexplained failure of test_small_groups_coalesced_or_on_lanes_bit_identical
[bf16x3] in ~60 runs (scripts/gpu_r6_poison.py is the long form: three poisons x five groupings x three modes, green)."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3", "f32"])
def test_results_do_not_depend_on_stale_device_memory(species_models, mode, monkeypatch):
    import torch
    from nanoreviser_amd.engine import Reviser
    m1, m2 = species_models["ecoli"]
    T, n, N = 11, 6_037, 13_011
    g = torch.Generator(device="cuda").manual_seed(78)
    sig = (torch.randn(n, T, 50, device="cuda", generator=g) * 1.36 - 0.10).clamp_(-8.4, 4.8)
    feat = torch.rand(n, T, 6, device="cuda", generator=g)
    sig_ev = (torch.randn(N, 50, device="cuda", generator=g) * 1.36 - 0.10).clamp_(-8.4, 4.8)
    feat_ev = torch.rand(N, 6, device="cuda", generator=g)

    def outs(k):
        return (torch.full((k, 6), float("nan"), device="cuda"), torch.full((k, 5), float("nan"), device="cuda"),
                torch.full((k,), -7, dtype=torch.int8, device="cuda"), torch.full((k,), -7, dtype=torch.int8, device="cuda"))

    def run(rv):
        w, r = outs(n), outs(N - T)
        torch.cuda.synchronize()
        rv.predict_device(sig.data_ptr(), feat.data_ptr(), n, *[x.data_ptr() for x in w])
        rv.predict_read_device(sig_ev.data_ptr(), feat_ev.data_ptr(), N, *[x.data_ptr() for x in r])
        rv.sync()
        torch.cuda.synchronize()
        return [x.cpu() for x in w + r]

    def poison(bits):
        blocks = [torch.empty(1 << 28, dtype=torch.int32, device="cuda").fill_(bits) for _ in range(8)]     # 8 GiB
        torch.cuda.synchronize()
        del blocks
        torch.cuda.empty_cache()
        torch.cuda.synchronize()

    monkeypatch.setenv("NRV_PRECISION", mode)
    monkeypatch.delenv("NRV_COALESCE", raising=False)
    rv = Reviser(m1, m2, batch=4096)
    ref = run(rv)
    rv.close()
    for bits, coalesce, batch in ((0x7FC00000, "1", 1000), (0x7F7FFFFF, "0", 1000), (0x7FC00000, "0", 992)):
        poison(bits)
        monkeypatch.setenv("NRV_COALESCE", coalesce)
        rv = Reviser(m1, m2, batch=batch)
        for _ in range(2):
            for i, (x, y) in enumerate(zip(ref, run(rv))):
                assert torch.equal(x, y), (mode, hex(bits), coalesce, batch, i, int((x != y).sum()))
        rv.close()
