"""Does any result depend on what the engine's workspace held before?  Device memory is filled with NaN bit patterns (and, second
pass, with large finite garbage), handed back to the driver, and the engine - whose hipMalloc'd workspace then lies in that memory -
runs ragged launch groups in every mode; outputs must be the bits of a run on a clean device.  python3 scripts/gpu_r6_poison.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species

m1, m2 = load_species("ecoli")
T, n, N = 11, 10_037, 30_011
g = torch.Generator(device="cuda").manual_seed(77)
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


def poison(kind, gib=24):
    blocks = []
    for _ in range(gib):
        t = torch.empty(1 << 28, dtype=torch.float32, device="cuda")           # 1 GiB
        if kind == "nan":
            t.fill_(float("nan"))
        else:
            t.view(torch.int32).fill_(0x7F7FFFFF if kind == "max" else 0x4B3C614E)   # FLT_MAX / 1.2e7
        blocks.append(t)
    torch.cuda.synchronize()
    del blocks
    torch.cuda.empty_cache()
    torch.cuda.synchronize()


bad = 0
for mode in ("f16x2", "bf16x3", "f32"):
    os.environ["NRV_PRECISION"] = mode
    os.environ.pop("NRV_COALESCE", None)
    rv = Reviser(m1, m2, batch=4096)
    ref = run(rv)
    rv.close()
    for kind in ("nan", "max", "big"):
        for coalesce, batch in (("1", 4096), ("1", 1000), ("0", 1000), ("0", 992), ("0", 512)):
            poison(kind)
            os.environ["NRV_COALESCE"] = coalesce
            rv = Reviser(m1, m2, batch=batch)
            got = run(rv)
            got2 = run(rv)
            rv.close()
            for i, (x, y, z) in enumerate(zip(ref, got, got2)):
                for tag, q in (("first", y), ("second", z)):
                    if not torch.equal(x, q):
                        d = (x != q) if x.dim() == 1 else (x != q).any(1)
                        idx = torch.nonzero(d).flatten()
                        print(f"MISMATCH mode {mode} poison {kind} coalesce {coalesce} batch {batch} output {i} ({tag} run): {idx.numel()} rows, first {idx[:6].tolist()}")
                        bad += 1
    print(f"{mode}: done", flush=True)
print("poison test:", "FAILED" if bad else "ok", bad)
sys.exit(1 if bad else 0)
