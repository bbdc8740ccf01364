import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
m1, m2 = load_species("ecoli")
T = 13
rv = Reviser(m1.with_window(T), m2.with_window(T), batch=4096)

g = torch.Generator(device="cuda").manual_seed(1234)
def run(s, f):
    k = s.shape[0]
    p1 = torch.empty(k, 6, device="cuda"); p2 = torch.empty(k, 5, device="cuda")
    a1 = torch.empty(k, dtype=torch.int8, device="cuda"); a2 = torch.empty(k, dtype=torch.int8, device="cuda")
    rv.predict_device(s.data_ptr(), f.data_ptr(), k, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
    rv.sync()
    return p1, p2
for n in (4096, 8192, 4096 * 8, 4096 * 64, 1 << 20):
    sig = (torch.randn(n, T, 50, device="cuda", generator=g) * 1.36 - 0.10).clamp_(-8.4, 4.8)
    feat = torch.rand(n, T, 6, device="cuda", generator=g)
    o = run(sig, feat)
    perm = torch.randperm(n, device="cuda", generator=g)
    op = run(sig[perm].contiguous(), feat[perm].contiguous())
    bad = (o[0][perm] != op[0]).any(-1)
    print(n, "bad rows", int(bad.sum()), "first bad (permuted idx)", bad.nonzero()[:8].flatten().tolist(), flush=True)
    if bad.any():
        idx = bad.nonzero().flatten()
        print("   bad idx mod 4096:", sorted(set((idx % 4096).tolist()))[:20], " groups:", sorted(set((idx // 4096).tolist()))[:20])
        src = perm[idx]
        print("   source idx mod 4096:", sorted(set((src % 4096).tolist()))[:20], " groups:", sorted(set((src // 4096).tolist()))[:20])
