"""Dev script: GPU correctness + per-kernel timing for both precisions (not part of the test-suite)."""
import os, sys, time
import torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd import hoststage as hs
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from oracle import nrv_oracle as O

def fixture(key="ch10_read5252"):
    g = np.load(f"tests/golden/reads/{key}.npz")
    rd = hs.collapse_events(g["ev_start"], g["ev_mean"], g["ev_stdv"], g["ev_model_state"], g["ev_move"], g["raw_signal"])
    return rd, hs.read_tensors(rd)

def cmp(tag, got, ref):
    p1, p2, a1, a2 = got; q1, q2, b1, b2 = ref
    print(f"{tag}: n={len(a1)} max|dp1|={np.abs(p1-q1).max():.3e} max|dp2|={np.abs(p2-q2).max():.3e} "
          f"argmax mism {int((a1!=b1).sum())}/{int((a2!=b2).sum())}", flush=True)

PRECS = sys.argv[1:] or ["f32", "bf16x3"]
for sp in ("ecoli", "human"):
    m1, m2 = load_species(sp)
    T = m1.T
    rd, rt = fixture()
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, T)
    n = 700
    sw, fw = np.ascontiguousarray(sw[:n]), np.ascontiguousarray(fw[:n])
    ref64 = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float64)
    for act in ("hard_sigmoid", "sigmoid"):
        if act == "sigmoid":
            ref64 = O.predict_pair(m1.tensors, m2.tensors, sw, fw, np.float64, recurrent_act="sigmoid")
        for prec in PRECS:
            rv = Reviser(m1, m2, precision=prec, recurrent_activation=act)
            got = rv.predict_pair(sw, fw)
            cmp(f"{sp} {act} {prec} windows vs f64", got, ref64)
            got_r = rv.predict_read(rt.sig_ev[:n + T], rt.feat_ev[:n + T])
            cmp(f"{sp} {act} {prec} read-mode vs window-mode", got_r, got)
            rv.close()

# timing at T=13 synthetic, batch 4096
m1, m2 = load_species("ecoli")
T = 13
a, b = m1.with_window(T), m2.with_window(T)
sig, rd_ = O.synth_windows(4096, T)
ds = torch.from_numpy(sig).cuda(); dr = torch.from_numpy(rd_).cuda()
p1 = torch.empty(4096, 6, device="cuda"); p2 = torch.empty(4096, 5, device="cuda")
a1 = torch.empty(4096, dtype=torch.int8, device="cuda"); a2 = torch.empty(4096, dtype=torch.int8, device="cuda")
for prec in PRECS + PRECS:
    rv2 = Reviser(a, b, precision=prec)
    for it in range(3):
        rv2.predict_device(ds.data_ptr(), dr.data_ptr(), 4096, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
    rv2.sync()
    rv2.prof_enable(True)
    t0 = time.time()
    K = 20
    for it in range(K):
        rv2.predict_device(ds.data_ptr(), dr.data_ptr(), 4096, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
    rv2.sync()
    dt = (time.time() - t0) / K
    prof = rv2.prof_read()
    print(f"T={T} {prec}: {dt*1e3:.3f} ms/batch -> {4096/dt/1e6:.3f} M windows/s", flush=True)
    for k, (ms, c) in prof.items():
        print(f"    {k:40s} {ms/max(c,1)*1e3:9.1f} us")
    rv2.close()
