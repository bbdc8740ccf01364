"""Quick look at the f16x2 mode: parity vs the committed fp64 goldens (both species), vs bf16x3 mode, and timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_read
from nanoreviser_amd import hoststage as hs
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from nanoreviser_amd import workload as W
MODES = tuple(os.environ.get("H2_MODES", "bf16x3,f16x2").split(","))
mg = np.load("tests/golden/model_goldens.npz")
for sp in ("ecoli", "human"):
    m1, m2 = load_species(sp)
    for mode in MODES:
        rv = Reviser(m1, m2, precision=mode)
        worst = [0, 0]; flips = 0
        for key in ("ch10_read5252", "ch13_read2251", "ch141_read5436"):
            _, _, rt = load_read(key)
            sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
            idx = mg[f"{key}/idx"]
            p1, p2, a1, a2 = rv.predict_pair(np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx]))
            worst[0] = max(worst[0], np.abs(p1 - mg[f"{key}/{sp}/p1"]).max()); worst[1] = max(worst[1], np.abs(p2 - mg[f"{key}/{sp}/p2"]).max())
            flips += int((a1 != mg[f"{key}/{sp}/a1"]).sum() + (a2 != mg[f"{key}/{sp}/a2"]).sum())
            r = rv.predict_read(rt.sig_ev[:1500], rt.feat_ev[:1500])
            w = rv.predict_pair(np.ascontiguousarray(sw[:1489]), np.ascontiguousarray(fw[:1489]))
            same = all(np.array_equal(x, y) for x, y in zip(r, w))
        print(sp, mode, "max|dp| vs fp64 m1 %.2e m2 %.2e flips %d read==window %s" % (worst[0], worst[1], flips, same), flush=True)
        rv.close()
import torch
m1, m2 = load_species("ecoli")
T = 13
sig, rd = W.synth_windows(4096, T)
d_sig, d_rd = torch.from_numpy(sig).cuda(), torch.from_numpy(rd).cuda()
p1 = torch.empty(4096, 6, device="cuda"); p2 = torch.empty(4096, 5, device="cuda")
a1 = torch.empty(4096, dtype=torch.int8, device="cuda"); a2 = torch.empty(4096, dtype=torch.int8, device="cuda")
for mode in MODES:
    rv = Reviser(m1.with_window(T), m2.with_window(T), precision=mode)
    rv.set_stream(torch.cuda.current_stream().cuda_stream)
    args = (d_sig.data_ptr(), d_rd.data_ptr(), 4096, p1.data_ptr(), p2.data_ptr(), a1.data_ptr(), a2.data_ptr())
    for _ in range(300): rv.predict_device(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): rv.predict_device(*args)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    rv.prof_enable(1); rv.prof_read()
    for _ in range(32): rv.predict_device(*args)
    torch.cuda.synchronize()
    pr = rv.prof_read()
    print(mode, "%.3f ms/step" % ms, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in pr.items()}, flush=True)
    rv.close()
