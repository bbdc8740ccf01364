"""Randomised differential run: random window lengths / batch sizes / launch-group sizes / modes, HIP
engine vs the C oracle (f32) on seeded synthetic windows.  Dev tool; the fixed cases live in tests/."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from oracle import nrv_oracle as O
from oracle import c_oracle as CO
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
m1, m2 = load_species("ecoli")
worst, cases, t0 = 0.0, 0, time.time()
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 90:
    T = int(rng.integers(1, 33)); n = int(rng.integers(1, 2500)); batch = int(rng.choice([32, 64, 96, 128, 500, 512, 1024, 4096]))
    prec = str(rng.choice(["f16x2", "f16x2", "bf16x3", "f32"])); act = str(rng.choice(["hard_sigmoid", "sigmoid"]))
    a, b = m1.with_window(T), m2.with_window(T)
    sig, rd = O.synth_windows(n, T, seed=int(rng.integers(1 << 30)))
    rv = Reviser(a, b, precision=prec, recurrent_activation=act, batch=batch)
    p1, p2, a1, a2 = rv.predict_pair(sig, rd)
    c1, ca1 = CO.predict(a.flat(), T, 6, sig, rd, threads=8, recurrent_act=act) if "recurrent_act" in CO.predict.__code__.co_varnames else (None, None)
    if c1 is None:
        q1, q2, b1, b2 = O.predict_pair(a.tensors, b.tensors, sig, rd, np.float32, recurrent_act=act)
    else:
        c2, ca2 = CO.predict(b.flat(), T, 5, sig, rd, threads=8, recurrent_act=act)
        q1, q2, b1, b2 = c1, c2, ca1, ca2
    d = max(float(np.abs(p1 - q1).max()), float(np.abs(p2 - q2).max()))
    bad = int((a1 != b1).sum() + (a2 != b2).sum())
    # argmax may differ only on near ties (both are f32 paths)
    if bad:
        for arr, brr, p in ((a1, b1, q1), (a2, b2, q2)):
            for i in np.nonzero(arr != brr)[0]:
                gap = p[i, brr[i]] - p[i, arr[i]]
                assert gap <= 2e-4, (T, n, batch, prec, act, i, gap)
    if d > 1e-4:
        # two f32-grade evaluations may differ by the SUM of their deviations on an ill-conditioned window:
        # fp64 arbitrates (the engine must be within the bar, or no worse than 3x the f32 oracle's own miss)
        r1, r2, _, _ = O.predict_pair(a.tensors, b.tensors, sig, rd, np.float64, recurrent_act=act)
        dh = max(float(np.abs(p1 - r1).max()), float(np.abs(p2 - r2).max()))
        do = max(float(np.abs(q1 - r1).max()), float(np.abs(q2 - r2).max()))
        assert dh <= max(1e-4, 3 * do), (T, n, batch, prec, act, d, dh, do)
    worst = max(worst, d); cases += 1
    rv.close()
print(f"fuzz: {cases} cases, worst max|dp| vs f32 oracle {worst:.2e}, all argmax equal or near-tie")
