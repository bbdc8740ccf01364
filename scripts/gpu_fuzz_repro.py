"""Replays scripts/gpu_fuzz.py's random sequence up to one failing case and looks at it: against the fp64
oracle, against the C f32 oracle, with lanes off and with groups of 4096."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
from oracle import nrv_oracle as O
seed, want = int(sys.argv[1]), tuple(int(x) for x in sys.argv[2:5])      # seed T n batch
rng = np.random.default_rng(seed)
m1, m2 = load_species("ecoli")
while True:
    T = int(rng.integers(1, 33)); n = int(rng.integers(1, 2500)); batch = int(rng.choice([32, 64, 96, 128, 500, 512, 1024, 4096]))
    prec = str(rng.choice(["f16x2", "f16x2", "bf16x3", "f32"])); act = str(rng.choice(["hard_sigmoid", "sigmoid"]))
    s = int(rng.integers(1 << 30))
    if (T, n, batch) == want:
        break
print("case", T, n, batch, prec, act, s)
a, b = m1.with_window(T), m2.with_window(T)
sig, rd = O.synth_windows(n, T, seed=s)
q1, q2, b1, b2 = O.predict_pair(a.tensors, b.tensors, sig, rd, np.float64, recurrent_act=act)
c1, c2, ca1, ca2 = O.predict_pair(a.tensors, b.tensors, sig, rd, np.float32, recurrent_act=act)      # what the fuzz compares with
print("NumPy f32 oracle vs fp64: m1 %.2e m2 %.2e" % (np.abs(c1 - q1).max(), np.abs(c2 - q2).max()))
outs = {}
for tag, env, bt, pr in (("lanes", "1", batch, prec), ("nolanes", "0", batch, prec), ("b4096", "1", 4096, prec), ("f32mode", "1", batch, "f32")):
    os.environ["NRV_LANES"] = env
    rv = Reviser(a, b, precision=pr, recurrent_activation=act, batch=bt)
    p1, p2, a1, a2 = rv.predict_pair(sig, rd)
    rv.close()
    outs[tag] = (p1, p2, a1, a2)
    i1 = int(np.abs(p1 - q1).max(axis=1).argmax()); i2 = int(np.abs(p2 - q2).max(axis=1).argmax())
    print(f"{tag}: vs fp64 m1 {np.abs(p1 - q1).max():.2e} (win {i1}) m2 {np.abs(p2 - q2).max():.2e} (win {i2}); vs NumPy f32 m1 {np.abs(p1 - c1).max():.2e} m2 {np.abs(p2 - c2).max():.2e}; argmax diffs vs fp64 {int((a1 != b1).sum() + (a2 != b2).sum())}")
for tag in ("nolanes", "b4096"):
    print(tag, "bit-identical to lanes:", all(np.array_equal(x, y) for x, y in zip(outs["lanes"], outs[tag])))
w = int(np.abs(outs["lanes"][1] - c2).max(axis=1).argmax())
print("worst window vs C (m2):", w, "HIP", outs["lanes"][1][w], "C", c2[w], "fp64", q2[w])
