"""Parity of the default mode against the committed fp64 goldens on one fixture read (quick A/B of env knobs)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_read
from nanoreviser_amd import hoststage as hs
from nanoreviser_amd.engine import Reviser
from nanoreviser_amd.weights import load_species
mg = np.load("tests/golden/model_goldens.npz")
for sp in ("ecoli", "human"):
    m1, m2 = load_species(sp)
    rv = Reviser(m1, m2, precision=os.environ.get("QP_PREC", "f16x2"))
    key = "ch10_read5252"
    _, _, rt = load_read(key)
    sw, fw = hs.sliding_windows(rt.sig_ev, rt.feat_ev, 11)
    idx = mg[f"{key}/idx"]
    p1, p2, a1, a2 = rv.predict_pair(np.ascontiguousarray(sw[idx]), np.ascontiguousarray(fw[idx]))
    print(sp, "max|dp| m1 %.2e m2 %.2e flips %d" % (np.abs(p1 - mg[f"{key}/{sp}/p1"]).max(), np.abs(p2 - mg[f"{key}/{sp}/p2"]).max(),
          int((a1 != mg[f"{key}/{sp}/a1"]).sum() + (a2 != mg[f"{key}/{sp}/a2"]).sum())), flush=True)
    rv.close()
