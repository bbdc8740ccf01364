"""Synthetic workloads of BASELINE.json's configs: the window / read generators that
`bench.py`, the GPU tests and the scripts under `scripts/` share.

The distributions are matched to the statistics of the 40 940 events of the five fixture reads
(unitest/test_data/fast5 of the reference):

  signal       ~ N(-0.10, 1.36^2) clipped to [-8.4, 4.8]           (normalised samples, preprocessing.py:85-170)
  colour/300   uniform over {30, 100, 180, 250}/300                 (preprocessing.py:173-175)
  mean/shift   ~ N(0.992, 0.100^2)        std/scale ~ |N(0, 0.65^2)|
  length/10    = (2 + Geom(0.15))/10 capped at 46.5
  ab_mean      ~ N(110.6, 20.8^2)         ab_std ~ LogNormal(ln 4.4, 0.8)

Feature column order is nanorevtrainutils.py:169.
"""
from __future__ import annotations

import numpy as np


def _features(rng, shape):
    color = rng.choice(np.array([30., 100., 180., 250.]), shape) / 300.0
    smean = rng.normal(0.992, 0.100, shape)
    sstd = np.abs(rng.normal(0.0, 0.65, shape))
    ln = np.minimum(2 + rng.geometric(0.15, shape), 465) / 10.0
    abm = rng.normal(110.6, 20.8, shape)
    abs_ = rng.lognormal(np.log(4.4), 0.8, shape)
    return np.stack([color, smean, sstd, ln, abm, abs_], axis=-1)


def synth_windows(n: int, T: int, seed: int = 20260):
    """n independent T-event windows (config C4): signal (n,T,50) f32, read (n,T,6) f32."""
    rng = np.random.default_rng(seed)
    sig = np.clip(rng.normal(-0.10, 1.36, (n, T, 50)), -8.4, 4.8)
    read = _features(rng, (n, T))
    return sig.astype(np.float32), read.astype(np.float32)


def synth_read(n_events: int, seed: int = 20265):
    """One synthetic read of n_events bases as per-event arrays (config C5): sig_ev (N,50) f32,
    feat_ev (N,6) f32.  Its N - T sliding windows (nanorevtrainutils.py:198-209) are formed on the
    device by nrv_predict_read*."""
    rng = np.random.default_rng(seed)
    sig = np.clip(rng.normal(-0.10, 1.36, (n_events, 50)), -8.4, 4.8)
    feat = _features(rng, (n_events,))
    return sig.astype(np.float32), feat.astype(np.float32)
