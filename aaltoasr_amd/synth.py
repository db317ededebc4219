"""synth.py -- seeded synthetic workloads shared by tests and bench.py.

Shapes follow BASELINE.md section 3: frames ~ N(0,1), means ~ N(0,1),
variances exp(U(ln 0.25, ln 4)), mixture weights Dirichlet(1); audio =
Gaussian noise + three sinusoids.
"""
from __future__ import annotations

import numpy as np

SEED = 20260928


def make_model(D=39, G=256, S=32, comps=8, seed=SEED, tied=False, var_lo=0.25, var_hi=4.0,
               comps_range=None):
    """Returns (mean[G,D], var[G,D], mix_off[S+1], mix_idx[K], mix_w[K]).

    comps_range=(lo,hi) draws a per-state component count uniformly; disjoint
    layouts then need sum(n_s) <= G.  tied=True draws indices from the pool.
    """
    rng = np.random.default_rng(seed)
    mean = rng.standard_normal((G, D))
    var = np.exp(rng.uniform(np.log(var_lo), np.log(var_hi), (G, D)))
    if comps_range is None:
        n = np.full(S, comps, np.int64)
    else:
        n = rng.integers(comps_range[0], comps_range[1] + 1, S)
    off = np.zeros(S + 1, np.int32)
    off[1:] = np.cumsum(n)
    K = int(off[-1])
    if tied:
        idx = rng.integers(0, G, K).astype(np.int32)
    else:
        if K > G:
            raise ValueError("disjoint layout needs sum(n_s)=%d <= G=%d" % (K, G))
        idx = np.arange(K, dtype=np.int32)
    w = np.empty(K)
    for s in range(S):
        a, b = off[s], off[s + 1]
        if b > a:
            w[a:b] = rng.dirichlet(np.ones(b - a))
    return mean, var, off, idx, w


def make_clustering(mean, n_clusters, seed=SEED + 3, iters=4):
    """Deterministic k-means on the Gaussian means (what aku's gcluster produces in
    spirit): returns gauss_to_cluster [G] int32 with every cluster non-empty."""
    rng = np.random.default_rng(seed)
    mean = np.asarray(mean, np.float64)
    G = mean.shape[0]
    cent = mean[rng.choice(G, n_clusters, replace=False)].copy()
    assign = np.zeros(G, np.int64)
    for _ in range(iters):
        d2 = (mean * mean).sum(1)[:, None] - 2.0 * mean @ cent.T + (cent * cent).sum(1)[None, :]
        assign = d2.argmin(1)
        for c in range(n_clusters):
            m = assign == c
            if m.any():
                cent[c] = mean[m].mean(0)
    for c in range(n_clusters):          # no empty cluster: steal a Gaussian
        if not (assign == c).any():
            donor = np.bincount(assign, minlength=n_clusters).argmax()
            assign[np.flatnonzero(assign == donor)[0]] = c
    return assign.astype(np.int32)


def make_frames(F, D=39, seed=SEED + 1, scale=1.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((F, D)) * scale).astype(np.float32)


def make_audio(n_samples, seed=SEED + 2, sample_rate=16000):
    """int16 = clip(round(2000*N(0,1) + 6000*sum_j sin(2 pi f_j t)))."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sample_rate
    x = 2000.0 * rng.standard_normal(n_samples)
    for fj in (220.0, 1370.0, 3100.0):
        x += 6000.0 * np.sin(2 * np.pi * fj * t)
    return np.clip(np.rint(x), -32767, 32767).astype(np.int16)
