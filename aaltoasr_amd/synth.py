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


def push_states_over_the_f16_limits(model, states, kappa2=120.0):
    """Per-state precision routing test models: in every state of `states` the first component's Gaussian is moved away
    from the pool's pivot (the mean of the means) until its conditioning estimate sqrt(sum_d (p (mu - pivot)^2)^2) is
    `kappa2` -- above the two-term fp16 form's limit (80), below the matrix path's (200), with the sum itself below its
    limits as well for the synthetic models' variances.  Returns a new model tuple (disjoint pools: only those states
    are affected)."""
    mean, var, off, idx, w = model
    mean = np.array(mean, np.float64)
    pivot = mean.mean(0).astype(np.float32).astype(np.float64)
    for s in states:
        g = int(idx[off[s]])
        mc = mean[g] - pivot
        k2 = np.sqrt((((mc * mc) / var[g]) ** 2).sum())
        mean[g] = pivot + mc * np.sqrt(kappa2 / k2)
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


def make_feature_config(seed=SEED + 4, dim_cep=12, sample_rate=16000):
    """Text of the production feature graph (the structure of the reference's
    aku/tests/mfcc_cms_norm.feaconf: audiofile -> fft (power) -> mel + power -> dct ->
    merge -> delta, delta-delta -> merge -> normalization -> lin_transform -> mean_subtractor)
    with seeded synthetic normalisation vectors and a seeded well-conditioned 39 x 39
    transform in place of trained ones."""
    rng = np.random.default_rng(seed)
    d = 3 * (dim_cep + 1)
    mean = np.concatenate([rng.normal(0.0, 4.0, dim_cep), [19.5], np.zeros(2 * (dim_cep + 1))])
    scale = np.exp(rng.uniform(np.log(0.05), np.log(0.5), d))
    a = np.eye(d) + 0.15 * rng.standard_normal((d, d))
    bias = 0.1 * rng.standard_normal(d)

    def vec(v):
        return " ".join("%.6g" % x for x in np.asarray(v).ravel())

    mods = [
        ("audiofile", "audiofile", None, [("sample_rate", sample_rate), ("copy_borders", 1), ("pre_emph_coef", 0.97)]),
        ("fft", "fft", "audiofile", [("magnitude", 0)]),
        ("mel", "mel", "fft", []),
        ("power", "power", "fft", []),
        ("mfcc", "dct", "mel", [("dim", dim_cep)]),
        ("mfcc_power", "merge", "mfcc power", []),
        ("delta1", "delta", "mfcc_power", [("width", 2)]),
        ("delta2", "delta", "delta1", [("width", 3)]),
        ("mfcc_p_d_dd", "merge", "mfcc_power delta1 delta2", []),
        ("normalization", "normalization", "mfcc_p_d_dd", [("mean", vec(mean)), ("scale", vec(scale))]),
        ("transform", "lin_transform", "normalization", [("dim", d), ("matrix", vec(a)), ("bias", vec(bias))]),
        ("cms", "mean_subtractor", "transform", [("left", 50), ("right", 25)]),
    ]
    out = []
    for name, typ, src, opts in mods:
        t = "module\n{\n  name %s\n  type %s\n" % (name, typ)
        for k, v in opts:
            t += "  %s %s\n" % (k, v)
        if src:
            t += "  sources %s\n" % src
        out.append(t + "}\n")
    return "\n".join(out)
