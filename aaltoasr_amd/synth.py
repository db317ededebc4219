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


def push_states_over_the_f16_limits(model, states, kappa2=100.0):
    """Per-state precision routing test models: in every state of `states` the first component's Gaussian is moved away
    from the pool's pivot (the mean of the means) until its conditioning estimate sqrt(sum_d (p (mu - pivot)^2)^2) is
    `kappa2` -- above the plain two-term fp16 layout's limit (80), below the slab-constant layout's (160), with the sum
    itself (~380-420 for the synthetic models' variances) between their limits too (250 / 500).  Returns a new model tuple (disjoint pools: only those states
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


def make_speechlike_audio(n_samples, seed=SEED + 5, sample_rate=16000):
    """A source-filter imitation of speech, seeded: segments of 40-300 ms that are silence (noise 40-60 dB down),
    voiced (an impulse train whose pitch glides between 80 and 260 Hz plus a little aspiration noise, through three
    formant resonators), fricative (white noise through one wide resonator at 2.5-6.5 kHz) or a short burst, each
    with its own level.  What it is for: features whose distribution has the modes trained acoustic models see --
    silence against speech in the energy and low cepstral dimensions above all -- so that a model fitted to them
    (fit_model) has the conditioning of a trained one rather than that of one stationary blob.  int16."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(seed)
    out = np.empty(n_samples, np.float64)
    level = 10.0 ** (rng.uniform(-6.0, 3.0) / 20.0)          # the recording's gain

    def resonator(f, bw):
        r = np.exp(-np.pi * bw / sample_rate)
        th = 2.0 * np.pi * f / sample_rate
        a = np.array([1.0, -2.0 * r * np.cos(th), r * r])
        return np.array([a.sum()]), a                        # unit gain at 0 Hz

    pos = 0
    zis = [np.zeros(2) for _ in range(3)]
    while pos < n_samples:
        n = min(n_samples - pos, int(rng.uniform(0.04, 0.30) * sample_rate))
        kind = rng.choice(4, p=[0.22, 0.50, 0.20, 0.08])
        if kind == 0:                                         # silence
            x = rng.standard_normal(n)
            rms = rng.uniform(3.0, 30.0)
            filt = []
        elif kind == 1:                                       # voiced
            f0 = np.linspace(rng.uniform(80.0, 260.0), rng.uniform(80.0, 260.0), n)
            phase = np.cumsum(f0 / sample_rate) + rng.uniform()
            x = np.zeros(n)
            x[np.flatnonzero(np.diff(np.floor(phase), prepend=np.floor(phase[0])) > 0)] = 1.0
            x = x * 12.0 + rng.standard_normal(n) * 0.02
            rms = rng.uniform(1200.0, 7000.0)
            filt = [resonator(rng.uniform(250, 850), rng.uniform(60, 130)),
                    resonator(rng.uniform(850, 2500), rng.uniform(80, 180)),
                    resonator(rng.uniform(2300, 3500), rng.uniform(100, 250))]
        elif kind == 2:                                       # fricative
            x = rng.standard_normal(n)
            rms = rng.uniform(150.0, 2500.0)
            filt = [resonator(rng.uniform(2500, 6500), rng.uniform(700, 1500))]
        else:                                                 # burst: noise with a fast decay
            n = min(n, int(0.06 * sample_rate))
            x = rng.standard_normal(n) * np.exp(-np.arange(n) / (0.012 * sample_rate))
            rms = rng.uniform(300.0, 2500.0)
            filt = [resonator(rng.uniform(1500, 5000), rng.uniform(500, 1200))]
        for j, (b, a) in enumerate(filt):
            x, zis[j] = lfilter(b, a, x, zi=zis[j] * 0.0)
        x *= rms / max(np.sqrt((x * x).mean()), 1e-9)         # the segment's level
        ramp = min(80, n // 2)                                # 5 ms edges: no clicks between segments
        if ramp > 0:
            e = 0.5 - 0.5 * np.cos(np.pi * np.arange(ramp) / ramp)
            x[:ramp] *= e
            x[n - ramp:] *= e[::-1]
        out[pos:pos + n] = x
        pos += n
    out = out * level + rng.standard_normal(n_samples) * 2.0   # the microphone's floor
    return np.clip(np.rint(out), -32767, 32767).astype(np.int16)


def fit_model(frames, S=3125, comps=16, seed=SEED + 6, iters=4, minvar=0.1, smooth=4.0):
    """A model FITTED to data, of the shape HmmSet::read_all (aku/HmmSet.cc:351-357) loads in production: S states of
    `comps` diagonal Gaussians each (disjoint pool, G = S * comps).  Two levels, seeded, in numpy: the frames are
    clustered into S "states" by Lloyd iterations (what the tied states of a trained model partition the feature space
    into), every state's frames into `comps` components by repeated splitting; a component's mean is its frames' mean, its variance their
    variance smoothed towards the state's with `smooth` pseudo-counts and floored at `minvar` times the global variance
    of the dimension (aku's estimate --minvar 0.1 on unit-variance features, aku/estimate.cc:131), its weight its share
    of the state's frames.  Returns the tuple make_model returns."""
    rng = np.random.default_rng(seed)
    X = np.ascontiguousarray(frames, np.float32)
    F, D = X.shape
    if F < S:
        raise ValueError("fit_model: fewer frames (%d) than states (%d)" % (F, S))
    gmean = X.mean(0, dtype=np.float64)
    gvar = X.var(0, dtype=np.float64)
    floor = minvar * gvar
    Z = ((X - gmean.astype(np.float32)) / np.sqrt(gvar).astype(np.float32)).astype(np.float32)   # cluster in whitened units

    def assign(Zs, cent, chunk=16384):
        c2 = (cent * cent).sum(1)
        a = np.empty(Zs.shape[0], np.int32)
        for i in range(0, Zs.shape[0], chunk):
            d = c2[None, :] - 2.0 * (Zs[i:i + chunk] @ cent.T)
            a[i:i + chunk] = d.argmin(1)
        return a

    def lloyd(Zs, k, n_iter):
        """k centres of Zs by seeded Lloyd iterations; an empty cluster takes a frame of the largest one."""
        n = Zs.shape[0]
        c = Zs[rng.choice(n, k, replace=False)].copy()
        for _ in range(n_iter):
            aa = assign(Zs, c)
            cnt = np.bincount(aa, minlength=k)
            o = np.argsort(aa, kind="stable")
            starts = np.searchsorted(aa[o], np.arange(k))
            nz = cnt > 0
            sums = np.zeros((k, D), np.float64)
            sums[nz] = np.add.reduceat(Zs[o].astype(np.float64), starts[nz], axis=0)
            c[nz] = (sums[nz] / cnt[nz, None]).astype(np.float32)
            for e in np.flatnonzero(~nz):
                donor = int(cnt.argmax())
                c[e] = Zs[rng.choice(np.flatnonzero(aa == donor))]
                cnt[donor] -= 1
        return c

    # first level in two steps (a tree, as the tying of a trained model's states is): sqrt(S) coarse clusters, each then
    # divided into its share of the S states
    C1 = max(1, min(int(round(np.sqrt(S))), S))
    coarse = lloyd(Z, C1, iters)
    ac = assign(Z, coarse)
    cnt_c = np.bincount(ac, minlength=C1)
    share = np.maximum(1, np.floor(cnt_c / F * S).astype(np.int64))
    share = np.minimum(share, np.maximum(cnt_c, 1))
    while share.sum() > S:
        share[np.argmax(share)] -= 1
    while share.sum() < S:
        room = cnt_c - share
        share[np.argmax(np.where(room > 0, cnt_c / share, -1))] += 1
    cent = np.empty((S, D), np.float32)
    a = np.zeros(F, np.int32)
    pos = 0
    for c in range(C1):
        rows = np.flatnonzero(ac == c)
        k = int(share[c])
        cent[pos:pos + k] = lloyd(Z[rows], k, iters) if rows.size >= k and rows.size > 0 else coarse[c] + 0.01 * rng.standard_normal((k, D)).astype(np.float32)
        if rows.size:
            a[rows] = pos + assign(Z[rows], cent[pos:pos + k])   # a frame stays inside its coarse cluster
        pos += k
    G = S * comps

    def seg_sums(values, keys, n_keys, order=None):
        """Per-key sums of the rows of `values` (in their own type) and the keys' counts."""
        o = np.argsort(keys, kind="stable") if order is None else order
        cnt = np.bincount(keys, minlength=n_keys)
        nz = cnt > 0
        starts = np.searchsorted(keys[o], np.arange(n_keys))
        out = np.zeros((n_keys, values.shape[1]), values.dtype)
        out[nz] = np.add.reduceat(values[o], starts[nz], axis=0)
        return out, cnt

    # second level, all states at once: every state's frames are split `log2(comps)` times -- a cell along the dimension
    # of its largest (whitened) variance at its mean -- the way mixture splitting grows a trained model's components
    levels = int(round(np.log2(comps)))
    if 2 ** levels != comps:
        raise ValueError("fit_model: comps must be a power of two")
    key = a.astype(np.int64)
    n_keys = S
    ar = np.arange(F)
    for _ in range(levels):
        o = np.argsort(key, kind="stable")
        s1, cnt = seg_sums(Z, key, n_keys, o)
        s2, _ = seg_sums(Z * Z, key, n_keys, o)
        nk = np.maximum(cnt, 1)[:, None]
        mu = s1 / nk
        dim_split = (s2 / nk - mu * mu).argmax(1)
        thr = mu[np.arange(n_keys), dim_split]
        bit = Z[ar, dim_split[key]] > thr[key]
        key = key * 2 + bit
        n_keys *= 2
    c2 = np.zeros((S, comps, D), np.float32)                          # where a component without frames is put: its cell's parent
    s1, cnt = seg_sums(Z, key, G)
    c2.reshape(G, D)[:] = np.repeat(cent, comps, 0)
    c2.reshape(G, D)[cnt > 0] = (s1[cnt > 0] / cnt[cnt > 0, None]).astype(np.float32)
    Xc = X.astype(np.float64) - gmean                                # centred: sums of squares without cancellation
    o = np.argsort(key, kind="stable")
    s1, cnt = seg_sums(Xc, key, G, o)
    s2, _ = seg_sums(Xc * Xc, key, G, o)
    t1, t2, cnt_s = s1.reshape(S, comps, D).sum(1), s2.reshape(S, comps, D).sum(1), cnt.reshape(S, comps).sum(1)
    ns = np.maximum(cnt_s, 1)[:, None]
    svar = np.where(cnt_s[:, None] > 1, np.maximum(t2 / ns - (t1 / ns) ** 2, floor), np.maximum(0.5 * gvar, floor))
    svar_g = np.repeat(svar, comps, 0)
    nk = np.maximum(cnt, 1)[:, None]
    mu_c = s1 / nk
    ss = np.maximum(s2 - cnt[:, None] * mu_c * mu_c, 0.0)             # sum of squared deviations from the component's mean
    var = np.maximum((ss + smooth * svar_g) / (cnt[:, None] + smooth), floor)
    mean = mu_c + gmean
    empty = cnt == 0
    if empty.any():                                                   # a component without frames: where it started, the state's variance
        mean[empty] = gmean + c2.reshape(G, D)[empty].astype(np.float64) * np.sqrt(gvar)
        var[empty] = svar_g[empty]
    w = (cnt + 0.5).reshape(S, comps)
    w = (w / w.sum(1, keepdims=True)).reshape(G)
    off = (np.arange(S + 1) * comps).astype(np.int32)
    idx = np.arange(G, dtype=np.int32)
    return mean, var, off, idx, w


def conditioning(mean, var, pivot=None):
    """(kappa, kappa2) of every Gaussian around `pivot` (default: the mean of the means, the engine's): the sum and the
    2-norm over the dimensions of p (mu - pivot)^2, the estimates gmm.h's limits are written in."""
    mean = np.asarray(mean, np.float64)
    if pivot is None:
        pivot = mean.mean(0).astype(np.float32).astype(np.float64)
    t = (mean - pivot) ** 2 / np.asarray(var, np.float64)
    return t.sum(1), np.sqrt((t * t).sum(1))
