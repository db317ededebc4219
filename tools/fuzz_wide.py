"""Randomised parity sweep of models of feature dimension > 63 (the model as dimension parts): plain scoring in
every precision, AASR_PREC_F64 up to 192 dimensions, Gaussian clustering, model-side CMLLR (one transform / regression
classes) and clustering under CMLLR, against the oracle.  Same failure criteria as tools/fuzz_parity.py.
`python tools/fuzz_wide.py SEED N`; exits non-zero on a failure."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VISIBLE_LL = -103.97
TOL = float(os.environ.get("AASR_FUZZ_TOL", "1e-4"))


def run(seed=1, N=20, verbose=False):
    from aaltoasr_amd import capi
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(seed)
    worst, fails = {}, []

    def note(key, got, want, ctx):
        d = np.abs(got - want)
        vis = want > VISIBLE_LL
        evis = float(d[vis].max()) if vis.any() else 0.0
        worst[key] = max(worst.get(key, 0.0), evis)
        with np.errstate(under="ignore"):
            floor_bad = (~vis) & (np.exp(got.astype(np.float64)).astype(np.float32) > np.float32(2.0 ** -149))
        if evis > TOL or floor_bad.any():
            fails.append("%s %s visible err %.3g, %d values that should flush" % (key, ctx, evis, int(floor_bad.sum())))
            if verbose:
                print("FAIL", fails[-1])

    for it in range(N):
        D = int(rng.choice([64, 65, 72, 80, 96, 127, 128, 130, 160, 200]))
        S = int(rng.integers(1, 40))
        n = rng.integers(int(rng.integers(0, 2)), int(rng.integers(2, 20)) + 1, S)
        if n.sum() == 0:
            n[0] = 1
        K = int(n.sum())
        tied = bool(rng.integers(0, 2))
        G = int(K if not tied else max(4, K // 2))
        # wide vectors put every state far down: keep the Gaussians broad and the frames near them so that values are visible
        mean = rng.standard_normal((G, D)) * rng.uniform(0.2, 0.8)
        var = np.exp(rng.uniform(np.log(0.5), np.log(3.0), (G, D)))
        off = np.zeros(S + 1, np.int32)
        off[1:] = np.cumsum(n)
        idx = (rng.integers(0, G, K) if tied else np.arange(K)).astype(np.int32)
        w = rng.uniform(0.01, 1.0, K)
        if rng.integers(0, 3) == 0 and K > 2:
            w[rng.integers(0, K)] = 0.0
        F = int(rng.integers(1, 200))
        frames = (mean[rng.integers(0, G, F)] * rng.uniform(0.0, 1.0) + rng.standard_normal((F, D)) * rng.uniform(0.3, 1.0)).astype(np.float32)
        if F > 4:
            frames[:2] *= 6.0      # a few far out: the floor
        om = O.DiagModel(mean, var, off, idx, w)
        want = om.score(frames.astype(np.float64))
        g = capi.Gmm.from_arrays(mean, var, off, idx, w)
        ctx = "seed %d it %d D %d S %d G %d tied %d F %d" % (seed, it, D, S, G, tied, F)
        for prec in (0, 3, 4):
            g.set_precision(prec)
            note("score prec=%d" % prec, g.score(frames), want, ctx)
        if D <= 192:
            e64 = float(np.abs(g.score_f64(frames.astype(np.float64)) - want).max())
            worst["f64"] = max(worst.get("f64", 0.0), e64)
            if e64 > 1e-9 * max(1.0, float(np.abs(want).max())):
                fails.append("f64 %s err %.3g" % (ctx, e64))
        g.set_precision(4)
        Cn = int(rng.integers(1, max(2, int(0.3 * G)))) if G >= 4 else 0
        clustered = Cn >= 1 and Cn <= 0.3 * G
        if clustered:
            g2c = rng.integers(0, Cn, G)
            g2c[rng.integers(0, G, max(1, G // 10))] = -1
            pairs = [(int(i), int(c)) for i, c in enumerate(g2c) if c >= 0]
            minc, ming = float(rng.choice([0.0, 0.1, 0.5])), float(rng.choice([0.0, 0.1, 0.3, 0.7]))
            om.set_clustering(Cn, pairs, minc, ming)
            wantc, cnt = om.score_clustered(frames.astype(np.float64), want_counts=True)
            g.set_clustering(Cn, pairs)
            g.set_clustering_min_evals(minc, ming)
            for prec in (0, 4):
                g.set_precision(prec)
                note("clustered prec=%d" % prec, g.score(frames), wantc, ctx + " C %d minc %g ming %g" % (Cn, minc, ming))
                if not np.array_equal(g.cluster_exact_counts(F), cnt):
                    fails.append("clustered prec=%d %s: exact-evaluation counts differ" % (prec, ctx))
            g.set_clustering(0)
        if rng.integers(0, 2) == 0:
            T = int(rng.integers(1, 4))
            Wt = np.stack([np.hstack([0.1 * rng.standard_normal(D)[:, None],
                                      np.eye(D) * rng.uniform(0.9, 1.1, D) + 0.01 * rng.standard_normal((D, D))])
                           for _ in range(T)])
            g2t = (np.zeros(G, np.int32) if T == 1 and rng.integers(0, 2) else rng.integers(-1, T, G).astype(np.int32))
            g.set_cmllr(g2t, Wt)
            want_a = O.score_adapted(om, frames.astype(np.float64), g2t, Wt)
            for prec in (0, 4):
                g.set_precision(prec)
                note("cmllr T=%d prec=%d" % (T, prec), g.score(frames), want_a, ctx)
            if clustered:
                g.set_clustering(Cn, pairs)
                g.set_clustering_min_evals(minc, ming)
                want_ca, cnt_a = om.score_clustered_classes(frames.astype(np.float64), g2t, Wt, want_counts=True)
                note("cmllr clustered T=%d" % T, g.score(frames), want_ca, ctx + " C %d" % Cn)
                if not np.array_equal(g.cluster_exact_counts(F), cnt_a):
                    fails.append("cmllr clustered T=%d %s: exact-evaluation counts differ" % (T, ctx))
        g.close()
    return worst, fails


if __name__ == "__main__":
    worst, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 20, verbose=True)
    for k in sorted(worst):
        print("%-28s %.3g" % (k, worst[k]))
    print("failures: %d" % len(fails))
    sys.exit(1 if fails else 0)
