"""Randomised sweep of the full-covariance path (k_gmm_full_score / _bf16x3) against oracle.FullModel:
random dimensions, pool sizes, ragged / tied / zero-weight mixtures, covariance spectra spanning two
decades, some non-SPD ("invalid") covariances, frames 0.5-2.5 sigma wide.  A failure is |dll| > 1e-4 on a
state the reference's float storage can hold (ll > -103.97), or a value below that which would not flush there.  `python tools/fuzz_fullcov.py
SEED N`; exits non-zero on a failure."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(seed=1, N=40, verbose=False, scales=False):
    from aaltoasr_amd import capi
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(seed)
    rng_scale = np.random.default_rng(seed + 7777)   # scales=True: unnormalised features, standard deviations 0.01 ... 100
    worst, fails, refused = {}, [], 0
    for it in range(N):
        D = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 24, 39, 47, 63]))
        S = int(rng.integers(1, 24))
        n = rng.integers(int(rng.integers(0, 2)), int(rng.integers(1, 12)) + 1, S)
        if n.sum() == 0:
            n[0] = 1
        K = int(n.sum())
        tied = bool(rng.integers(0, 2))
        G = int(K if not tied else max(2, K // 2))
        mean = rng.standard_normal((G, D)) * rng.uniform(0.3, 2.0)
        cov = np.empty((G, D, D))
        lo, hi = np.log(rng.uniform(0.03, 0.5)), np.log(rng.uniform(0.6, 5.0))
        for g in range(G):
            q, _ = np.linalg.qr(rng.standard_normal((D, D)))
            ev = np.exp(rng.uniform(lo, hi, D))
            cov[g] = (q * ev) @ q.T
            cov[g] = 0.5 * (cov[g] + cov[g].T)
        if rng.integers(0, 5) == 0 and D >= 2:
            bad = int(rng.integers(0, G))
            cov[bad] = -cov[bad]                              # not SPD: the reference's invalid Gaussian
        off = np.zeros(S + 1, np.int32)
        off[1:] = np.cumsum(n)
        idx = (rng.integers(0, G, K) if tied else np.arange(K)).astype(np.int32)
        w = rng.uniform(0.01, 1.0, K)
        if rng.integers(0, 3) == 0 and K > 2:
            w[rng.integers(0, K)] = 0.0
        F = int(rng.integers(1, 300))
        frames = (rng.standard_normal((F, D)) * rng.uniform(0.5, 2.5)).astype(np.float32)
        fs = 1.0
        if scales:
            # the same pool in other units: x -> fs x (feature variances 1e-4 ... 1e4).  The factor rows then hold
            # coefficients ~ 1 / fs and the frames components ~ fs -- where the two-term fp16 rows' `lo` terms fall into
            # the subnormals unless the columns are rescaled (gmm_build_fullcov)
            fs = float(np.exp(rng_scale.uniform(np.log(0.01), np.log(100.0))))
            mean = mean * fs
            cov = cov * fs * fs
            frames = (frames.astype(np.float64) * fs).astype(np.float32)
        ctx = "seed %d it %d D %d S %d G %d tied %d F %d scale %.3g" % (seed, it, D, S, G, tied, F, fs)
        want = O.FullModel(mean, cov, off, idx, w).score(frames.astype(np.float64))
        try:
            g = capi.Gmm.from_full(mean, cov, off, idx, w)
        except capi.AasrError as e:
            refused += 1
            if verbose:
                print("refused:", ctx, e)
            continue
        for prec in (0, 3, 4):
            try:
                g.set_precision(prec)
            except capi.AasrError:
                continue
            got = g.score(frames)
            d = np.abs(got - want)
            vis = want > -103.97
            key = "full prec=%d" % prec
            evis = float(d[vis].max()) if vis.any() else 0.0
            eall = float(d.max())
            worst[key] = max(worst.get(key, 0.0), evis)
            worst[key + " (all)"] = max(worst.get(key + " (all)", 0.0), eall)
            if verbose and evis > float(os.environ.get("AASR_FUZZ_TOL", "1.0")):
                print("NOTE %s %s visible %.3g effective precision %d" % (key, ctx, evis, g.effective_precision()))
            with np.errstate(under="ignore"):
                flushes = (np.exp(got[~vis].astype(np.float64)).astype(np.float32) <= np.float32(2.0 ** -149)).all()
            if evis > 1e-4 or not flushes:
                at = int(d.argmax())
                fails.append("%s %s err %.3g (visible %.3g) at ll %.2f (got %.2f)" % (
                    key, ctx, eall, evis, want.ravel()[at], got.ravel()[at]))
                if verbose:
                    print("FAIL", fails[-1])
        # Gaussian clustering over the pool (the cluster branch does not look at the Gaussians' type): random
        # clusters incl. empty ones and Gaussians in none, random minimum counts; scores and exact-evaluation counts
        if rng.integers(0, 2) == 0 and G >= 2:
            C = int(rng.integers(1, max(1, min(G // 3, 40)) + 1))   # read_clustering refuses C > G / 2
            g2c = rng.integers(-1, C, G)
            pairs = [(int(i), int(c)) for i, c in enumerate(g2c) if c >= 0]
            minc, ming = float(rng.choice([0.0, 0.1, 0.3, 1.0])), float(rng.choice([0.0, 0.25, 0.5]))
            om = O.FullModel(mean, cov, off, idx, w)
            try:
                om.set_clustering(C, pairs, minc, ming)
                g.set_clustering(C, pairs)
                g.set_clustering_min_evals(minc, ming)
            except (capi.AasrError, ValueError) as e:
                if verbose:
                    print("clustering refused:", ctx, e)
                continue
            want_c, want_n = om.score_clustered(frames.astype(np.float64), want_counts=True)
            for prec in (0, 3, 4):
                try:
                    g.set_precision(prec)
                except capi.AasrError:
                    continue
                got = g.score(frames)
                got_n = g.cluster_exact_counts(F)
                d = np.abs(got - want_c)
                vis = want_c > -103.97
                key = "full clustered prec=%d" % prec
                evis = float(d[vis].max()) if vis.any() else 0.0
                worst[key] = max(worst.get(key, 0.0), evis)
                with np.errstate(under="ignore"):
                    flushes = (np.exp(got[~vis].astype(np.float64)).astype(np.float32) <= np.float32(2.0 ** -149)).all()
                if evis > 1e-4 or not flushes or not np.array_equal(got_n, want_n):
                    fails.append("%s %s C %d minc %.2f ming %.2f err %.3g counts equal %s" % (
                        key, ctx, C, minc, ming, evis, np.array_equal(got_n, want_n)))
                    if verbose:
                        print("FAIL", fails[-1])
    worst["refused"] = refused
    return worst, fails


if __name__ == "__main__":
    worst, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                       int(sys.argv[2]) if len(sys.argv) > 2 else 40, verbose=True,
                       scales=os.environ.get("AASR_FUZZ_SCALES") == "1")
    for k, v in worst.items():
        print("%-28s %s" % (k, ("%.3g" % v) if isinstance(v, float) else v))
    print("failures: %d" % len(fails))
    sys.exit(1 if fails else 0)
