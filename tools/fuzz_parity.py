"""Randomised parity sweep: random model shapes, precisions, layouts, clustering settings
against the oracle.  `run(seed, n)` returns (worst error per category, list of failures);
tests/test_fuzz_gpu.py runs fixed-seed slices of it in the suite, the command line
(`python tools/fuzz_parity.py SEED N`) runs longer sweeps and exits non-zero on a failure.

A failure is
  * |score - oracle| > 1e-4 on a state whose likelihood the reference's float storage can hold
    (ll > -103.97: aku/phone_probs.cc:224-262 stores (float)exp(ll), which is 0 below that, so
    the LNA output is the floor whatever the value -- tests/test_lna_gpu.py pins that),
  * below that, a value that would NOT flush in the reference's float storage (the observable
    contract where the digits of ll are not: see note()), or
  * a per-frame count of exactly evaluated clusters that differs from the oracle's."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VISIBLE_LL = -103.97
TOL = float(os.environ.get("AASR_FUZZ_TOL", "1e-4"))   # the contract; a lower value lists the closest calls
LOG_TINY = float(np.log(1e-50))


def run(seed=1, N=40, verbose=False, big=False):
    from aaltoasr_amd import capi
    from oracle import oracle as O
    O.build()
    L = capi.lib()
    L.aasr_debug_kappa.restype = C.c_double
    L.aasr_debug_kappa.argtypes = [C.c_void_p]
    rng = np.random.default_rng(seed)
    worst = {}
    fails = []

    def note(key, got, want, ctx, floor_slack=0.0):
        d = np.abs(got - want)
        err = float(d.max())
        worst[key] = max(worst.get(key, 0.0), err)
        vis = want > VISIBLE_LL
        evis = float(d[vis].max()) if vis.any() else 0.0
        worst[key + " (ll > -104)"] = max(worst.get(key + " (ll > -104)", 0.0), evis)
        # below the flush point the reference's float storage of the likelihood holds 0.0 and the LNA entry is
        # the floor whatever the digits of ll: the observable contract there is that the engine's value flushes
        # as well (as a float likelihood: 0, or the one denormal quantum a 1e-4 move across the edge produces).
        # (Parts merged at the 1e-50 floor -- outlier / class routing, |det| entering through the reference
        # exponent -- only ever move such a value further down.)
        with np.errstate(under="ignore"):
            floor_bad = (~vis) & (np.exp(got.astype(np.float64)).astype(np.float32) > np.float32(2.0 ** -149))
        if evis > TOL or floor_bad.any():
            at = int(np.argmax(np.where(floor_bad, d, -1.0))) if floor_bad.any() else int(d.argmax())
            fails.append("%s %s err %.3g (visible %.3g) at ll %.2f (got %.2f)" % (key, ctx, err, evis, want.ravel()[at],
                                                                                   got.ravel()[at]))
            if verbose:
                print("FAIL", fails[-1])

    for it in range(N):
        D = int(rng.choice([1, 2, 5, 13, 24, 39, 40, 47, 63]))
        S = int(rng.integers(1, 70))
        lo = int(rng.integers(0, 3))
        hi = int(rng.integers(max(1, lo), 30))
        if big:   # production-sized state inventories (many tiles, several row cuts per launch)
            D = int(rng.choice([13, 24, 39, 39, 39, 47]))
            S = int(rng.integers(300, 3600))
            hi = int(rng.integers(max(1, lo), 48))
        tied = bool(rng.integers(0, 2))
        n = rng.integers(lo, hi + 1, S)
        if n.sum() == 0:
            n[0] = 1
        K = int(n.sum())
        G = int(K if not tied else max(4, K // 2))
        mean = rng.standard_normal((G, D)) * rng.uniform(0.3, 2.0)
        var = np.exp(rng.uniform(np.log(0.2), np.log(5.0), (G, D)))
        if rng.integers(0, 4) == 0 and G >= 8:
            # a few ill-conditioned Gaussians (kappa >> 600): outlier routing
            tight = rng.choice(G, max(1, G // 16), replace=False)
            var[tight] *= 10.0 ** rng.uniform(-3.5, -2.0)
        off = np.zeros(S + 1, np.int32)
        off[1:] = np.cumsum(n)
        idx = (rng.integers(0, G, K) if tied else np.arange(K)).astype(np.int32)
        w = rng.uniform(0.01, 1.0, K)
        if rng.integers(0, 3) == 0 and K > 2:
            w[rng.integers(0, K)] = 0.0                       # a zero weight
        F = int(rng.integers(1, 400))
        if big:
            F = int(rng.integers(1, 4)) * 256 + int(rng.integers(0, 256))
        frames = (rng.standard_normal((F, D)) * rng.uniform(0.5, 2.5)).astype(np.float32)
        om = O.DiagModel(mean, var, off, idx, w)
        want = om.score(frames.astype(np.float64))
        g = capi.Gmm.from_arrays(mean, var, off, idx, w)
        ctx = "seed %d it %d D %d S %d G %d tied %d F %d kappa %.0f" % (
            seed, it, D, S, G, tied, F, L.aasr_debug_kappa(g._h))
        # per-state precision routing: how many of the sweep's models get a mixed layout, how many states the probe moves
        n16, moved = g.precision_states()
        worst["models routed (mixed layout)"] = worst.get("models routed (mixed layout)", 0) + (1 if 0 < n16 < S else 0)
        worst["states moved by the f16x2 probe"] = worst.get("states moved by the f16x2 probe", 0) + moved
        for layouts in (7, 2, 0, 4):
            g.set_layouts(layouts)
            for prec in (0, 3, 4):
                try:
                    g.set_precision(prec)
                except capi.AasrError:
                    continue
                note("score layouts=%d prec=%d" % (layouts, prec), g.score(frames), want, ctx)
        g.set_layouts(7)
        g.set_precision(0)
        # AASR_PREC_F64: the reference's arithmetic in double (device exp / log are the only difference)
        f64 = g.score_f64(frames.astype(np.float64))
        e64 = float(np.abs(f64 - want).max())
        worst["f64"] = max(worst.get("f64", 0.0), e64)
        if e64 > 1e-9 * max(1.0, float(np.abs(want).max())):
            fails.append("f64 %s err %.3g" % (ctx, e64))
        Cn = int(rng.integers(1, max(2, int(0.3 * G)))) if G >= 4 else 0
        if Cn >= 1 and Cn <= 0.3 * G:
            g2c = rng.integers(0, Cn, G)
            g2c[rng.integers(0, G, max(1, G // 10))] = -1     # some Gaussians in no cluster
            pairs = [(int(i), int(c)) for i, c in enumerate(g2c) if c >= 0]
            minc, ming = float(rng.choice([0.0, 0.1, 0.5])), float(rng.choice([0.0, 0.1, 0.3, 0.7]))
            om.set_clustering(Cn, pairs, minc, ming)
            wantc, cnt = om.score_clustered(frames.astype(np.float64), want_counts=True)
            try:
                g.set_clustering(Cn, pairs)
                g.set_clustering_min_evals(minc, ming)
            except capi.AasrError as e:
                if verbose:
                    print("skip clustering:", e)
                continue
            f64 = g.score_f64(frames.astype(np.float64))
            e64 = float(np.abs(f64 - wantc).max())
            worst["f64 clustered"] = max(worst.get("f64 clustered", 0.0), e64)
            if e64 > 1e-9 * max(1.0, float(np.abs(wantc).max())) or not np.array_equal(g.cluster_exact_counts(F), cnt):
                fails.append("f64 clustered %s C %d minc %g ming %g err %.3g" % (ctx, Cn, minc, ming, e64))
            for prec in (0, 3, 4):
                try:
                    g.set_precision(prec)
                except capi.AasrError:          # no bf16x3 rows for this model (centred form only)
                    continue
                cctx = ctx + " C %d minc %g ming %g" % (Cn, minc, ming)
                try:
                    gotc = g.score(frames)
                except capi.AasrError as e:     # a documented limit (refused loudly), not a parity result
                    if e.code != capi.AASR_ERR_UNSUPPORTED:
                        raise
                    worst["clustered refused"] = worst.get("clustered refused", 0) + 1
                    if verbose:
                        print("skip clustering:", e)
                    break
                note("clustered prec=%d" % prec, gotc, wantc, cctx)
                if not np.array_equal(g.cluster_exact_counts(F), cnt):
                    fails.append("clustered prec=%d %s: exact-evaluation counts differ" % (prec, cctx))
                    if verbose:
                        print("FAIL", fails[-1])
        # model-side CMLLR on top (a third of the models): one global or per-class transforms, plain
        # and clustered, against the oracle's adapted restatements
        rng2 = np.random.default_rng([seed, it, 7])   # its own stream: the sweeps above keep their draws
        if rng2.integers(0, 3) == 0 and D <= 40:
            T = int(rng2.integers(1, 4))
            Wt = np.stack([np.hstack([0.2 * rng2.standard_normal(D)[:, None],
                                      np.eye(D) * rng2.uniform(0.85, 1.15, D) + 0.02 * rng2.standard_normal((D, D))])
                           for _ in range(T)])
            g2t = (np.zeros(G, np.int32) if T == 1 and rng2.integers(0, 2) else rng2.integers(-1, T, G).astype(np.int32))
            try:
                g.set_clustering(0)
                g.set_cmllr(g2t, Wt)
            except capi.AasrError as e:
                if verbose:
                    print("skip cmllr:", e)
                continue
            want_a = O.score_adapted(om, frames.astype(np.float64), g2t, Wt)
            slack = float(max(abs(np.log(abs(np.prod(np.diag(Wt[t][:, 1:]))))) for t in range(T)) + np.log(T + 1.0) + 0.1)
            for prec in (0, 3, 4):
                try:
                    g.set_precision(prec)
                except capi.AasrError:
                    continue
                note("cmllr T=%d prec=%d" % (T, prec), g.score(frames), want_a, ctx, floor_slack=slack)
            # AASR_PREC_F64 under the same transform(s): the oracle's values
            e64 = float(np.abs(g.score_f64(frames.astype(np.float64)) - want_a).max())
            worst["f64 cmllr"] = max(worst.get("f64 cmllr", 0.0), e64)
            if e64 > 1e-9 * max(1.0, float(np.abs(want_a).max())):
                fails.append("f64 cmllr T=%d %s err %.3g" % (T, ctx, e64))
                if verbose:
                    print("FAIL", fails[-1])
            if Cn >= 1 and Cn <= 0.3 * G:
                try:
                    g.set_clustering(Cn, pairs)
                    g.set_clustering_min_evals(minc, ming)
                except capi.AasrError as e:
                    if verbose:
                        print("skip cmllr clustering:", e)
                    continue
                want_ca, cnt_a = om.score_clustered_classes(frames.astype(np.float64), g2t, Wt, want_counts=True)
                e64 = float(np.abs(g.score_f64(frames.astype(np.float64)) - want_ca).max())
                worst["f64 cmllr clustered"] = max(worst.get("f64 cmllr clustered", 0.0), e64)
                if e64 > 1e-9 * max(1.0, float(np.abs(want_ca).max())) or not np.array_equal(g.cluster_exact_counts(F), cnt_a):
                    fails.append("f64 cmllr clustered T=%d %s C %d err %.3g" % (T, ctx, Cn, e64))
                    if verbose:
                        print("FAIL", fails[-1])
                for prec in (0, 3, 4):
                    try:
                        g.set_precision(prec)
                    except capi.AasrError:
                        continue
                    try:
                        got = g.score(frames)
                    except capi.AasrError as e:
                        if e.code != capi.AASR_ERR_UNSUPPORTED:
                            raise
                        worst["cmllr clustered refused"] = worst.get("cmllr clustered refused", 0) + 1
                        if verbose:
                            print("skip cmllr clustering:", e)
                        break
                    note("cmllr clustered prec=%d" % prec, got, want_ca, ctx + " T %d C %d" % (T, Cn), floor_slack=slack)
                    if not np.array_equal(g.cluster_exact_counts(F), cnt_a):
                        fails.append("cmllr clustered prec=%d %s: exact-evaluation counts differ" % (prec, ctx))
    return worst, fails


if __name__ == "__main__":
    worst, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                       int(sys.argv[2]) if len(sys.argv) > 2 else 40, verbose=True,
                       big=len(sys.argv) > 3 and sys.argv[3] == "big")
    for k in sorted(worst):
        print("%-40s worst |err| %.3g" % (k, worst[k]))
    print("failures:", len(fails))
    sys.exit(1 if fails else 0)
