"""Randomised parity sweep (not part of the test suite): random model shapes,
precisions, layouts, clustering settings against the oracle.  Prints the worst
error per category; exits non-zero above 1e-4."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aaltoasr_amd import capi, synth
from oracle import oracle as O

O.build()
import ctypes as C
_L = capi.lib(); _L.aasr_debug_kappa.restype = C.c_double; _L.aasr_debug_kappa.argtypes = [C.c_void_p]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = {}
bad = 0
for it in range(N):
    D = int(rng.choice([1, 2, 5, 13, 24, 39, 40, 47, 63]))
    S = int(rng.integers(1, 70))
    lo = int(rng.integers(0, 3))
    hi = int(rng.integers(max(1, lo), 30))
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
    off = np.zeros(S + 1, np.int32); off[1:] = np.cumsum(n)
    idx = (rng.integers(0, G, K) if tied else np.arange(K)).astype(np.int32)
    w = rng.uniform(0.01, 1.0, K)
    if rng.integers(0, 3) == 0 and K > 2:
        w[rng.integers(0, K)] = 0.0                       # a zero weight
    F = int(rng.integers(1, 400))
    frames = (rng.standard_normal((F, D)) * rng.uniform(0.5, 2.5)).astype(np.float32)
    om = O.DiagModel(mean, var, off, idx, w)
    want = om.score(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    for layouts in (7, 2, 0, 4):
        g.set_layouts(layouts)
        for prec in (0, 3):
            try:
                g.set_precision(prec)
            except capi.AasrError:
                continue
            got = g.score(frames)
            err = float(np.abs(got - want).max())
            key = "score layouts=%d prec=%d" % (layouts, prec)
            worst[key] = max(worst.get(key, 0.0), err)
            vis = want > -103.97   # what (float)exp(ll) can hold: the LNA path flushes the rest
            if vis.any():
                worst[key + " (ll > -104)"] = max(worst.get(key + " (ll > -104)", 0.0),
                                                  float(np.abs(got - want)[vis].max()))
            if err > 1e-4:
                bad += 1
                print("FAIL", key, "it", it, "D", D, "S", S, "G", G, "tied", tied, "F", F, "err", err,
                      "kappa %.0f" % _L.aasr_debug_kappa(g._h), "max|x| %.1f" % np.abs(frames).max(),
                      "worst ll %.1f" % want.ravel()[np.abs(g.score(frames) - want).argmax()])
    g.set_layouts(7); g.set_precision(0)
    C = int(rng.integers(1, max(2, int(0.3 * G)))) if G >= 4 else 0
    if C >= 1 and C <= 0.3 * G:
        g2c = rng.integers(0, C, G)
        g2c[rng.integers(0, G, max(1, G // 10))] = -1     # some Gaussians in no cluster
        pairs = [(int(i), int(c)) for i, c in enumerate(g2c) if c >= 0]
        minc, ming = float(rng.choice([0.0, 0.1, 0.5])), float(rng.choice([0.0, 0.1, 0.3, 0.7]))
        om.set_clustering(C, pairs, minc, ming)
        wantc, cnt = om.score_clustered(frames.astype(np.float64), want_counts=True)
        try:
            g.set_clustering(C, pairs)
            g.set_clustering_min_evals(minc, ming)
            for prec in (0, 3):
                g.set_precision(prec)
                err = float(np.abs(g.score(frames) - wantc).max())
                key = "clustered prec=%d" % prec
                worst[key] = max(worst.get(key, 0.0), err)
                same = np.array_equal(g.cluster_exact_counts(F), cnt)
                if err > 1e-4 or not same:
                    bad += 1
                    print("FAIL", key, "it", it, "D", D, "S", S, "G", G, "C", C, minc, ming, "err", err, "counts", same)
        except capi.AasrError as e:
            print("skip clustering:", e)
for k in sorted(worst):
    print("%-32s worst |err| %.3g" % (k, worst[k]))
print("failures:", bad)
sys.exit(1 if bad else 0)
