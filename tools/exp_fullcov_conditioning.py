"""How the full-covariance kernels' error grows with the model's conditioning: Sigma scaled down
by `shrink` (sigma ~ sqrt(shrink)), frames a few sigma from the means.  Printed: worst |dll| of the
f32 and bf16x3 kernels against oracle.FullModel for ll > -104, beside kappa = max_g |R^-1 (mu - pivot)|^2."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aaltoasr_amd import capi, synth
from oracle import oracle as O
O.build()
D, G, S, comps = 13, 48, 12, 4
for shrink in (1.0, 1e-1, 1e-2, 1e-3, 1e-4):
    rng = np.random.default_rng(7)
    mean = rng.standard_normal((G, D)) * 1.5
    cov = np.empty((G, D, D))
    for g in range(G):
        a = rng.standard_normal((D, D)) * 0.35
        cov[g] = (a @ a.T + 0.1 * np.eye(D) + np.diag(rng.uniform(0.2, 1.0, D))) * shrink
    _, _, off, idx, w = synth.make_model(D=D, G=G, S=S, comps=comps, seed=3)
    pick = rng.integers(0, G, 400)
    L = np.linalg.cholesky(cov[pick])
    z = rng.standard_normal((400, D))
    z *= rng.uniform(2, 9, (400, 1)) / np.linalg.norm(z, axis=1, keepdims=True)
    frames = (mean[pick] + np.einsum("nij,nj->ni", L, z)).astype(np.float32)
    ref = O.FullModel(mean, cov, off, idx, w).score(frames.astype(np.float64))
    try:
        gm = capi.Gmm.from_full(mean, cov, off, idx, w)
    except capi.AasrError as e:
        print("shrink %.0e: refused (%s)" % (shrink, e))
        continue
    vis = ref > -103.97
    out = []
    for prec in (0, 3, 4):
        gm.set_precision(prec)
        out.append(float(np.abs(gm.score(frames) - ref)[vis].max()))
    print("shrink %.0e: visible %d, worst |dll| f32 %.3g bf16x3 %.3g f16x2 %.3g (effective precision %d, kappa %s)" % (shrink, int(vis.sum()), out[0], out[1], out[2], gm.effective_precision(), "-"))
