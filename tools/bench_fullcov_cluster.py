"""Clustered scoring over the configs[4] pool (10 000 full-covariance Gaussians, 200 000 frames, 200 clusters, --eval-ming 0.25):
ms per pass per precision (4: fp16 rows with masks, 3: bf16x3 rows with masks, 0: the f32 masked kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth
D, G, S, COMPS, F, C = 39, 10000, 625, 16, 200000, 200
rng = np.random.default_rng(synth.SEED)
mean = rng.standard_normal((G, D))
a = rng.standard_normal((G, D, D)) * 0.3
cov = a @ a.transpose(0, 2, 1) + 0.1 * np.eye(D)
_, _, off, idx, w = synth.make_model(D=D, G=G, S=S, comps=COMPS)
g = capi.Gmm.from_full(mean, cov, off, idx, w)
g2c = synth.make_clustering(mean, C, iters=2)
g.set_clustering(C, [(i, int(c)) for i, c in enumerate(g2c)])
g.set_clustering_min_evals(0.0, 0.25)
d_fr = torch.randn((F, D), device="cuda")
d_out = torch.empty((F, S), device="cuda")
for prec in (4, 3, 0):
    g.set_precision(prec)
    g.score_dev(d_fr, d_out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.score_dev(d_fr, d_out)
    e1.record()
    torch.cuda.synchronize()
    print("clustered full-covariance pool, precision %d: %.2f ms per pass" % (prec, e0.elapsed_time(e1) / 3), flush=True)
