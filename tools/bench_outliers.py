"""Diagnostic: configs[1] with a fraction of ill-conditioned Gaussians (sigma scaled down until
kappa >> 600).  With outlier routing the model stays on the matrix path; AASR_OUTLIER_ROUTING=0
shows the former behaviour (the whole model in the centred form)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
S, comps, F = 3125, 16, 1000000
mean, var, off, idx, w = synth.make_model(D=39, G=S * comps, S=S, comps=comps)
rng = np.random.default_rng(1)
bad = rng.choice(S * comps, int(frac * S * comps), replace=False)
var[bad] *= 2e-3
g = capi.Gmm.from_arrays(mean, var, off, idx, w)
d_fr = torch.randn((F, 39), device="cuda"); d_out = torch.empty((F, S), device="cuda")
for _ in range(2): g.score_dev(d_fr, d_out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): g.score_dev(d_fr, d_out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("%.1f %% outlier Gaussians, kernel layout %d: %.2f ms per 10^6 frames, %.2f M frames/s" % (
    100 * frac, g.active_layout(), ms, F / ms / 1e3))
