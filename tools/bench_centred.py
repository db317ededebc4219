"""Diagnostic: configs[1] through the centred-form kernel (AASR_PREC_F32_CENTRED), the path
ill-conditioned models (kappa > 600) take.  Prints ms per pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth
S, comps, F = 3125, 16, 1000000
g = capi.Gmm.from_arrays(*synth.make_model(D=39, G=S * comps, S=S, comps=comps))
g.set_precision(2)
d_fr = torch.randn((F, 39), device="cuda"); d_out = torch.empty((F, S), device="cuda")
for _ in range(2): g.score_dev(d_fr, d_out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): g.score_dev(d_fr, d_out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("centred kernel: %.2f ms per 10^6 frames x 50 k Gaussians, %.2f M frames/s" % (ms, F / ms / 1e3))
