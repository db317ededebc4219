"""Diagnostic: configs[1] with a different number of states (row alignment of the
[F x S] output).  Prints ms per pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth
S = int(sys.argv[1]); comps = 16
g = capi.Gmm.from_arrays(*synth.make_model(D=39, G=S * comps, S=S, comps=comps))
g.set_precision(3)
F = 1000000
d_fr = torch.randn((F, 39), device="cuda"); d_out = torch.empty((F, S), device="cuda")
for _ in range(2): g.score_dev(d_fr, d_out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): g.score_dev(d_fr, d_out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("S=%d  %.2f ms  %.3f ns per frame x Gaussian" % (S, ms, ms * 1e6 / (F * S * comps)))
