"""BASELINE configs[4] in one call: D=39, G=10 000 full-covariance Gaussians,
F=200 000 frames -> state log-likelihoods (k_gmm_full_score).  Prints ms per
launch and algorithmic TFLOP/s (d(d+3) = 1638 flop per frame x Gaussian pair,
SURVEY.md section 8d)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth

D, G, S, COMPS, F = 39, 10000, 625, 16, 200000
rng = np.random.default_rng(synth.SEED)
mean = rng.standard_normal((G, D))
a = rng.standard_normal((G, D, D)) * 0.3
cov = a @ a.transpose(0, 2, 1) + 0.1 * np.eye(D)
_, _, off, idx, w = synth.make_model(D=D, G=G, S=S, comps=COMPS)
g = capi.Gmm.from_full(mean, cov, off, idx, w)
PREC = int(sys.argv[1]) if len(sys.argv) > 1 else 4   # 4: f16x2 (two fp16 terms), 3: bf16x3 (three-term split), 0: f32 matrix kernel
g.set_precision(PREC)
d_fr = torch.randn((F, D), device="cuda")
d_out = torch.empty((F, S), device="cuda")
for _ in range(2):
    g.score_dev(d_fr, d_out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    g.score_dev(d_fr, d_out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
flop = float(D * (D + 3)) * F * G
print({4: "f16x2 (effective %d) " % g.effective_precision(), 3: "bf16x3 ", 0: "f32 "}[PREC] + "full-cov: %.2f ms/launch, %.2f M frames/s, %.1f TFLOP/s algorithmic (%.3f of 157.3)" % (
    ms, F / ms / 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3))
