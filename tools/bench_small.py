"""Diagnostic: latency of small scoring batches (a decoder's per-utterance blocks), 50 k Gaussians."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth
S, comps = 3125, 16
g = capi.Gmm.from_arrays(*synth.make_model(D=39, G=S * comps, S=S, comps=comps))
for F in (256, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    d_fr = torch.randn((F, 39), device="cuda"); d_out = torch.empty((F, S), device="cuda")
    for _ in range(3): g.score_dev(d_fr, d_out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.score_dev(d_fr, d_out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("F=%6d  %.3f ms  %.2f M frames/s" % (F, ms, F / ms / 1e3))
