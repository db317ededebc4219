"""Diagnostic: latency of one utterance through the whole path (aasr_run_utterance: H2D copy, MFCC
chain, 50 k-Gaussian scoring, LNA packing, D2H copy) for a few lengths."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aaltoasr_amd import capi, synth
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ft = capi.Feat.from_file(os.path.join(here, "tests", "golden", "mfcc_cms_norm.feaconf"))
S, comps = 3125, 16
gm = capi.Gmm.from_arrays(*synth.make_model(D=39, G=S * comps, S=S, comps=comps))
for sec in (1, 2, 5, 10, 30):
    pcm = synth.make_audio(16000 * sec, seed=sec)
    for _ in range(3): capi.run_utterance(ft, gm, pcm)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): lna, frames = capi.run_utterance(ft, gm, pcm)
    dt = (time.perf_counter() - t0) / n
    print("%2d s of audio, %4d frames: %.3f ms per utterance (%.0f x real time)" % (sec, frames, dt * 1e3, sec / dt))
