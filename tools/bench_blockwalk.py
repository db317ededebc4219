"""Diagnostic: walking a long recording block by block through aasr_feat_run (what
aku::FeatureGenerator::generate does on a block miss): ms per 256-frame block."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aaltoasr_amd import capi, synth
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ft = capi.Feat.from_file(os.path.join(here, "tests", "golden", "mfcc_cms_norm.feaconf"))
for minutes in (1, 10, 60):
    pcm = synth.make_audio(16000 * 60 * minutes, seed=minutes)
    last = ft.last_frame(len(pcm))
    ft.run(pcm, 0, 256)
    blocks = list(range(0, last + 1, 256))[:400]
    t0 = time.perf_counter()
    for b in blocks:
        ft.run(pcm, b, min(256, last + 1 - b))
    dt = time.perf_counter() - t0
    print("%2d min of audio (%d MB): %.3f ms per 256-frame block" % (minutes, pcm.nbytes >> 20, 1e3 * dt / len(blocks)))
