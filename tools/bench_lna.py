"""Diagnostic: the LNA kernel alone (449 280 frames x 3 125 states, 2-byte codes), dense rows and
the engine's line-padded rows (pitch 3 136 floats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi
F, S = 449280, 3125
for pitch in (S, 3136):
    buf = (torch.randn((F, pitch), device="cuda") * 8 - 60).contiguous()
    by = torch.empty((F, S * 2), dtype=torch.uint8, device="cuda")
    run = (lambda: capi.lna_encode_dev(buf, True, 2, None, by)) if pitch == S else \
          (lambda: capi.lna_encode_dev(buf, True, 2, None, by, num_states=S))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("LNA kernel, pitch %d: %.3f ms, %.2f TB/s" % (pitch, ms, F * S * 6 / ms / 1e9))
