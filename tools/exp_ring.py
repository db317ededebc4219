"""Experiment: configs[2] scoring + LNA in frame chunks through a small reused score buffer
(does a cache-resident ring spare the HBM round trip of the [F x S] float matrix?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth

S, G = 3125, 50000
gmm = capi.Gmm.from_arrays(*synth.make_model(D=39, G=G, S=S, comps=16))
gmm.set_precision(3)
F = 449280
dev = torch.device("cuda", 0)
fea = torch.randn((F, 39), device=dev) * 0.3
pitch = 3136
stream = torch.cuda.current_stream()
d_bytes = torch.empty((F, S * 2), dtype=torch.uint8, device=dev)
ref = None
for chunk in [F, 131072, 65536, 32768, 16384, 8192, 4096]:
    d_ll = torch.empty((min(chunk, F), pitch), dtype=torch.float32, device=dev)

    def step():
        for f0 in range(0, F, chunk):
            n = min(chunk, F - f0)
            gmm.score_dev_pitched(fea[f0:f0 + n], d_ll[:n], pitch, stream)
            capi.lna_encode_dev(d_ll[:n], True, 2, None, d_bytes[f0:f0 + n], stream, num_states=S)
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        step()
    e1.record(); torch.cuda.synchronize()
    cur = d_bytes.clone()
    if ref is None:
        ref = cur
    print("chunk %7d: %.3f ms per pass, identical %s" % (chunk, e0.elapsed_time(e1) / 3, bool(torch.equal(ref, cur))), flush=True)
