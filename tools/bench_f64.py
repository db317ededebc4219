"""Diagnostic: the AASR_PREC_F64 scoring kernel alone (50 000 Gaussians), device time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from aaltoasr_amd import capi, synth
model = synth.make_model(D=39, G=50000, S=3125, comps=16)
g = capi.Gmm.from_arrays(*model)
L = capi.lib()
L.aasr_gmm_score_f64_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
for F in (4096, 65536):
    x = torch.randn((F, 39), dtype=torch.float64, device="cuda")
    out = torch.empty((F, 3125), dtype=torch.float64, device="cuda")
    run = lambda: capi.check(L.aasr_gmm_score_f64_dev(g._h, x.data_ptr(), F, out.data_ptr(), None))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("F=%6d  %.2f ms  %.2f M frames/s  (%.1f G pairs/s)" % (F, ms, F / ms / 1e3, F * 50000 / ms / 1e6))
