"""Host-side cost of building / re-adapting a configs[1]-sized model."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth
model = synth.make_model(D=39, G=50000, S=3125, comps=16)
t = time.time(); g = capi.Gmm.from_arrays(*model); print("create_diag      %.3f s" % (time.time() - t))
W = np.zeros((1, 39, 40)); W[0, :, 1:] = np.eye(39) * 1.01; W[0, :, 0] = 0.1
g2t = np.zeros(50000, np.int32)
for k in range(3):
    t = time.time(); g.set_cmllr(g2t, W * (1 + 0.01 * k)); print("set_cmllr global %.3f s" % (time.time() - t))
t = time.time(); g.set_cmllr(); print("reset            %.3f s" % (time.time() - t))
