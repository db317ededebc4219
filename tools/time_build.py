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
# per-class transforms (regression classes): 4 classes over the pool
W4 = np.zeros((4, 39, 40))
for c in range(4):
    W4[c, :, 1:] = np.eye(39) * (1.0 + 0.01 * c); W4[c, :, 0] = 0.05 * c
g2t4 = (np.arange(50000) % 4).astype(np.int32)
for k in range(2):
    t = time.time(); g.set_cmllr(g2t4, W4 * (1 + 0.01 * k)); print("set_cmllr 4 classes %.3f s" % (time.time() - t))
fr = torch.randn((2000, 39), device="cuda"); out = torch.empty((2000, 3125), device="cuda")
g.score_dev(fr, out); torch.cuda.synchronize()
t = time.time()
for _ in range(5): g.score_dev(fr, out)
torch.cuda.synchronize(); print("score 2000 frames, 4 classes: %.2f ms" % ((time.time() - t) / 5 * 1e3))
fr = torch.randn((500000, 39), device="cuda"); out = torch.empty((500000, 3125), device="cuda")
g.score_dev(fr, out); torch.cuda.synchronize()
t = time.time()
for _ in range(3): g.score_dev(fr, out)
torch.cuda.synchronize(); dt = (time.time() - t) / 3
print("score 500000 frames, 4 classes: %.1f ms (%.2f M frames/s)" % (dt * 1e3, 0.5 / dt))
