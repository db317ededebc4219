"""Error of the expanded-form kernels against the conditioning estimate kappa."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aaltoasr_amd import capi, synth
from oracle import oracle as O
O.build()
L = capi.lib(); L.aasr_debug_kappa.restype = C.c_double; L.aasr_debug_kappa.argtypes = [C.c_void_p]
rng = np.random.default_rng(int(sys.argv[1]))
rows = []
for it in range(int(sys.argv[2])):
    D = 39; S = int(rng.integers(8, 70)); comps = int(rng.integers(4, 24)); G = S * comps
    ms, fs = rng.uniform(0.3, 2.5), rng.uniform(0.5, 3.0)
    mean = rng.standard_normal((G, D)) * ms
    var = np.exp(rng.uniform(np.log(0.15), np.log(5.0), (G, D)))
    _, _, off, idx, w = synth.make_model(D=D, G=G, S=S, comps=comps, seed=it)
    frames = (rng.standard_normal((300, D)) * fs).astype(np.float32)
    want = O.DiagModel(mean, var, off, idx, w).score(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    k = L.aasr_debug_kappa(g._h)
    e = []
    for prec in (0, 3, 4):
        g.set_precision(prec)
        vis = want > -110
        e.append(float(np.abs(g.score(frames) - want)[vis].max()) if vis.any() else 0.0)
    rows.append((k, e[0], e[1], g.active_layout()))
rows.sort()
for k, a, b, lay in rows:
    print("kappa %7.1f  f32 %.2e  bf16x3 %.2e  layout %d  f32/kappa %.2e bf16/kappa %.2e" % (k, a, b, lay, a / k, b / k))
