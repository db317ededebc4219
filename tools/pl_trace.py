"""Phase trace of the pipelined scoring kernel (an experiment build with AASR_PL_TRACE: AASR_BUILD_DEFINES=AASR_PL_TRACE=1
AASR_BUILD_LIBDIR=lib_trace python -m aaltoasr_amd.build; run with AASR_LIBDIR=aaltoasr_amd/lib_trace): one workgroup's
eight waves sum the shader-clock intervals of their tile loop.  Prints per wave and per tile: matrix phases H0 / H1, the
close logic behind each, the tile barrier, and what the matrix pipe could do in that time (60 matrix instructions x 32
cycles per wave and tile; two waves share a SIMD)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth

D, G, S, COMPS = 39, 50000, 3125, 16
F = int(os.environ.get("F", 449280))
capi.check(capi.lib().aasr_set_device(0))
model = synth.make_model(D=D, G=G, S=S, comps=COMPS)
g = capi.Gmm.from_arrays(*model)
d_fr = torch.randn((F, D), device="cuda")
pitch = (S + 31) // 32 * 32
d_out = torch.empty((F, pitch), device="cuda")
for _ in range(3):
    g.score_dev_pitched(d_fr, d_out, pitch)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.score_dev_pitched(d_fr, d_out, pitch)
e1.record()
torch.cuda.synchronize()
print("scoring call: %.3f ms per %d frames (traced build)" % (e0.elapsed_time(e1) / 5, F))
L = capi.lib()
if not hasattr(L, "aasr_debug_pl_trace"):
    sys.exit("this library has no trace (build with AASR_BUILD_DEFINES=AASR_PL_TRACE=1)")
buf = (C.c_uint64 * 96)()
L.aasr_debug_pl_trace.argtypes = [C.c_void_p]
assert L.aasr_debug_pl_trace(buf) == 0
t = np.array(list(buf), dtype=np.float64).reshape(8, 12)
NT = t[:, 8].copy()
bar_own = t[:, 5].copy()
t[:, 5] = NT
names = ["H0 matrix phase", "close logic behind H0", "H1 matrix phase", "tile barrier (+ copy issue)", "prefetch + close logic behind H1"]
print("cycles per tile (shader clock), wave by wave; a wave's 60 matrix instructions are 1920 pipe cycles per tile:")
print("%-36s" % "wave" + "".join("%9d" % w for w in range(8)) + "      mean")
for k, n in enumerate(names):
    per = t[:, k] / np.maximum(t[:, 5], 1)
    print("%-36s" % n + "".join("%9.0f" % v for v in per) + "  %8.0f" % per.mean())
tot = t[:, :5].sum(1) / np.maximum(t[:, 5], 1)
print("%-36s" % "sum" + "".join("%9.0f" % v for v in tot) + "  %8.0f" % tot.mean())
print("%-36s" % "tiles" + "".join("%9.0f" % v for v in t[:, 5]))
print("%-36s" % "  of the barrier: vmcnt(0) wait" + "".join("%9.0f" % v for v in t[:, 6] / NT))
print("%-36s" % "  of the barrier: s_barrier" + "".join("%9.0f" % v for v in bar_own / NT))
print("%-36s" % "  behind H1: fragment prefetch" + "".join("%9.0f" % v for v in t[:, 9] / NT))
print("%-36s" % "  behind H1: close logic" + "".join("%9.0f" % v for v in t[:, 10] / NT))
print("%-36s" % "whole workgroup cycles" + "".join("%9.0f" % v for v in t[:, 7]))
print("matrix-pipe share of a SIMD's time (2 waves x 1920 / the pair's mean tile time): %.3f" % (2 * 1920.0 / tot.mean()))
