"""BASELINE configs[1] shapes with Gaussian clustering switched on (SURVEY section
8f-1): 1 M frames x 50 k Gaussians, 1000 clusters, --eval-ming 0.25.  Prints ms
per pass for the clustered path next to the plain scoring kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth

D, G, S, COMPS = 39, 50000, 3125, 16
F = int(os.environ.get("F", 1000000))
C = int(os.environ.get("C", 1000))
model = synth.make_model(D=D, G=G, S=S, comps=COMPS)
g2c = synth.make_clustering(model[0], C, iters=2)
g = capi.Gmm.from_arrays(*model)
d_fr = torch.randn((F, D), device="cuda")
d_out = torch.empty((F, S), device="cuda")
# rows padded to whole cache lines where the kernels carry a pitch (what bench.py and the recipe driver do)
PITCH = (S + 31) // 32 * 32 if os.environ.get("PITCH", "1") == "1" else S
d_out_p = torch.empty((F, PITCH), device="cuda") if PITCH != S else d_out


def score():
    if PITCH != S and g.score_pitch_ok():
        g.score_dev_pitched(d_fr, d_out_p, PITCH)
    else:
        g.score_dev(d_fr, d_out)


def run(label, reps=3):
    for _ in range(1):
        score()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        score()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-34s %8.2f ms/pass  %6.2f M frames/s" % (label, ms, F / ms / 1e3), flush=True)


NAMES = {0: "f32", 3: "bf16x3", 4: "f16x2"}
PRECS = [int(x) for x in os.environ.get("PRECS", "0,3,4").split(",")]
for prec in PRECS:
    g.set_precision(prec)
    run("exact, %s tracks kernel" % NAMES[prec])
g.set_clustering(C, [(i, int(c)) for i, c in enumerate(g2c)])
for prec in PRECS:
    g.set_precision(prec)
    for minc, ming in ((0.0, 0.1), (0.0, 0.25)):
        g.set_clustering_min_evals(minc, ming)
        run("clustered %s C=%d ming=%.2f" % (NAMES[prec], C, ming))
        n = g.cluster_exact_counts(1000)
        print("   clusters evaluated exactly per frame: mean %.1f; frames left to the queue replay in the last sub-pass: %d" % (n.mean(), g.cluster_tie_frames()))
