"""Clustered scoring for LNA consumers (aasr_gmm_score_lna_dev): frames -> 2-byte codes next to the
unclustered figure (449 280 frames, 50 k Gaussians, 1000 clusters, --eval-ming 0.25)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth

D, G, S, COMPS, F, C = 39, 50000, 3125, 16, 449280, 1000
model = synth.make_model(D=D, G=G, S=S, comps=COMPS)
g = capi.Gmm.from_arrays(*model)
d_fr = torch.randn((F, D), device="cuda") * 0.3
d_scr = torch.empty(g.score_scratch_floats(F), dtype=torch.float32, device="cuda")
d_by = torch.empty((F, S * 2), dtype=torch.uint8, device="cuda")


def run(label, reps=3):
    g.score_lna_dev(d_fr, d_scr, d_by)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.score_lna_dev(d_fr, d_scr, d_by)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-44s %8.2f ms/pass  %6.2f M frames/s" % (label, ms, F / ms / 1e3), flush=True)
    return d_by.clone()


run("unclustered: score + pack")
g.set_clustering(C, [(i, int(c)) for i, c in enumerate(synth.make_clustering(model[0], C, iters=2))])
g.set_clustering_min_evals(0.0, 0.25)
run("clustered: centres, select, expand, masked score, merge, pack")
