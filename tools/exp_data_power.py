"""Does the DATA set the scoring kernel's pace?  The BASELINE model (synth.make_model) and the stationary fitted model, each
scored (engine path: frames -> 2-byte LNA codes) on the fitted model's standardised features and on N(0, 1) frames."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth, pipeline
D, S, COMPS = 39, 3125, 16
capi.check(capi.lib().aasr_set_device(0))
base_small = capi.Gmm.from_arrays(*synth.make_model(D=D, G=256, S=32, comps=8))
utts = [synth.make_audio(160000, seed=synth.SEED + 7000 + i) for i in range(360)]
runner = pipeline.FullChainBench(base_small, 360, 10.0, 0, torch.device("cuda", 0), utts=utts)
runner.features_only(); torch.cuda.synchronize()
X = runner.d_fea.cpu().numpy(); X = ((X - X.mean(0)) / X.std(0)).astype(np.float32)
F = X.shape[0]
fitted = capi.Gmm.from_arrays(*synth.fit_model(X, S=S, comps=COMPS))
basem = capi.Gmm.from_arrays(*synth.make_model(D=D, G=S * COMPS, S=S, comps=COMPS))
frames = {"fitted features": torch.from_numpy(X).cuda(), "N(0,1) frames": torch.randn((F, D), device="cuda")}
d_by = torch.empty((F, S * 2), dtype=torch.uint8, device="cuda")
for mname, g in (("BASELINE model", basem), ("fitted model", fitted)):
    d_scr = torch.empty(g.score_scratch_floats(F), dtype=torch.float32, device="cuda")
    for fname, d_f in frames.items():
        for _ in range(2): g.score_lna_dev(d_f, d_scr, d_by, True, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): g.score_lna_dev(d_f, d_scr, d_by, True, 2)
        e1.record(); torch.cuda.synchronize()
        print("%-15s on %-16s: %.3f ms per %d frames" % (mname, fname, e0.elapsed_time(e1) / 8, F), g.engine_parts() and [p["pivot_groups"] for p in g.engine_parts()["parts"]])
