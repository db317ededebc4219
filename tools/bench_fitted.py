"""Engine path (frames -> 2-byte LNA codes, aasr_gmm_score_lna_dev) of a model fitted to one synthetic hour of the
engine's own features, for kernel traces: python tools/bench_fitted.py [speechlike|stationary] [reps]
(tools/kstats.sh fitted -- python tools/bench_fitted.py speechlike 5)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth, pipeline

kind = sys.argv[1] if len(sys.argv) > 1 else "speechlike"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
D, S, COMPS = 39, 3125, 16
capi.check(capi.lib().aasr_set_device(0))
base = capi.Gmm.from_arrays(*synth.make_model(D=D, G=256, S=32, comps=8))
mk = synth.make_speechlike_audio if kind == "speechlike" else synth.make_audio
utts = [mk(160000, seed=synth.SEED + 7000 + i) for i in range(360)]
runner = pipeline.FullChainBench(base, 360, 10.0, 0, torch.device("cuda", 0), utts=utts)
runner.features_only()
torch.cuda.synchronize()
X = runner.d_fea.cpu().numpy()
X = ((X - X.mean(0)) / X.std(0)).astype(np.float32)
model = synth.fit_model(X, S=S, comps=COMPS)
g = capi.Gmm.from_arrays(*model)
print("parts", g.engine_parts())
print(g.engine_plan_note())
F = X.shape[0]
d_f = torch.from_numpy(X).cuda()
d_scr = torch.empty(g.score_scratch_floats(F), dtype=torch.float32, device="cuda")
d_by = torch.empty((F, S * 2), dtype=torch.uint8, device="cuda")
g.score_lna_dev(d_f, d_scr, d_by, True, 2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    g.score_lna_dev(d_f, d_scr, d_by, True, 2)
e1.record()
torch.cuda.synchronize()
print("engine path: %.3f ms per %d frames" % (e0.elapsed_time(e1) / reps, F))
if len(sys.argv) > 3 and sys.argv[3] == "clustered":
    C_ = 1000
    g2c = synth.make_clustering(model[0], C_, iters=2)
    g.set_clustering(C_, [(i, int(c)) for i, c in enumerate(g2c)])
    g.set_clustering_min_evals(0.0, 0.25)
    pitch = (S + 31) // 32 * 32
    d_ll = torch.empty((F, pitch), dtype=torch.float32, device="cuda")
    g.score_dev_pitched(d_f, d_ll, pitch)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.score_dev_pitched(d_f, d_ll, pitch)
    e1.record()
    torch.cuda.synchronize()
    print("clustered: %.3f ms per %d frames" % (e0.elapsed_time(e1) / 3, F))
