"""Engine path of the configs[2] model with a share of its states pushed over the two-term limits (bench.py's
precision_routing models): parts, plan, ms; for kernel traces (tools/kstats.sh routed -- python tools/bench_routed.py 0.01)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth

share = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
D, G, S, COMPS = 39, 50000, 3125, 16
F = 449280
capi.check(capi.lib().aasr_set_device(0))
model = synth.make_model(D=D, G=G, S=S, comps=COMPS)
rng = np.random.default_rng(synth.SEED + 99)
for sh in (0.01, 0.10, 0.40):
    bad = sorted(rng.choice(S, max(1, int(round(sh * S))), replace=False).tolist())
    if abs(sh - share) < 1e-9:
        break
d_f = torch.randn((F, D), device="cuda")
d_by = torch.empty((F, S * 2), dtype=torch.uint8, device="cuda")
for name, mdl in (("all f16x2", model), ("%.0f %% pushed" % (100 * share), synth.push_states_over_the_f16_limits(model, bad))):
    g = capi.Gmm.from_arrays(*mdl)
    print(name, "parts", g.engine_parts())
    print("   plan:", g.engine_plan_note())
    d_scr = torch.empty(g.score_scratch_floats(F), dtype=torch.float32, device="cuda")
    g.score_lna_dev(d_f, d_scr, d_by, True, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.score_lna_dev(d_f, d_scr, d_by, True, 2)
    e1.record()
    torch.cuda.synchronize()
    print("   engine path: %.3f ms" % (e0.elapsed_time(e1) / 5))
    g.close()
    # the public layout (column = state), rows padded to whole lines: what bench.py's scoring_ms times
if len(sys.argv) > 2 and sys.argv[2] == "public":
    g = capi.Gmm.from_arrays(*synth.push_states_over_the_f16_limits(model, bad))
    pitch = (S + 31) // 32 * 32
    d_ll = torch.empty((F, pitch), dtype=torch.float32, device="cuda")
    g.score_dev_pitched(d_f, d_ll, pitch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.score_dev_pitched(d_f, d_ll, pitch)
    e1.record()
    torch.cuda.synchronize()
    print("   public layout: %.3f ms" % (e0.elapsed_time(e1) / 5))
