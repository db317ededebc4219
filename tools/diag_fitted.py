"""Where a fitted model's worst |d ll| against the oracle sits: per engine part (two fp16 terms / three bf16 terms /
ordinary model), against the value's depth below the frame's best state and the Gaussian's conditioning around its
group's pivot.  python tools/diag_fitted.py [speechlike|stationary] [frames]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aaltoasr_amd import capi, synth, pipeline
from oracle import oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "speechlike"
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 256
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
runner.release()
model = synth.fit_model(X, S=S, comps=COMPS)
g = capi.Gmm.from_arrays(*model)
parts = g.engine_parts()
print("parts", parts)
print(g.engine_plan_note())
rng = np.random.default_rng(synth.SEED + 71)
fi = np.sort(rng.choice(X.shape[0], NF, replace=False))
sub = np.ascontiguousarray(X[fi])
ref = O.DiagModel(*model).score(sub.astype(np.float64))
got = g.score(sub)
vis = ref > -103.0
err = np.abs(got - ref)
best = ref.max(1, keepdims=True)
depth = best - ref
colmap = g.engine_layout(0)[0]
col0 = [0]
for p in parts["parts"]:
    col0.append(col0[-1] + (p["states"] + 31) // 32 * 32)
# part of a state from its column
bounds = []
c = 0
for i, p in enumerate(parts["parts"]):
    lay = g.engine_layout(i)
    bounds.append(lay[1])
bounds.append(parts["cols"])
part_of = np.searchsorted(np.array(bounds[1:]), colmap, side="right")
for i, p in enumerate(parts["parts"]):
    m = vis & (part_of == i)[None, :]
    if not m.any():
        continue
    e = err[m]
    print("part %d arith %d states %d: visible %d  max %.3g  >1e-4: %d  >5e-5: %d   inside window max %.3g" % (
        i, p["arith"], p["states"], m.sum(), e.max(), (e > 1e-4).sum(), (e > 5e-5).sum(),
        err[m & (depth < 36.0)].max() if (m & (depth < 36.0)).any() else 0))
    for lo, hi in ((0, 36), (36, 60), (60, 80), (80, 200)):
        mm = m & (depth >= lo) & (depth < hi)
        if mm.any():
            print("    depth %3d-%3d: n %8d  max %.3g  rms %.3g" % (lo, hi, mm.sum(), err[mm].max(), np.sqrt((err[mm] ** 2).mean())))
# worst ten
idx = np.argsort(np.where(vis, err, 0).ravel())[::-1][:10]
for k in idx:
    f, s = divmod(int(k), S)
    print("frame %d state %d part %d  ref %.4f got %.4f  err %.3g depth %.1f" % (f, s, part_of[s], ref[f, s], got[f, s], err[f, s], depth[f, s]))
for prec, name in ((3, "bf16x3"), (0, "f32")):
    try:
        g.set_precision(prec)
        e2 = np.abs(g.score(sub) - ref)
        print("whole model %s: max %.3g" % (name, e2[vis].max()))
    except Exception as ex:
        print(name, ex)
