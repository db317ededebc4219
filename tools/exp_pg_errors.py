#!/usr/bin/env python
"""exp_pg_errors.py -- per-state visible error of the multi-pivot two-term path against the oracle, next to the state's
conditioning around its group's pivot (which estimate predicts the error?).  GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from aaltoasr_amd import capi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tools.exp_pivot_groups import bench_features  # noqa: E402

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "speech"
    n_utts = int(os.environ.get("EXP_UTTS", "120"))
    S = int(os.environ.get("EXP_STATES", "1000"))
    normalise = int(os.environ.get("EXP_NORM", "0"))
    torch.cuda.set_device(0)
    capi.check(capi.lib().aasr_set_device(0))
    X = bench_features(kind, n_utts)
    if normalise:
        X = ((X - X.mean(0)) / X.std(0)).astype(np.float32)
    model = synth.fit_model(X, S=S, comps=16)
    mean, var, off, idx, w = model
    g = capi.Gmm.from_arrays(*model)
    print("parts", g.engine_parts(), "f16 states", g.precision_states())
    print("plan", g.engine_plan_note())
    lay = g.engine_layout(0)
    if lay is None:
        sys.exit(0)
    colmap, col0, begin, real_end, piv = lay
    # frames: near the Gaussians of sampled states (their own training frames are the natural probes)
    rng = np.random.default_rng(3)
    fi = np.sort(rng.choice(X.shape[0], 3000, replace=False))
    fr = np.ascontiguousarray(X[fi])
    ref = O.DiagModel(*model).score(fr.astype(np.float64))
    got = g.score(fr)
    vis = ref > -103.0
    err = np.where(vis, np.abs(got - ref), 0.0)
    st_err = err.max(0)
    # conditioning of every state around its group's pivot
    grp = np.full(S, -1)
    for s in range(S):
        c = colmap[s] - col0
        for p in range(len(begin)):
            if begin[p] <= c < real_end[p] and colmap[s] < col0 + real_end[-1]:
                grp[s] = p
    k1 = np.zeros(S); k2 = np.zeros(S); kinf = np.zeros(S); kx = np.zeros(S)
    for s in range(S):
        if grp[s] < 0:
            continue
        gi = idx[off[s]:off[s + 1]]
        t = (mean[gi] - piv[grp[s]].astype(np.float64)) ** 2 / var[gi]
        k1[s] = t.sum(1).max(); k2[s] = np.sqrt((t * t).sum(1)).max(); kinf[s] = t.max()
    in0 = grp >= 0
    ep = g.engine_parts()
    c0 = 0
    for i, part in enumerate(ep["parts"]):
        lay_i = g.engine_layout(i)
        lo = lay_i[1]
        hi = ep["parts"][i + 1:] and g.engine_layout(i + 1)[1] or ep["cols"]
        ms = (colmap >= lo) & (colmap < hi)
        vv = vis[:, ms]
        print("part %d (arith %d): %d states, max visible err %.3g, p99.99 %.3g" % (
            i, part["arith"], ms.sum(), err[:, ms].max() if ms.any() else 0,
            np.quantile(err[:, ms][vv], 0.9999) if vv.any() else 0))
    print("states in part 0: %d; visible values %d" % (in0.sum(), vis.sum()))
    order = np.argsort(-st_err)
    print("worst states: err, grp, kappa, kappa2, kappa_inf, n visible")
    for s in order[:25]:
        print("  s=%4d err %.3g grp %2d k %.0f k2 %.1f kinf %.1f vis %d" % (s, st_err[s], grp[s], k1[s], k2[s], kinf[s], vis[:, s].sum()))
    m = in0 & (vis.sum(0) > 0)
    for name, v in (("kappa", k1), ("kappa2", k2), ("kappa_inf", kinf)):
        print("corr(err, %s) over part-0 states with visible values: %.3f" % (name, np.corrcoef(st_err[m], v[m])[0, 1]))
    bad = m & (st_err > 5e-5)
    print("states over 5e-5: %d of %d; over 1e-4: %d" % (bad.sum(), m.sum(), (m & (st_err > 1e-4)).sum()))
    if bad.any():
        print("  their kappa2 range %.1f..%.1f, kinf %.1f..%.1f" % (k2[bad].min(), k2[bad].max(), kinf[bad].min(), kinf[bad].max()))
    # error by depth below the state's best component's peak and by the frame's own conditioning around the pivot
    prec = 1.0 / var
    peak = 0.5 * np.log(prec).sum(1)[idx] + np.log(w)        # per component
    depth = np.zeros_like(err); tfr = np.zeros_like(err)
    for s in np.flatnonzero(in0):
        gi = idx[off[s]:off[s + 1]]
        pk = peak[off[s]:off[s + 1]].max()
        depth[:, s] = pk - ref[:, s]
        xp = fr.astype(np.float64) - piv[grp[s]]
        pm = prec[gi].max(0)                                   # the state's largest precision per dimension
        t = xp * xp * pm
        tfr[:, s] = np.sqrt((t * t).sum(1))
    sel = vis & in0[None, :]
    print("max err by depth (peak - ll) in part 0:")
    for lo, hi in ((0, 20), (20, 40), (40, 60), (60, 80), (80, 100), (100, 150), (150, 400)):
        mm = sel & (depth >= lo) & (depth < hi)
        if mm.any():
            print("   depth %3d-%3d: n %8d max err %.3g  p99.9 %.3g" % (lo, hi, mm.sum(), err[mm].max(), np.quantile(err[mm], 0.999)))
    print("max err by the frame's own 2-norm of p x'^2 (largest precision of the state):")
    for lo, hi in ((0, 40), (40, 80), (80, 120), (120, 160), (160, 240), (240, 1e9)):
        mm = sel & (tfr >= lo) & (tfr < hi)
        if mm.any():
            print("   t2 %3d-%3.0f: n %8d max err %.3g  p99.9 %.3g" % (lo, min(hi, 999), mm.sum(), err[mm].max(), np.quantile(err[mm], 0.999)))
    # per frame: the LNA window (within 36 of the frame's best state)
    best = ref.max(1, keepdims=True)
    win = sel & (ref > best - 36.0)
    print("inside the 2-byte LNA window (ll > frame max - 36): n %d max err %.3g" % (win.sum(), err[win].max() if win.any() else 0))
    # where the error sits for the worst (frame, state)
    f, s = np.unravel_index(np.argmax(err), err.shape)
    gi = idx[off[s]:off[s + 1]]
    z = (fr[f].astype(np.float64) - mean[gi]) / np.sqrt(var[gi])
    xp = fr[f].astype(np.float64) - piv[max(grp[s], 0)]
    print("worst value: frame %d state %d ref %.4f got %.4f; nearest comp |z|^2 %.1f; max |x'|/sd_global %.1f" % (
        f, s, ref[f, s], got[f, s], (z * z).sum(1).min(), np.abs(xp / X.std(0)).max()))
    t = (xp ** 2) / var[gi]
    print("   frame's own p x'^2 per comp: sum max %.0f, 2-norm max %.1f" % (t.sum(1).max(), np.sqrt((t * t).sum(1)).max()))
