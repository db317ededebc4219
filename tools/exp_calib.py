"""exp_calib.py -- calibration sweep of the expanded (matrix-core) forms' error against the oracle, per state, next to
candidate conditioning measures.  Needs a library built with the conditioning limits wide open, so that EVERY state is
scored by the split-term kernels whatever its conditioning (build and run: tools/exp_calib.sh); GPU.

For every model (the population of tools/fuzz_fitted.py: fitted to random blobs) three sets of frames -- on the data,
0.7 and 1.5 sigma off the data -- are scored by the one-pivot two-term (fp16) and three-term (bf16) forms; per state the
largest visible error, the frame it occurred on and the measures of the (frame, dominant component) pair are recorded:

  k_d = (mu_d - pivot_d) / sd_d,  u_d = (x_d - pivot_d) / sd_d,  z_d = u_d - k_d
  kappa = sum k_d^2, kappa2 = |k^2|_2, kinf = max k_d^2
  L2 = |k u|_2, Q2 = |u^2 / 2|_2, L1 = |k u|_1, Q1 = |u^2|_1 / 2, z2 = |z|^2

Output: gpurun_out/calib/calib_<tag>.npz with one row per (model, frame set, form, state).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VISIBLE = -103.0


def pair_measures(x, mean, var, pivot):
    """measures of frame x against every Gaussian in mean / var ([n, D]) around `pivot`"""
    sd = np.sqrt(var)
    k = (mean - pivot) / sd
    u = (x[None, :] - pivot) / sd
    z = u - k
    ku = k * u
    q = 0.5 * u * u
    return dict(kappa=(k * k).sum(1), kappa2=np.sqrt((k ** 4).sum(1)), kinf=(k * k).max(1),
                L2=np.sqrt((ku * ku).sum(1)), Q2=np.sqrt((q * q).sum(1)), L1=np.abs(ku).sum(1), Q1=q.sum(1),
                z2=(z * z).sum(1), umax=np.abs(u).max(1), LQ2=np.sqrt(((ku - q) ** 2).sum(1)))


def run(seed0, n_seeds, per_seed, tag):
    import torch  # noqa: F401
    from aaltoasr_amd import capi, synth
    from oracle import oracle as O
    from tools.fuzz_fitted import blobs
    O.build()
    capi.check(capi.lib().aasr_set_device(0))
    os.environ["AASR_F16_PROBE_TOL"] = "1e9"   # the load-time probe moves nothing
    rows = []
    t0 = time.time()
    for seed in range(seed0, seed0 + n_seeds):
        rng = np.random.default_rng(seed)
        for it in range(per_seed):
            D = int(rng.choice([13, 24, 39, 39, 39]))
            S = int(rng.integers(24, 320))
            comps = int(rng.choice([4, 8, 16, 16]))
            F0 = max(6000, 12 * S * comps // 4)
            X = blobs(rng, F0, D)
            try:
                model = synth.fit_model(X, S=S, comps=comps, seed=int(rng.integers(1, 1 << 30)),
                                        minvar=float(rng.choice([0.1, 0.03, 0.3])))
            except ValueError:
                continue
            mean, var, off, idx, w = model
            g = capi.Gmm.from_arrays(*model)
            force_sc = os.environ.get("AASR_EXP_FORCE_SC") == "1"
            ep = g.engine_parts()
            if force_sc:
                if not ep or len(ep["parts"]) != 1 or ep["parts"][0]["arith"] != 4:
                    print("seed %d it %d: no single slab-constant part: %s %s" % (seed, it, ep, g.engine_plan_note()))
                    g.close()
                    continue
            elif ep:
                print("seed %d it %d: engine parts built (limits not open?)" % (seed, it))
                g.close()
                continue
            pivot = mean.mean(0).astype(np.float32).astype(np.float64)
            om = O.DiagModel(*model)
            nf = 1024
            base = np.ascontiguousarray(X[rng.choice(X.shape[0], nf, replace=False)])
            sets = [base]
            for amp in (0.7, 1.5):
                sets.append((base + rng.standard_normal(base.shape).astype(np.float32) * X.std(0) * amp).astype(np.float32))
            logw = np.log(np.maximum(w, 1e-300))
            cst = -0.5 * np.log(var).sum(1)
            for si, fr in enumerate(sets):
                ref = om.score(fr.astype(np.float64))
                vis = ref > VISIBLE
                for prec in ((4,) if force_sc else (4, 3)):
                    g.set_precision(prec)
                    eff = g.effective_precision()
                    n16 = g.precision_states()[0]
                    if prec == 4 and (eff != 4 or n16 != S):
                        continue   # (fp16 range: the model has no whole two-term layout)
                    got = g.score(fr)
                    err = np.where(vis, np.abs(got - ref), 0.0)
                    fi = err.argmax(0)
                    for s in range(S):
                        if not vis[:, s].any():
                            continue
                        f = int(fi[s])
                        gi = idx[off[s]:off[s + 1]]
                        pm = pair_measures(fr[f].astype(np.float64), mean[gi], var[gi], pivot)
                        llc = cst[gi] + logw[off[s]:off[s + 1]] - 0.5 * pm["z2"]
                        j = int(llc.argmax())
                        kap = ((mean[gi] - pivot) ** 2 / var[gi])
                        rows.append((seed, it, si, prec, D, s, err[f, s], ref[f, s], ref[f].max(),
                                     kap.sum(1).max(), np.sqrt((kap * kap).sum(1)).max(), kap.max(),
                                     pm["kappa"][j], pm["kappa2"][j], pm["kinf"][j], pm["L2"][j], pm["Q2"][j], pm["L1"][j],
                                     pm["Q1"][j], pm["z2"][j], pm["umax"][j], pm["LQ2"][j], cst[gi][j] + logw[off[s] + j],
                                     int(vis[:, s].sum())))
            g.close()
            print("seed %d it %d D %d S %d comps %d: %d rows, %.0f s" % (seed, it, D, S, comps, len(rows), time.time() - t0), flush=True)
    cols = ["seed", "it", "set", "prec", "D", "state", "err", "ref", "best", "st_kappa", "st_kappa2", "st_kinf", "kappa",
            "kappa2", "kinf", "L2", "Q2", "L1", "Q1", "z2", "umax", "LQ2", "peak", "nvis"]
    a = np.array(rows, np.float64)
    os.makedirs("gpurun_out/calib", exist_ok=True)
    np.savez_compressed("gpurun_out/calib/calib_%s.npz" % tag, rows=a, cols=np.array(cols))
    for prec in (4, 3):
        m = a[:, 3] == prec
        if m.any():
            print("prec %d: %d rows, max err %.3g, rows over 1e-4: %d" % (prec, m.sum(), a[m, 6].max(), (a[m, 6] > 1e-4).sum()))


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "a")
