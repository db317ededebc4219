"""Randomised parity sweep over models FITTED to data (synth.fit_model on random mixtures of well separated, differently
scaled blobs): the models whose conditioning needs the engine's pivot groups (gmm_plan_engine_parts).
Per model: the public score layout, the engine's own layout through the LNA pass (2- and 4-byte), Gaussian clustering over
the engine parts (scores and exact-evaluation counts), all against the oracle (aku/Distributions.cc:1040-1062, 2078-2086,
2684-2722; aku/phone_probs.cc:224-262).  `run(seed, n)` returns (worst error per category, failures);
`python tools/fuzz_fitted.py SEED N` exits non-zero on a failure.

Tolerances: 1e-4 on every visible value (ll > -103: every value the reference's float storage holds) of the public layout
and of the clustered pass, and on every 4-byte LNA value whose likelihood is a normal float; LNA codes never more than one
step apart (states and frames whose likelihood is a normal float); clustered counts bit-equal."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VISIBLE = -103.0
TOL = 1e-4
TOL_TAIL = TOL   # (round 5 allowed 1.5e-4 off the LNA window: gone)


def blobs(rng, F, D):
    nb = int(rng.integers(2, 9))
    spread = float(rng.uniform(1.5, 6.0))
    cent = rng.standard_normal((nb, D)) * spread
    cent[0] = 0
    lo = float(rng.uniform(0.15, 0.6))
    sc = np.exp(rng.uniform(np.log(lo), np.log(1.8), (nb, D)))
    which = rng.integers(0, nb, F)
    X = cent[which] + rng.standard_normal((F, D)) * sc[which]
    if rng.random() < 0.5:   # units other than one: the engine's column scales
        X = X * np.exp(rng.uniform(np.log(0.05), np.log(20.0), D)) + rng.standard_normal(D) * 3.0
    return X.astype(np.float32)


def run(seed=1, N=10, verbose=False):
    import torch
    from aaltoasr_amd import capi, synth
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(seed)
    worst, fails = {}, []

    def note(cat, v):
        worst[cat] = max(worst.get(cat, 0.0), float(v))

    for it in range(N):
        D = int(rng.choice([13, 24, 39, 39, 39]))
        S = int(rng.integers(24, 320))
        comps = int(rng.choice([4, 8, 16, 16]))
        F0 = max(6000, 12 * S * comps // 4)
        X = blobs(rng, F0, D)
        tag = "seed %d it %d D %d S %d comps %d" % (seed, it, D, S, comps)
        try:
            model = synth.fit_model(X, S=S, comps=comps, seed=int(rng.integers(1, 1 << 30)), minvar=float(rng.choice([0.1, 0.03, 0.3])))
        except ValueError:
            continue
        os.environ["AASR_PG_PIVOT_COST"] = str(int(rng.choice([16, 64, 256])))   # (test hook: small models would not pay for pivots)
        try:
            g = capi.Gmm.from_arrays(*model)
        finally:
            os.environ.pop("AASR_PG_PIVOT_COST", None)
        parts = g.engine_parts()
        note("models with engine parts", 1.0 if parts else 0.0)
        worst["n parts"] = worst.get("n parts", 0) + (1 if parts else 0)
        if parts:
            worst["pivot groups max"] = max(worst.get("pivot groups max", 0), parts["parts"][0]["pivot_groups"])
        nf = int(rng.choice([97, 512, 700, 2100]))
        fr = np.ascontiguousarray(X[rng.choice(X.shape[0], nf, replace=False)])
        if rng.random() < 0.3:    # frames the model has not seen, further out
            fr = (fr + rng.standard_normal(fr.shape).astype(np.float32) * X.std(0) * 0.7).astype(np.float32)
        om = O.DiagModel(*model)
        ref, lik = om.score(fr.astype(np.float64), want_lik=True)
        vis = ref > VISIBLE
        win = vis & (ref > ref.max(1, keepdims=True) - 36.0)
        got = g.score(fr)
        err = np.abs(got - ref)
        if vis.any():
            note("public visible", err[vis].max())
            if err[vis].max() > TOL_TAIL or (err[vis] <= TOL).mean() < 0.9999:
                fails.append("%s: public layout %.3g on visible values (%.5f within 1e-4) parts %s" % (tag, err[vis].max(), (err[vis] <= TOL).mean(), parts))
        if win.any():
            note("public window", err[win].max())
            if err[win].max() > TOL:
                # the worst value: its state's conditioning around the pool's pivot and how far out the frame is
                fi_, si_ = np.unravel_index(np.argmax(np.where(win, err, 0.0)), err.shape)
                mean_, var_, off_, idx_, _w = model
                gi = idx_[off_[si_]:off_[si_ + 1]]
                k1, k2 = synth.conditioning(mean_, var_)
                z2 = (((fr[fi_].astype(np.float64) - mean_[gi]) ** 2) / var_[gi]).sum(1)
                fails.append("%s: public layout %.3g inside the LNA window, parts %s; worst value ll %.1f (frame's best %.1f), state's kappa %.0f "
                             "kappa2 %.0f, frame %.1f sigma from its nearest Gaussian" % (
                                 tag, err[win].max(), parts, ref[fi_, si_], ref[fi_].max(), k1[gi].max(), k2[gi].max(), np.sqrt(z2.min())))
        # engine layout through the LNA pass
        d_f = torch.from_numpy(fr).cuda()
        d_scr = torch.empty(g.score_scratch_floats(nf), dtype=torch.float32, device="cuda")
        for nbytes in (4, 2):
            d_by = torch.empty((nf, S * nbytes), dtype=torch.uint8, device="cuda")
            g.score_lna_dev(d_f, d_scr, d_by, True, nbytes)
            torch.cuda.synchronize()
            lp_ref, by_ref = O.lna_encode(lik, True, nbytes)
            by = d_by.cpu().numpy()
            if nbytes == 4:
                lp = by.view("<f4").reshape(nf, S).astype(np.float64)
                # (ll below ln 2^-126: the reference stores a DENORMAL float likelihood -- quantised, the band
                # conftest.assert_lp_denormal_band pins to one quantum; not this sweep's subject)
                m = ref > -87.0
                if m.any():
                    e4 = np.abs(lp - lp_ref)[m].max()
                    note("lna 4-byte", e4)
                    if e4 > TOL:
                        fails.append("%s: 4-byte LNA %.3g" % (tag, e4))
            else:
                a = by.reshape(nf, S, 2).astype(np.int32)
                b = by_ref.reshape(nf, S, 2).astype(np.int32)
                ca, cb = a[..., 0] * 256 + a[..., 1], b[..., 0] * 256 + b[..., 1]
                # frames whose best state is a normal float likelihood; of those the states above the band as well: inside
                # the band one quantum of the reference's denormal float is many code steps (conftest.assert_lp_denormal_band)
                ok = (ref.max(1, keepdims=True) > -80.0) & (ref > -87.0)
                if ok.any():
                    dmax = int(np.abs(ca - cb)[ok].max())
                    note("lna code steps", dmax)
                    if dmax > 1:
                        fails.append("%s: 2-byte LNA codes %d steps apart" % (tag, dmax))
        # Gaussian clustering over the parts
        if rng.random() < 0.6:
            C = int(rng.integers(max(2, S * comps // 200), max(3, min(S * comps // 8, 400))))
            g2c = synth.make_clustering(model[0], C, seed=int(rng.integers(1, 1 << 30)), iters=2)
            pairs = [(int(i), int(c)) for i, c in enumerate(g2c)]
            minc, ming = [(0.0, 0.25), (0.2, 0.0), (0.05, 0.1)][int(rng.integers(0, 3))]
            om.set_clustering(C, pairs, minc, ming)
            nfc = min(nf, 600)
            want, want_n = om.score_clustered(fr[:nfc].astype(np.float64), want_counts=True)
            try:
                g.set_clustering(C, pairs)
                g.set_clustering_min_evals(minc, ming)
                gc = g.score(np.ascontiguousarray(fr[:nfc]))
                cnt = g.cluster_exact_counts(nfc)
            except capi.AasrError as e:
                note("clustered refused", 1.0)
                if verbose:
                    print(tag, "clustered refused:", e)
            else:
                v = want > VISIBLE
                w = v & (want > want.max(1, keepdims=True) - 36.0)
                if not np.array_equal(cnt, want_n):
                    fails.append("%s: clustered exact-evaluation counts differ (C %d minc %g ming %g)" % (tag, C, minc, ming))
                if v.any():
                    ec = np.abs(gc - want)
                    note("clustered visible", ec[v].max())
                    if ec[v].max() > TOL_TAIL or (w.any() and ec[w].max() > TOL):
                        fails.append("%s: clustered %.3g visible / %.3g window (C %d minc %g ming %g) parts %s" % (
                            tag, ec[v].max(), ec[w].max() if w.any() else 0.0, C, minc, ming, parts))
        g.close()
        if verbose:
            print(tag, parts, {k: ("%.3g" % v) for k, v in worst.items()})
    return worst, fails


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    worst, fails = run(seed, n, verbose="-v" in sys.argv)
    print("worst:", {k: float("%.3g" % v) for k, v in worst.items()})
    print("failures: %d" % len(fails))
    for f in fails:
        print("FAIL", f)
    sys.exit(1 if fails else 0)
