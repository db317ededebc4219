#!/usr/bin/env python
"""exp_pivot_groups.py -- multi-pivot engine parts on fitted models: what the planner builds, parity against the oracle on
sampled frames, timing of the engine path (frames -> LNA codes) against the BASELINE model's.

    python tools/exp_pivot_groups.py [small|full|speech] ..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from aaltoasr_amd import capi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def blobs(F, D=39, seed=1, n_blobs=6, spread=3.0):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((n_blobs, D)) * spread
    cent[0] = 0
    sc = np.exp(rng.uniform(np.log(0.3), np.log(1.5), (n_blobs, D)))
    which = rng.integers(0, n_blobs, F)
    return (cent[which] + rng.standard_normal((F, D)) * sc[which]).astype(np.float32)


def check(model, frames, label, n_oracle=64, time_it=False):
    t0 = time.time()
    g = capi.Gmm.from_arrays(*model)
    t_build = time.time() - t0
    parts = g.engine_parts()
    k, k2 = synth.conditioning(model[0], model[1])
    print("[%s] S=%d G=%d build %.2f s  kappa max %.0f kappa2 max %.0f  eff prec %d  f16 states %s  parts %s" % (
        label, g.num_states, model[0].shape[0], t_build, k.max(), k2.max(), g.effective_precision(), g.precision_states(), parts),
        flush=True)
    print("   plan:", g.engine_plan_note(), flush=True)
    rng = np.random.default_rng(5)
    idx = np.sort(rng.choice(frames.shape[0], min(n_oracle, frames.shape[0]), replace=False))
    sub = np.ascontiguousarray(frames[idx])
    ref = O.DiagModel(*model).score(sub.astype(np.float64))
    got = g.score(sub)
    vis = ref > -103.0
    err = np.abs(got - ref)[vis].max()
    print("   public layout: max |dll| over %d visible values = %.3g (worst all %.3g)" % (vis.sum(), err, np.abs(got - ref).max()), flush=True)
    dev = torch.device("cuda:0")
    d_f = torch.from_numpy(sub).to(dev)
    S = g.num_states
    d_scr = torch.empty(g.score_scratch_floats(sub.shape[0]), dtype=torch.float32, device=dev)
    d_by = torch.empty((sub.shape[0], S * 4), dtype=torch.uint8, device=dev)
    g.score_lna_dev(d_f, d_scr, d_by, True, 4)
    torch.cuda.synchronize()
    lp = d_by.cpu().numpy().view("<f4").astype(np.float64).reshape(sub.shape[0], S)
    lik = np.exp(ref)
    lp_ref, _ = O.lna_encode(lik, True, 4)
    e2 = np.abs(lp - lp_ref)[lp_ref > -80].max()
    print("   engine path (4-byte LNA): max |dlp| = %.3g" % e2, flush=True)
    if time_it:
        d_all = torch.from_numpy(frames).to(dev)
        F = frames.shape[0]
        d_scr = torch.empty(g.score_scratch_floats(F), dtype=torch.float32, device=dev)
        d_by = torch.empty((F, S * 2), dtype=torch.uint8, device=dev)
        for _ in range(2):
            g.score_lna_dev(d_all, d_scr, d_by, True, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.score_lna_dev(d_all, d_scr, d_by, True, 2)
        e1.record()
        torch.cuda.synchronize()
        print("   engine path frames -> 2-byte LNA: %.3f ms per %d frames" % (e0.elapsed_time(e1) / 5, F), flush=True)
        if not g.score_pitch_ok():
            return err, e2
        d_out = torch.empty((F, (S + 31) // 32 * 32), dtype=torch.float32, device=dev)
        for _ in range(2):
            g.score_dev_pitched(d_all, d_out, d_out.shape[1])
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            g.score_dev_pitched(d_all, d_out, d_out.shape[1])
        e1.record()
        torch.cuda.synchronize()
        print("   public layout scoring: %.3f ms" % (e0.elapsed_time(e1) / 5), flush=True)
    g.close()
    return err, e2


def bench_features(kind, n_utts=360):
    from aaltoasr_amd.pipeline import FullChainBench
    dev = torch.device("cuda:0")
    mk = synth.make_speechlike_audio if kind == "speech" else synth.make_audio
    utts = [mk(160000, seed=synth.SEED + 7000 + i) for i in range(n_utts)]
    g0 = capi.Gmm.from_arrays(*synth.make_model(D=39, G=256, S=32, comps=8))
    r = FullChainBench(g0, n_utts, 10.0, 0, dev, utts=utts)
    r.features_only()
    torch.cuda.synchronize()
    X = r.d_fea.cpu().numpy()
    r.release()
    g0.close()
    return X


if __name__ == "__main__":
    what = sys.argv[1:] or ["small"]
    torch.cuda.set_device(0)
    capi.check(capi.lib().aasr_set_device(0))
    if "small" in what:
        X = blobs(30000)
        for S, comps in ((96, 8), (200, 16), (64, 3)):
            m = synth.fit_model(X, S=S, comps=comps)
            check(m, X[:4096], "blobs S=%d x %d" % (S, comps))
    if "routing" in what:
        import bench
        model = synth.make_model(D=39, G=50000, S=3125, comps=16)
        F = 449280
        dev = torch.device("cuda:0")
        d_all = torch.randn((F, 39), device=dev, dtype=torch.float32)
        d_by = torch.empty((F, 3125 * 2), dtype=torch.uint8, device=dev)
        rng = np.random.default_rng(synth.SEED + 99)
        base = None
        for share in (0.0, 0.01, 0.10, 0.40):
            bad = sorted(rng.choice(3125, int(round(share * 3125)), replace=False).tolist()) if share else []
            g = capi.Gmm.from_arrays(*(synth.push_states_over_the_f16_limits(model, bad) if bad else model))
            d_scr = torch.empty(g.score_scratch_floats(F), dtype=torch.float32, device=dev)
            for _ in range(2):
                g.score_lna_dev(d_all, d_scr, d_by, True, 2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.score_lna_dev(d_all, d_scr, d_by, True, 2)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            if base is None:
                base = ms
            print("share %.2f: engine path %.3f ms (+%.3f; scoring x%.3f at 8.5 ms)  f16 states %s parts %s" % (
                share, ms, ms - base, (8.5 + ms - base) / 8.5, g.precision_states(), g.engine_parts()), flush=True)
            print("    plan:", g.engine_plan_note(), flush=True)
            g.close()
    n_utts = int(os.environ.get("EXP_UTTS", "360"))
    n_states = int(os.environ.get("EXP_STATES", "3125"))
    for kind in ("full", "speech"):
        if kind in what:
            t0 = time.time()
            X = bench_features(kind, n_utts)
            print("features %s: %s in %.1f s, std %.3g..%.3g" % (kind, X.shape, time.time() - t0, X.std(0).min(), X.std(0).max()), flush=True)
            if int(os.environ.get("EXP_NORM", "1")):
                X = ((X - X.mean(0)) / X.std(0)).astype(np.float32)
            t0 = time.time()
            m = synth.fit_model(X, S=n_states, comps=16)
            print("fit_model: %.1f s" % (time.time() - t0), flush=True)
            check(m, X, "fitted %s" % kind, time_it=True)
            if "base" in what:
                check(synth.make_model(D=39, G=50000, S=3125, comps=16), X, "BASELINE model on %s features" % kind, time_it=True)
