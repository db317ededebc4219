"""Randomised feature-graph sweep against the oracle: sample rates, window widths (incl. the
non-power-of-two ones), VTLN variants, delta widths, CMS windows, concat, utterances down to one
frame, negative and post-EOF frame ranges; every module's output is compared.  `run(seed, n)`
returns (worst relative error per module type, list of failures); tests/test_fuzz_gpu.py runs
fixed-seed slices in the suite, `python tools/fuzz_features.py SEED N` longer sweeps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mod(name, typ, src=None, **kw):
    t = "module\n{\n  name %s\n  type %s\n" % (name, typ)
    for k, v in kw.items():
        t += "  %s %s\n" % (k.replace("_DASH_", "-"), v)
    if src:
        t += "  sources %s\n" % src
    return t + "}\n"


def run(seed=1, N=40, verbose=False):
    from aaltoasr_amd import capi, synth
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(seed)
    worst = {}
    fails = []
    for it in range(N):
        sr = int(rng.choice([8000, 16000, 16000, 22050, 11025]))
        fr = float(rng.choice([100, 125, 125, 160]))
        # 220, 276, 364, 442: half lengths with prime factors 11, 23, 13, 17 (KissFFT's generic butterfly)
        width = int(rng.choice([0, 0, 256, 320, 400, 512, 200, 240, 220, 276, 364, 442]))
        kw = dict(sample_rate=sr, frame_rate=fr)
        if width:
            kw["window_width"] = width
        if rng.integers(0, 4) == 0:
            kw["copy_borders"] = 0
        if rng.integers(0, 3) == 0:
            kw["pre_emph_coef"] = "%.3f" % rng.uniform(0.9, 0.99)
        cfg = mod("a", "audiofile", **kw)
        cfg += mod("f", "fft", "a", magnitude=int(rng.integers(0, 2)), log=0)
        spec = "f"
        if rng.integers(0, 2):
            opts = [dict(), dict(pwlin_vtln=1), dict(slapt=1), dict(sinc_interpolation_rad=0),
                    dict(lanczos_window=0, sinc_interpolation_rad=int(rng.integers(2, 9)))][int(rng.integers(0, 5))]
            cfg += mod("v", "vtln", "f", **opts)
            spec = "v"
        cfg += mod("m", "mel", spec, root=int(rng.integers(0, 2)))
        cfg += mod("p", "power", spec)
        nd = int(rng.integers(4, 13))
        cfg += mod("c", "dct", "m", dim=nd, zeroth=int(rng.integers(0, 2)))
        cfg += mod("cp", "merge", "c p")
        w1, w2 = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        cfg += mod("d1", "delta", "cp", width=w1)
        cfg += mod("d2", "delta", "d1", width=w2)
        cfg += mod("all", "merge", "cp d1 d2")
        last = "all"
        if rng.integers(0, 2):
            # the production tail: normalization + lin_transform (with delta, delta-delta and the merge
            # in front of it this is the shape the fused temporal kernel takes)
            d3 = 3 * (nd + 1)
            vec = lambda v: " ".join("%.6g" % x for x in np.asarray(v).ravel())
            cfg += mod("nrm", "normalization", last, mean=vec(rng.normal(0, 2, d3)), scale=vec(rng.uniform(0.05, 0.5, d3)))
            lt = dict(dim=d3)
            if rng.integers(0, 3):
                lt["matrix"] = vec(np.eye(d3) + 0.1 * rng.standard_normal((d3, d3)))
            if rng.integers(0, 2):
                lt["bias"] = vec(0.1 * rng.standard_normal(d3))
            cfg += mod("lt", "lin_transform", "nrm", **lt)
            last = "lt"
        if rng.integers(0, 2):
            cfg += mod("cms", "mean_subtractor", last, left=int(rng.integers(0, 40)), right=int(rng.integers(0, 40)))
            last = "cms"
        if rng.integers(0, 2):
            cfg += mod("cc", "concat", last, left=int(rng.integers(0, 3)), right=int(rng.integers(0, 3)))
            last = "cc"
        try:
            ch = O.FeatureChain(cfg)
        except ValueError as e:
            try:
                capi.Feat(cfg)
                fails.append("seed %d it %d: oracle rejects, engine accepts: %s" % (seed, it, e))
            except capi.AasrError:
                pass
            continue
        try:
            ft = capi.Feat(cfg)
        except capi.AasrError as e:
            if e.code == capi.AASR_ERR_UNSUPPORTED or "must be even" in str(e):
                continue   # odd windows fail in the reference's kiss_fftr_alloc as well
            fails.append("seed %d it %d: engine rejects: %s\n%s" % (seed, it, e, cfg))
            continue
        if "v" in ch.by_name:
            prm = {"warp_factor": "%.3f" % rng.uniform(0.85, 1.2)} if not ch.by_name["v"].prm["slapt"] else \
                  {"slapt_coef": "%.4f %.4f" % tuple(rng.uniform(-0.03, 0.03, 2))}
            ch.set_parameters("v", prm)
            ft.set_parameters("v", "{\n" + "".join(" %s %s\n" % kv for kv in prm.items()) + "}\n")
        ww = ch.base.prm["width"]
        n_samples = int(rng.choice([ww + 1, ww + 5, 3 * ww, int(sr * rng.uniform(0.3, 2.5))]))
        pcm = synth.make_audio(n_samples, seed=it, sample_rate=sr)
        if ft.last_frame(n_samples) != ch.last_frame(n_samples) or ft.eof_frame(n_samples) != ch.num_frames(n_samples):
            fails.append("seed %d it %d: last_frame %d vs %d" % (seed, it, ft.last_frame(n_samples), ch.last_frame(n_samples)))
            continue
        lo = int(rng.integers(-20, 3))
        n = max(1, int(ch.last_frame(n_samples) + 1 - lo + rng.integers(0, 15)))
        for m in ch.mods:
            want = ch.generate(pcm, lo, n, module=m.name)
            got = ft.run(pcm, lo, n, module=m.name, dtype=np.float64)
            scale = max(1.0, float(np.abs(want).max()))
            # with logf as glibc computes it on the device the whole chain agrees to summation-order
            # level in double; the mel module with root = 1 goes through pow() (<= 1 ulp of double)
            tol = 1e-12 * scale if m.type in ("audiofile", "fft", "power", "vtln") else 1e-10 * scale
            err = float(np.abs(got - want).max())
            worst[m.type] = max(worst.get(m.type, 0.0), err / scale)
            if not err <= tol:
                fails.append("seed %d it %d module %s (%s) err %g tol %g sr %d fr %g width %d samples %d lo %d n %d" % (
                    seed, it, m.name, m.type, err, tol, sr, fr, ww, n_samples, lo, n))
                break
    if verbose:
        for f in fails:
            print("FAIL", f)
    return worst, fails


if __name__ == "__main__":
    worst, fails = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1,
                       int(sys.argv[2]) if len(sys.argv) > 2 else 40, verbose=True)
    for k in sorted(worst):
        print("%-18s worst relative |err| %.3g" % (k, worst[k]))
    print("failures:", len(fails))
    sys.exit(1 if fails else 0)
