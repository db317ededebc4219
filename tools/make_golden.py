"""Writes tests/golden/oracle_vectors.npz: outputs of the CPU oracle on seeded
inputs for the parts of the path the reference holds no golden vectors for
(GMM scoring, clustering, CMLLR, LNA packing, full-precision features).  The
inputs are regenerated from the seeds by the tests; only the expected outputs
are stored.  Re-run after an intended oracle change:  python tools/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aaltoasr_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def vectors():
    O.build()
    out = {}
    # (1) config[0]-shaped model: D=39, G=256, S=32 x 8, 200 frames incl. far ones
    model = synth.make_model(D=39, G=256, S=32, comps=8)
    frames = synth.make_frames(200)
    frames[::25] *= 6.0                      # float-underflow / floor paths
    om = O.DiagModel(*model)
    ll, lik = om.score(frames.astype(np.float64), want_lik=True)
    out["cfg0_state_loglik"] = ll
    for nb in (2, 4):
        lp, by = O.lna_encode(lik, True, nb)
        out["cfg0_lna%d_lp" % nb] = lp
        out["cfg0_lna%d_bytes" % nb] = by
    out["cfg0_gauss_loglik_f1_4"] = om.gauss_loglik(frames[1:4].astype(np.float64))
    # (2) tied pool, ragged mixtures
    model = synth.make_model(D=39, G=2048, S=128, tied=True, comps_range=(1, 23))
    frames = synth.make_frames(120, seed=77)
    out["tied_state_loglik"] = O.DiagModel(*model).score(frames.astype(np.float64))
    # (3) clustering
    model = synth.make_model(D=39, G=2048, S=128, comps=16)
    g2c = synth.make_clustering(model[0], 64)
    om = O.DiagModel(*model)
    om.set_clustering(64, [(g, int(c)) for g, c in enumerate(g2c)], 0.0, 0.25)
    s, n = om.score_clustered(synth.make_frames(150, seed=78).astype(np.float64), want_counts=True)
    out["cluster_state_loglik"], out["cluster_exact_counts"] = s, n
    out["cluster_g2c"] = g2c
    # (4) full covariance
    rng = np.random.default_rng(synth.SEED + 5)
    D, G = 8, 64
    mean = rng.standard_normal((G, D))
    a = rng.standard_normal((G, D, D)) * 0.4
    cov = a @ a.transpose(0, 2, 1) + 0.1 * np.eye(D)
    _, _, off, idx, w = synth.make_model(D=D, G=G, S=8, comps=8)
    out["full_state_loglik"] = O.FullModel(mean, cov, off, idx, w).score(synth.make_frames(100, D=D, seed=79))
    # (5) features of the reference's own test audio at full precision
    pcm, _ = O.read_wav_pcm16(os.path.join(ROOT, "tests", "golden", "short.wav"))
    for name in ("mfcc_p_dd", "mfcc_cms_norm"):
        ch = O.FeatureChain(open(os.path.join(ROOT, "tests", "golden", name + ".feaconf")).read())
        out["fea_" + name] = ch.generate(pcm, -15, 106)
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz")
    np.savez_compressed(path, **vectors())
    print("wrote", path, os.path.getsize(path), "bytes")
