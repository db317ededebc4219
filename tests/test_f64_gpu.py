"""AASR_PREC_F64: the reference's own arithmetic in double on the device
(DiagonalGaussian::compute_log_likelihood aku/Distributions.cc:1040-1062, Mixture::compute_likelihood
:2078-2086, HmmSet's 1e-50 clamp, the phone_probs tail aku/phone_probs.cc:224-262).  Against the
oracle only the device's exp / log (<= 1 ulp) and the order of the normaliser's sum can differ: state
log-likelihoods agree to ~1e-14 and, from the same features, the LNA files are byte-identical -- which
shows that the 0.5 % of 2-byte codes that differ in the default arithmetic are float rounding and
nothing else."""
import os

import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("D,G,S,comps,tied", [(39, 256, 32, 8, False), (13, 64, 16, 4, True), (1, 15, 5, 3, False),
                                              (63, 40, 8, 5, False), (24, 300, 50, 9, True)])
def test_f64_scores_equal_the_oracle(capi, oracle, D, G, S, comps, tied):
    rng = np.random.default_rng(D + G)
    mean, var, off, idx, w = synth.make_model(D=D, G=G, S=S, comps=comps, seed=D, tied=tied)
    var[1, :] *= 1e-3                       # a variance-floored Gaussian: no conditioning issue in this form
    var[2, D // 2] = 0.0                    # an "invalid" Gaussian (precision 0, constant 0)
    w[off[1]] = 0.0                         # a zero weight
    frames = (rng.standard_normal((257, D)) * rng.uniform(0.5, 3.0)).astype(np.float64)
    frames[:3] = mean[:3] + 1e-3
    om = oracle.DiagModel(mean, var, off, idx, w)
    want = om.score(frames)
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    got = g.score_f64(frames)
    assert np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max())
    assert np.all(got[want == np.log(1e-50)] == np.log(1e-50))
    # float entry points under AASR_PREC_F64: one rounding of the frames, one of the result
    g.set_precision(1)
    f32 = frames.astype(np.float32)
    got32 = g.score(f32)
    want32 = om.score(f32.astype(np.float64))
    assert np.abs(got32 - want32).max() <= 1.5e-7 * max(1.0, np.abs(want32).max())
    try:
        g.set_precision(3)
    except capi.AasrError:
        g.set_precision(0)
    assert np.abs(g.score(f32) - want32).max() <= 2e-4
    # one global CMLLR transform: adapted frames and |prod diag A| in double as well
    A = np.eye(D) * rng.uniform(0.9, 1.1, D) + 0.02 * rng.standard_normal((D, D))
    W = np.hstack([0.1 * rng.standard_normal(D)[:, None], A])
    g.set_cmllr(np.zeros(G, np.int32), W[None])
    want_a = oracle.score_adapted(om, frames, np.zeros(G, np.int32), W[None])
    got_a = g.score_f64(frames)
    assert np.abs(got_a - want_a).max() <= 1e-10 * max(1.0, np.abs(want_a).max())
    # refused where it is not built: per-class transforms
    g2t = (np.arange(G) % 2).astype(np.int32)
    g.set_cmllr(g2t, np.stack([W, W]))
    with pytest.raises(capi.AasrError, match="AASR_PREC_F64 is built for diagonal pools without"):
        g.score_f64(frames)


@pytest.mark.parametrize("nbytes,normalize", [(2, True), (4, True), (4, False)])
def test_f64_lna_files_are_the_oracles(capi, oracle, nbytes, normalize):
    cfg = open(os.path.join(GOLDEN, "mfcc_cms_norm.feaconf")).read()
    model = synth.make_model(D=39, G=512, S=64, comps=8, seed=5)
    pcm = synth.make_audio(16000 * 4, seed=77)
    ft = capi.Feat(cfg)
    g = capi.Gmm.from_arrays(*model)
    g.set_precision(1)
    data, n = capi.run_utterance(ft, g, pcm, lnabytes=nbytes, normalize=normalize)
    # (1) scoring + normalisation + packing: from the SAME features, the oracle's file byte for byte
    efea = ft.run(pcm, 0, n, dtype=np.float64)
    om = oracle.DiagModel(*model)
    _, lik = om.score(efea, want_lik=True)
    lp_e, by_e = oracle.lna_encode(lik, normalize, nbytes)
    got = np.frombuffer(data[5:], np.uint8).reshape(n, 64, nbytes)
    assert np.array_equal(got, by_e.reshape(n, 64, nbytes))
    # (2) end to end against the oracle's own feature chain: the features agree to ~1e-14 (logf is glibc's
    # algorithm on the device), which neither a 2-byte code nor a float log-probability sees
    fea = oracle.FeatureChain(cfg).generate(pcm, 0, n)
    assert np.abs(efea - fea).max() <= 1e-10
    _, lik = om.score(fea, want_lik=True)
    lp_ref, by_ref = oracle.lna_encode(lik, normalize, nbytes)
    vals = (got == by_ref.reshape(n, 64, nbytes)).all(axis=2).mean()
    print("f64 LNA vs the oracle's chain: identical values %.6f" % vals)
    assert vals >= 0.9999
    if nbytes == 4:
        lp = np.frombuffer(data[5:], "<f4").reshape(n, 64)
        assert np.abs(lp - lp_ref).max() <= 2e-6
    # the default arithmetic on the same input, for the contrast the docstring draws
    g.set_precision(3)
    data3, _ = capi.run_utterance(ft, g, pcm, lnabytes=nbytes, normalize=normalize)
    got3 = np.frombuffer(data3[5:], np.uint8).reshape(n, 64, nbytes)
    vals3 = (got3 == by_ref.reshape(n, 64, nbytes)).all(axis=2).mean()
    assert vals3 < vals or vals == 1.0
