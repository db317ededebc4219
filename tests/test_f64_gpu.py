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

from conftest import assert_ll

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
    assert_ll(g.score(f32), want32, "default arithmetic on the same frames")
    # one global CMLLR transform: adapted frames and |prod diag A| in double as well
    A = np.eye(D) * rng.uniform(0.9, 1.1, D) + 0.02 * rng.standard_normal((D, D))
    W = np.hstack([0.1 * rng.standard_normal(D)[:, None], A])
    g.set_cmllr(np.zeros(G, np.int32), W[None])
    want_a = oracle.score_adapted(om, frames, np.zeros(G, np.int32), W[None])
    got_a = g.score_f64(frames)
    assert np.abs(got_a - want_a).max() <= 1e-10 * max(1.0, np.abs(want_a).max())
    # per-class transforms (regression classes): every component evaluates its class's A f + b, scaled by its
    # |prod diag A|, summed in component order; Gaussians without a transform stay plain
    A2 = np.eye(D) * rng.uniform(0.8, 1.2, D) + 0.03 * rng.standard_normal((D, D))
    W2 = np.hstack([0.2 * rng.standard_normal(D)[:, None], A2])
    g2t = (np.arange(G) % 3).astype(np.int32) - 1        # -1: unadapted
    g.set_cmllr(g2t, np.stack([W, W2]))
    want_c = oracle.score_adapted(om, frames, g2t, np.stack([W, W2]))
    got_c = g.score_f64(frames)
    assert np.abs(got_c - want_c).max() <= 1e-10 * max(1.0, np.abs(want_c).max())
    assert np.abs(got_c - want_a).max() > 1e-3           # and it is a different model
    # the float entry point under the mode rounds those values once
    g.set_precision(1)
    assert np.abs(g.score(frames.astype(np.float32)) - oracle.score_adapted(
        om, frames.astype(np.float32).astype(np.float64), g2t, np.stack([W, W2]))).max() <= 2e-5
    g.set_precision(0)
    # ... and together with Gaussian clustering: the selection on the raw frames (plain centres), the members of
    # selected clusters on their class's frames
    if G >= 16:
        pairs = [(i, i % 4) for i in range(G) if i % 7]
        om.set_clustering(4, pairs, 0.0, 0.25)
        g.set_clustering(4, pairs)
        g.set_clustering_min_evals(0.0, 0.25)
        want_cc, n_cc = om.score_clustered_classes(frames, g2t, np.stack([W, W2]), want_counts=True)
        got_cc = g.score_f64(frames)
        assert np.array_equal(g.cluster_exact_counts(len(frames)), n_cc)
        assert np.abs(got_cc - want_cc).max() <= 1e-10 * max(1.0, np.abs(want_cc).max())


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


def test_f64_with_gaussian_clustering_is_the_oracles_cluster_branch(capi, oracle):
    """The production configuration (-C ... --eval-ming 0.25) in AASR_PREC_F64: centres on the double
    frames, the same selection, every component either exact or its centre's likelihood
    (aku/Distributions.cc:2684-2722) -- state log-likelihoods to 1e-12, exact-evaluation counts equal,
    the LNA file of an utterance byte-identical to the oracle's clustered phone_probs; also under one
    global CMLLR transform (adapted members, plain centres)."""
    rng = np.random.default_rng(9)
    mean, var, off, idx, w = synth.make_model(D=39, G=1152, S=96, comps=12, seed=8)
    g2c = synth.make_clustering(mean, 40)
    g2c[3] = g2c[500] = -1
    pairs = [(int(a), int(c)) for a, c in enumerate(g2c) if c >= 0]
    frames = synth.make_frames(300, seed=6).astype(np.float64)
    frames[:5] *= 40.0                                   # far frames: centres underflow, members go exact
    om = oracle.DiagModel(mean, var, off, idx, w)
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    g.set_clustering(40, pairs)
    for minc, ming in ((0.0, 0.25), (0.2, 0.0), (0.0, 0.0)):
        om.set_clustering(40, pairs, minc, ming)
        g.set_clustering_min_evals(minc, ming)
        want, want_n = om.score_clustered(frames, want_counts=True)
        got = g.score_f64(frames)
        assert np.array_equal(g.cluster_exact_counts(len(frames)), want_n)
        assert np.abs(got - want).max() <= 1e-10 * max(1.0, np.abs(want).max()), (minc, ming)
    A = np.eye(39) * rng.uniform(0.9, 1.1, 39) + 0.02 * rng.standard_normal((39, 39))
    W = np.hstack([0.1 * rng.standard_normal(39)[:, None], A])
    g.set_cmllr(np.zeros(1152, np.int32), W[None])
    want_a, n_a = om.score_clustered_adapted(frames, W, want_counts=True)
    got_a = g.score_f64(frames)
    assert np.array_equal(g.cluster_exact_counts(len(frames)), n_a)
    assert np.abs(got_a - want_a).max() <= 1e-10 * max(1.0, np.abs(want_a).max())
    g.set_cmllr()
    # an utterance through the whole path
    cfg = open(os.path.join(GOLDEN, "mfcc_cms_norm.feaconf")).read()
    pcm = synth.make_audio(16000 * 3, seed=78)
    ft = capi.Feat(cfg)
    g.set_precision(1)
    data, n = capi.run_utterance(ft, g, pcm, lnabytes=2)
    fea = oracle.FeatureChain(cfg).generate(pcm, 0, n)
    ll = om.score_clustered(fea)
    _, by_ref = oracle.lna_encode(np.exp(ll), True, 2)
    got = np.frombuffer(data[5:], np.uint8).reshape(n, 96, 2)
    same = (got == by_ref.reshape(n, 96, 2)).all(axis=2).mean()
    assert same >= 0.9999, same
