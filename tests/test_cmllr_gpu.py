"""GPU parity of model-side constrained MLLR:
aasr_gmm_set_cmllr against a numpy restatement of AdaptedGaussian
(aku/ModelModules.hh:172-173, 208-212): likelihood = g(A f + b) * |prod diag A|
(the reference's full_matrix_determinant returns the product of A's diagonal,
aku/LinearAlgebra.cc:73-86)."""
import numpy as np
import pytest

from conftest import assert_ll

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_adapted(oracle, model, frames, g2t, W):
    """Per-Gaussian adapted log-likelihoods -> mixtures, in double."""
    mean, var, off, idx, w = model
    om = oracle.DiagModel(mean, var, off, idx, w)
    G = mean.shape[0]
    x = frames.astype(np.float64)
    ll = np.empty((x.shape[0], G))
    base = om.gauss_loglik(x)
    per_t = {}
    for t in range(W.shape[0]):
        A, b = W[t][:, 1:], W[t][:, 0]
        det = abs(np.prod(np.diag(A)))
        with np.errstate(divide="ignore"):
            per_t[t] = om.gauss_loglik(x @ A.T + b) + np.log(det)
    for g in range(G):
        ll[:, g] = base[:, g] if g2t[g] < 0 else per_t[int(g2t[g])][:, g]
    lik = np.exp(ll)
    out = np.empty((x.shape[0], om.S))
    for s in range(om.S):
        a, b_ = om.mix_off[s], om.mix_off[s + 1]
        out[:, s] = np.log(np.maximum(lik[:, om.mix_idx[a:b_]] @ om.mix_w[a:b_], 1e-50))
    return out


def _transforms(n, D, seed):
    rng = np.random.default_rng(seed)
    W = np.empty((n, D, D + 1))
    for t in range(n):
        W[t][:, 1:] = np.eye(D) * rng.uniform(0.8, 1.2, D) + 0.05 * rng.standard_normal((D, D))
        W[t][:, 0] = 0.2 * rng.standard_normal(D)
    return W


@pytest.mark.parametrize("D,G,S,comps", [(13, 64, 8, 8), (39, 128, 16, 8), (80, 96, 12, 8), (130, 60, 10, 6)])
def test_global_transform(capi, oracle, D, G, S, comps):
    model = synth.make_model(D=D, G=G, S=S, comps=comps, seed=D)
    frames = synth.make_frames(150, D=D, seed=3)
    W = _transforms(1, D, seed=1)
    g2t = np.zeros(G, np.int32)
    ref = _oracle_adapted(oracle, model, frames, g2t, W)
    g = capi.Gmm.from_arrays(*model)
    plain = g.score(frames)
    g.set_cmllr(g2t, W)
    got = g.score(frames)
    assert np.abs(got - ref).max() <= 1e-4
    assert np.abs(got - plain).max() > 1e-2       # the transform does something
    g.set_cmllr()                                   # reset_transform
    assert np.array_equal(g.score(frames), plain)
    if D > 63:   # the model as dimension parts: regression classes become class sub-models of the same kind, and back
        g2 = (np.arange(G) % 3 - 1).astype(np.int32)
        W2 = _transforms(2, D, seed=2)
        g.set_cmllr(g2, W2)
        assert_ll(g.score(frames), _oracle_adapted(oracle, model, frames, g2, W2), "regression classes, D = %d" % D)
        g.set_cmllr(g2t, W)
        assert_ll(g.score(frames), ref, "back to one transform, D = %d" % D)
        g.set_cmllr()
        assert np.array_equal(g.score(frames), plain)


@pytest.mark.parametrize("D,G,S,comps", [(13, 64, 8, 8), (39, 96, 12, 8), (80, 96, 12, 8)])
def test_regression_classes_and_unadapted_gaussians(capi, oracle, D, G, S, comps):
    """Three regression classes assigned per Gaussian (UNIT_GAUSSIAN style), some
    Gaussians left unadapted: components of one mixture use different transforms."""
    model = synth.make_model(D=D, G=G, S=S, comps=comps, seed=7)
    frames = synth.make_frames(130, D=D, seed=4)
    W = _transforms(3, D, seed=2)
    rng = np.random.default_rng(5)
    g2t = rng.integers(-1, 3, G).astype(np.int32)
    ref = _oracle_adapted(oracle, model, frames, g2t, W)
    g = capi.Gmm.from_arrays(*model)
    g.set_cmllr(g2t, W)
    got = g.score(frames)
    assert_ll(got, ref, "regression classes")
    g.set_precision(3)      # the per-class factor rows on the bf16 pipe (k_gmm_full_score_bf16x3)
    assert_ll(g.score(frames), ref, "regression classes, three-term rows")


def test_zero_diagonal_kills_the_adapted_gaussians(capi, oracle):
    """prod diag A == 0 -> likelihood 0 for the adapted Gaussians."""
    D, G = 8, 16
    model = synth.make_model(D=D, G=G, S=2, comps=8, seed=9)
    frames = synth.make_frames(40, D=D, seed=8)
    W = _transforms(2, D, seed=3)
    W[1][2, 3] = 0.0
    W[1][2, 1 + 2] = 0.0
    g2t = np.array([0] * 8 + [1] * 8, np.int32)
    ref = _oracle_adapted(oracle, model, frames, g2t, W)
    g = capi.Gmm.from_arrays(*model)
    g.set_cmllr(g2t, W)
    got = g.score(frames)
    assert np.abs(got - ref).max() <= 1e-4
    assert np.allclose(got[:, 1], np.log(1e-50), atol=1e-5)


def test_argument_validation(capi):
    model = synth.make_model(D=8, G=16, S=2, comps=8)
    g = capi.Gmm.from_arrays(*model)
    with pytest.raises(capi.AasrError, match="out of range"):
        g.set_cmllr(np.full(16, 4, np.int32), _transforms(2, 8, 0))


def test_class_routing_across_speaker_changes(capi, oracle):
    """Per-class transforms go through per-class sub-models that only depend on the class
    membership: new matrices for the same membership, a new membership, an empty class, a state
    held by one class only, tied Gaussians, ragged and empty states, then back to global / none."""
    rng = np.random.default_rng(17)
    D, G = 13, 96
    mean, var, _, _, _ = synth.make_model(D=D, G=G, S=4, comps=4, seed=5)
    n = np.array([3, 0, 9, 1, 16, 5, 2])
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
    idx = rng.integers(0, G, off[-1]).astype(np.int32)          # tied pool
    w = rng.uniform(0.05, 1.0, off[-1])
    model = (mean, var, off, idx, w)
    frames = synth.make_frames(150, D=D, seed=3)
    g = capi.Gmm.from_arrays(*model)
    g2t = rng.integers(-1, 3, G).astype(np.int32)
    g2t[idx[off[2]:off[3]]] = 1                                  # state 2 entirely in class 1
    for seed in (1, 2):                                          # same membership, new speaker
        W = _transforms(4, D, seed)                              # class 3 has no Gaussian at all
        g.set_cmllr(g2t, W)
        ref = _oracle_adapted(oracle, model, frames, g2t, W)
        for prec in (4, 3, 0):
            g.set_precision(prec)
            assert np.abs(g.score(frames) - ref).max() <= 1e-4
    g2t2 = rng.integers(-1, 2, G).astype(np.int32)               # new membership
    W2 = _transforms(2, D, 9)
    g.set_cmllr(g2t2, W2)
    assert np.abs(g.score(frames) - _oracle_adapted(oracle, model, frames, g2t2, W2)).max() <= 1e-4
    g.set_cmllr(np.zeros(G, np.int32), W2[:1])                   # one transform for all: frame transform path
    assert np.abs(g.score(frames) - _oracle_adapted(oracle, model, frames, np.zeros(G, np.int32), W2[:1])).max() <= 1e-4
    g.set_cmllr()                                                # unadapted again
    plain = oracle.DiagModel(*model).score(frames.astype(np.float64))
    assert np.abs(g.score(frames) - plain).max() <= 1e-4


def test_global_transform_in_place_on_every_kernel(capi, oracle):
    """A global transform is applied without re-packing the rows: log|det| enters at the kernels'
    output (the track kernels' reference exponent; an extra pass for the general and the centred
    kernel), a zero diagonal puts every state at the floor, repeated changes do not accumulate."""
    D, G = 13, 64
    model = synth.make_model(D=D, G=G, S=8, comps=8, seed=11)
    frames = synth.make_frames(90, D=D, seed=2)
    g = capi.Gmm.from_arrays(*model)
    g2t = np.zeros(G, np.int32)
    for seed in (1, 2, 3):
        W = _transforms(1, D, seed)
        g.set_cmllr(g2t, W)
        ref = _oracle_adapted(oracle, model, frames, g2t, W)
        for prec, mask in ((3, 7), (0, 7), (0, 2), (0, 0), (0, 4)):
            g.set_precision(prec)
            g.set_layouts(mask)
            assert np.abs(g.score(frames) - ref).max() <= 1e-4, (seed, prec, mask)
        g.set_layouts(7)
    W0 = _transforms(1, D, 4)
    W0[0, 3, 4] = 0.0                                  # A[3][3] = 0 -> |det| = 0
    g.set_cmllr(g2t, W0)
    assert np.allclose(g.score(frames), np.log(1e-50), atol=1e-5)
    g.set_cmllr()
    assert np.abs(g.score(frames) - oracle.DiagModel(*model).score(frames.astype(np.float64))).max() <= 1e-4


def test_routed_scoring_in_many_passes(capi, oracle):
    """Outlier routing and class routing keep partial scores in a bounded scratch and walk the
    frames in passes; with the budget shrunk to a few KB the result must not change by a bit."""
    import ctypes as C
    L = capi.lib()
    L.aasr_debug_set_pass_bytes.argtypes = [C.c_double]
    L.aasr_debug_set_pass_bytes.restype = None
    rng = np.random.default_rng(41)
    D, G = 13, 160
    mean, var, off, idx, w = synth.make_model(D=D, G=G, S=20, comps=8, seed=19)
    var[rng.choice(G, 8, replace=False)] *= 1e-3                 # outliers
    frames = synth.make_frames(2100, D=D, seed=7)
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    try:
        one = g.score(frames)
        L.aasr_debug_set_pass_bytes(20000.0)
        many = g.score(frames)
        assert np.array_equal(one.view(np.uint32), many.view(np.uint32))
        L.aasr_debug_set_pass_bytes(0.0)
        g2t = rng.integers(-1, 3, G).astype(np.int32)
        W = _transforms(3, D, 6)
        g.set_cmllr(g2t, W)
        one = g.score(frames)
        L.aasr_debug_set_pass_bytes(30000.0)
        many = g.score(frames)
        assert np.array_equal(one.view(np.uint32), many.view(np.uint32))
        ref = _oracle_adapted(oracle, (mean, var, off, idx, w), frames[:200], g2t, W)
        assert_ll(many[:200], ref, "per-class transforms over outlier-routed Gaussians")
    finally:
        L.aasr_debug_set_pass_bytes(0.0)
