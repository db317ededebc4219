"""Gaussian clustering: the product's clustered scoring vs
the oracle's restatement of PDFPool::precompute_likelihoods' cluster branch
(aku/Distributions.cc:2684-2722) on the same frames.  Tolerance 1e-4 on the
state log-likelihoods (BASELINE north_star); the per-frame number of clusters
evaluated exactly is an integer and must match."""
import os

import numpy as np
import pytest

from conftest import CODES_EQUAL_MIN, assert_ll

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _pairs(g2c):
    return [(int(g), int(c)) for g, c in enumerate(g2c) if c >= 0]


def _check(capi, oracle, model, g2c, C, minc, ming, frames, pairs=None, tol=TOL, counts=True):
    mean, var, off, idx, w = model
    pairs = _pairs(g2c) if pairs is None else pairs
    om = oracle.DiagModel(mean, var, off, idx, w)
    om.set_clustering(C, pairs, minc, ming)
    want, want_n = om.score_clustered(frames.astype(np.float64), want_counts=True)
    gm = capi.Gmm.from_arrays(mean, var, off, idx, w)
    gm.set_clustering(C, pairs)
    gm.set_clustering_min_evals(minc, ming)
    # every kernel that carries the selection masks: grouped / independent track
    # layouts, f32 and bf16x3 contraction
    for layouts in (7, 2):
        gm.set_layouts(layouts)
        for prec in (0, 3, 4):
            gm.set_precision(prec)
            got = gm.score(frames)
            got_n = gm.cluster_exact_counts(len(frames))
            if counts:
                assert np.array_equal(got_n, want_n), (got_n[:10], want_n[:10])
            err = np.abs(got - want).max()
            assert err <= tol, (layouts, prec, err)
    gm.set_layouts(7)
    gm.set_precision(0)
    got = gm.score(frames)
    return gm, om, got, want


@pytest.mark.parametrize("minc,ming", [(0.0, 0.1), (0.0, 0.25), (0.3, 0.0), (0.2, 0.5), (0.0, 0.0),
                                       (1.0, 1.0)])
def test_clustered_scores_match_oracle(capi, oracle, minc, ming):
    model = synth.make_model(D=39, G=2048, S=128, comps=16)
    g2c = synth.make_clustering(model[0], 64)
    frames = synth.make_frames(500)
    gm, om, got, want = _check(capi, oracle, model, g2c, 64, minc, ming, frames)
    if minc == 1.0:
        # every cluster evaluated exactly == no clustering at all
        assert np.abs(got - om.score(frames.astype(np.float64))).max() <= TOL
    if minc == 0.0 and ming == 0.0:
        # nothing exact: the approximation must actually differ from exact scoring
        assert np.abs(want - om.score(frames.astype(np.float64))).max() > 0.1


def test_tied_pool_ragged_states_unclustered_gaussians(capi, oracle):
    model = synth.make_model(D=24, G=700, S=90, tied=True, comps_range=(1, 23))
    g2c = synth.make_clustering(model[0], 40)
    g2c[::17] = -1                      # Gaussians in no cluster: always exact
    _check(capi, oracle, model, g2c, 40, 0.1, 0.2, synth.make_frames(300, D=24))


def test_clustered_rows_padded_to_whole_lines_equal_the_dense_rows(capi):
    """aasr_gmm_score_dev_pitched under clustering (the plain plan: masked track kernel + merge carry the row pitch):
    bit for bit the dense call's values, the padding untouched by the merge."""
    import torch
    model = synth.make_model(D=39, G=1500, S=125, comps=12)
    g2c = synth.make_clustering(model[0], 50)
    gm = capi.Gmm.from_arrays(*model)
    gm.set_clustering(50, _pairs(g2c))
    gm.set_clustering_min_evals(0.0, 0.25)
    d_fr = torch.from_numpy(synth.make_frames(1500)).cuda()
    for prec in (0, 3, 4):
        gm.set_precision(prec)
        assert gm.score_pitch_ok()
        dense = torch.empty((1500, 125), device="cuda")
        gm.score_dev(d_fr, dense)
        padded = torch.full((1500, 128), 7.0, device="cuda")
        gm.score_dev_pitched(d_fr, padded, 128)
        torch.cuda.synchronize()
        assert torch.equal(padded[:, :125], dense), prec
    gm.set_precision(1)   # AASR_PREC_F64 writes dense rows
    assert not gm.score_pitch_ok()


def test_far_frames_underflowed_centres_fall_back_to_exact(capi, oracle):
    """Centre likelihood 0.0 in double -> PDFPool::compute_likelihood re-evaluates
    (aku/Distributions.cc:2636-2644)."""
    model = synth.make_model(D=39, G=1024, S=64, comps=16)
    g2c = synth.make_clustering(model[0], 32)
    frames = synth.make_frames(256)
    frames[::3] *= 9.0                  # centre ll below -745 for most clusters
    frames[1::3] *= 4.0
    # Underflowed centres tie at likelihood 0.0 and the reference pops ties in the order of
    # its std::priority_queue: frames whose stopping point falls among the zeros are replayed
    # on the device (k_cluster_select_heap), so the number of clusters popped matches too.
    gm, om, got, want = _check(capi, oracle, model, g2c, 32, 0.0, 0.1, frames, counts=True)
    zero_centres = np.exp(-0.5 * ((frames[:, None, :].astype(np.float64) - om.c_mean[None]) ** 2
                                  * om.c_prec[None]).sum(-1) + om.c_cst[None]) == 0.0
    assert zero_centres[::3].mean() > 0.5 and not zero_centres[2::3].any()
    assert gm.cluster_tie_frames() > 0          # the replay really ran


def test_empty_clusters_tie_and_the_queue_order_decides(capi, oracle):
    """The round-1 fuzz case (tools/fuzz_parity.py seed 1, iteration 26): D = 2, 66 random clusters
    of which some are empty, --eval-minc 0.1 --eval-ming 0.  An empty cluster is an "invalid"
    centre with log-likelihood 0 (aku/Distributions.cc:1276-1287), so several clusters tie at
    likelihood 1.0 and the reference's priority queue (aku/Distributions.hh:291-299,
    aku/Distributions.cc:2684-2722) pops only as many of them as --eval-minc still needs.  The
    per-frame number of clusters evaluated exactly must equal the oracle's replay of libstdc++'s
    heap, for every kernel that carries the masks."""
    rng = np.random.default_rng(126)
    D, S, G, C = 2, 38, 244, 66
    n = rng.integers(1, 12, S)
    n[-1] += G - n.sum() if n.sum() < G else 0
    K = int(n.sum())
    off = np.zeros(S + 1, np.int32)
    off[1:] = np.cumsum(n)
    idx = rng.integers(0, G, K).astype(np.int32)
    w = rng.uniform(0.01, 1.0, K)
    mean = rng.standard_normal((G, D)) * 1.3
    var = np.exp(rng.uniform(np.log(0.2), np.log(5.0), (G, D)))
    g2c = rng.integers(0, C, G)
    empty = [3, 17, 18, 40, 65, 0, 33]
    g2c[np.isin(g2c, empty)] = 5
    g2c[rng.integers(0, G, 20)] = -1
    frames = (rng.standard_normal((300, D)) * 1.7).astype(np.float32)
    for minc, ming in ((0.1, 0.0), (0.05, 0.0), (0.1, 0.1), (0.5, 0.3)):
        gm, om, got, want = _check(capi, oracle, (mean, var, off, idx, w), g2c, C, minc, ming, frames)
    # the tie is real: some frames stop inside the group of empty clusters
    gm.set_clustering_min_evals(0.1, 0.0)
    gm.score(frames)
    assert 0 < gm.cluster_tie_frames() <= 300


def test_histogram_selection_equals_the_queue_replay(capi):
    """k_cluster_select's histogram selection against the step-by-step replay of the reference's
    priority queue on the same frames (every frame forced through k_cluster_select_heap): scores
    bit for bit, counts equal."""
    model = synth.make_model(D=39, G=4096, S=256, comps=16)
    g2c = synth.make_clustering(model[0], 200)
    frames = synth.make_frames(2000)
    frames[::7] *= 3.0
    gm = capi.Gmm.from_arrays(*model)
    gm.set_clustering(200, _pairs(g2c))
    for minc, ming in ((0.0, 0.25), (0.1, 0.0), (0.3, 0.5)):
        gm.set_clustering_min_evals(minc, ming)
        fast = gm.score(frames)
        fast_n = gm.cluster_exact_counts(len(frames))
        capi.debug_cluster_heap(True)
        try:
            slow = gm.score(frames)
            slow_n = gm.cluster_exact_counts(len(frames))
            assert gm.cluster_tie_frames() == len(frames)
        finally:
            capi.debug_cluster_heap(False)
        assert np.array_equal(fast_n, slow_n)
        assert np.array_equal(fast.view(np.uint32), slow.view(np.uint32))


def test_gcl_file_counts_last_pair_twice(capi, oracle, tmp_path):
    model = synth.make_model(D=20, G=600, S=60, comps=10)
    mean, var, off, idx, w = model
    g2c = synth.make_clustering(mean, 30)
    path = str(tmp_path / "m.gcl")
    oracle.write_gcl(path, 30, g2c)
    n, pairs = oracle.read_gcl(path, 600)
    assert n == 30 and len(pairs) == 601 and pairs[-1] == pairs[-2]
    frames = synth.make_frames(200, D=20)
    om = oracle.DiagModel(mean, var, off, idx, w)
    om.set_clustering(n, pairs, 0.0, 0.15)
    want, want_n = om.score_clustered(frames.astype(np.float64), want_counts=True)
    gm = capi.Gmm.from_arrays(mean, var, off, idx, w)
    gm.read_clustering(path)
    assert gm.num_clusters == 30
    gm.set_clustering_min_evals(0.0, 0.15)
    got = gm.score(frames)
    assert np.array_equal(gm.cluster_exact_counts(200), want_n)
    assert np.abs(got - want).max() <= TOL
    # without the repeated pair exactly one centre (the last Gaussian's) is different
    dup = om.c_mean.copy()
    om.set_clustering(n, pairs[:-1], 0.0, 0.15)
    changed = np.flatnonzero((dup != om.c_mean).any(1))
    assert list(changed) == [pairs[-1][1]]


def test_many_clusters_multi_pass(capi, oracle):
    """> 1024 clusters (32 keys per lane in k_cluster_select)."""
    model = synth.make_model(D=13, G=6000, S=300, comps=20)
    g2c = synth.make_clustering(model[0], 1500, iters=2)
    _check(capi, oracle, model, g2c, 1500, 0.05, 0.2, synth.make_frames(130, D=13))


def test_several_passes_of_equal_size_give_the_single_pass_result(capi, oracle, monkeypatch):
    """A clustered run takes as many frames per pass as its scratch budget allows (24 GB) and cuts what is left into passes
    of equal size (gmm_cluster_score_launch).  With the budget cut down (a diagnostic entry point) 3 000 frames take five
    and then two passes: the scores are those of the single pass, bit for bit -- on a plain model
    and on a model with engine parts (the merge writes the public layout from the parts' columns pass by pass)."""
    import ctypes
    from test_pivot_groups_gpu import blobs
    L = capi.lib()
    L.aasr_debug_cluster_pass_bytes.argtypes = [ctypes.c_double]
    L.aasr_debug_cluster_pass_bytes.restype = None
    plain = synth.make_model(D=39, G=2048, S=128, comps=16)
    Xb = blobs(30000)
    fitted = synth.fit_model(Xb, S=200, comps=16)
    for name, model, frames in (("plain", plain, synth.make_frames(3000, D=39)), ("parts", fitted, np.ascontiguousarray(Xb[:3000]))):
        C = 40
        g2c = synth.make_clustering(model[0], C, iters=2)
        if name == "parts":
            monkeypatch.setenv("AASR_PG_PIVOT_COST", "64")
        gm = capi.Gmm.from_arrays(*model)
        assert (gm.engine_parts() is not None) == (name == "parts")
        gm.set_clustering(C, _pairs(g2c))
        gm.set_clustering_min_evals(0.0, 0.25)
        outs, passes = [], []
        try:
            # (bytes per frame of a pass: ~430 for the plain model, ~2 400 with the parts' engine rows)
            for budget in ((0.0, 3.2e5, 8.0e5) if name == "plain" else (0.0, 1.7e6, 4.5e6)):
                L.aasr_debug_cluster_pass_bytes(budget)
                outs.append(gm.score(frames))
                passes.append(L.aasr_debug_cluster_last_passes())
        finally:
            L.aasr_debug_cluster_pass_bytes(0.0)
        assert passes[0] == 1 and passes[1] >= 4 and 2 <= passes[2] < passes[1], (name, passes)
        for sc in outs[1:]:
            assert np.array_equal(sc, outs[0]), name
        gm.close()


def test_more_clusters_than_a_selection_wave_holds(capi, oracle):
    """> 4096 clusters: every frame takes the replay of the reference's priority queue (k_cluster_select_heap) and the
    log-domain merge; scores and counts as the oracle's."""
    rng = np.random.default_rng(41)
    model = synth.make_model(D=6, G=15000, S=500, comps=30)
    C = 4300
    g2c = rng.integers(0, C, 15000)
    g2c[rng.integers(0, 15000, 40)] = -1
    frames = synth.make_frames(70, D=6)
    _check(capi, oracle, model, g2c, C, 0.02, 0.1, frames)


def test_centres_within_a_float_ulp_of_each_other_are_ranked_by_their_doubles(capi, oracle):
    """The selection ranks float keys (half the bytes and registers of the doubles); two centres whose log-likelihoods
    round to the same float are told apart by k_cluster_select_pending on doubles, as the reference's double-precision
    queue does -- here every cluster has a twin 1e-7 away and the minimum cluster count is odd, so the stopping point
    keeps falling between twins.  Counts and scores as the oracle's; nothing is left to the queue replay (the twins are
    not equal in double)."""
    rng = np.random.default_rng(77)
    D, C2, per = 12, 30, 8                       # 30 clusters + their 30 twins, 8 Gaussians each
    G = 2 * C2 * per
    base_mean = rng.standard_normal((C2 * per, D)) * 1.5
    base_var = np.exp(rng.uniform(np.log(0.3), np.log(2.0), (C2 * per, D)))
    mean = np.concatenate([base_mean, base_mean * (1.0 + 1e-7) + 1e-7])
    var = np.concatenate([base_var, base_var])
    g2c = np.concatenate([np.repeat(np.arange(C2), per), C2 + np.repeat(np.arange(C2), per)])
    S = 48
    n = np.full(S, G // S)
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
    idx = rng.permutation(G).astype(np.int32)
    w = rng.uniform(0.1, 1.0, G)
    frames = (rng.standard_normal((400, D)) * 1.2).astype(np.float32)
    for minc, ming in ((0.15, 0.0), (0.05, 0.0), (0.0, 0.113)):
        gm, om, got, want = _check(capi, oracle, (mean, var, off, idx, w), g2c, 2 * C2, minc, ming, frames)
        assert gm.cluster_tie_frames() == 0
    # the float keys of twins do coincide for most frames: the second kernel is what kept the counts right
    ll = (-0.5 * ((frames[:, None, :].astype(np.float64) - om.c_mean[None]) ** 2 * om.c_prec[None]).sum(-1) + om.c_cst[None])
    same = (ll[:, :C2].astype(np.float32) == ll[:, C2:].astype(np.float32)) & (ll[:, :C2] != ll[:, C2:])
    assert same.mean() > 0.3


def test_clustering_errors(capi, tmp_path):
    mean, var, off, idx, w = synth.make_model(D=8, G=100, S=10, comps=10)
    gm = capi.Gmm.from_arrays(mean, var, off, idx, w)
    with pytest.raises(capi.AasrError, match="insensible"):
        gm.set_clustering(31, [(0, 0)])
    with pytest.raises(capi.AasrError, match="Gauss index out of bounds"):
        gm.set_clustering(10, [(100, 0)])
    with pytest.raises(capi.AasrError, match="Cluster index out of bounds"):
        gm.set_clustering(10, [(1, 10)])
    with pytest.raises(capi.AasrError, match="no clustering"):
        gm.set_clustering_min_evals(0.0, 0.1)
    with pytest.raises(capi.AasrError, match="could not open"):
        gm.read_clustering(str(tmp_path / "missing.gcl"))
    gm.set_clustering(10, [(g, g % 10) for g in range(100)])
    gm.set_clustering_min_evals(0.0, 0.1)
    assert gm.num_clusters == 10
    gm.set_clustering(0)
    assert gm.num_clusters == 0
    frames = synth.make_frames(10, D=8)
    assert np.isfinite(gm.score(frames)).all()


@pytest.mark.parametrize("S,comps,G,C,nbytes", [(64, 16, 1024, 32, 2), (300, 9, 2700, 120, 4), (3125, 4, 12500, 500, 2)])
def test_clustered_scoring_to_lna(capi, oracle, golden_dir, S, comps, G, C, nbytes):
    """Clustered scoring for LNA consumers (run_utterance / the recipe driver / aasr_gmm_score_lna_dev)
    against the oracle's clustered phone_probs (codes within one step, log-probabilities within
    1e-4), incl. the register-resident packing instances for S = 64, 300 and 3125."""
    import os
    cfg = open(os.path.join(golden_dir, "mfcc_cms_norm.feaconf")).read()
    model = synth.make_model(D=39, G=G, S=S, comps=comps, seed=S)
    g2c = synth.make_clustering(model[0], C, iters=2)
    pairs = _pairs(g2c)
    ft = capi.Feat(cfg)
    gm = capi.Gmm.from_arrays(*model)
    gm.set_clustering(C, pairs)
    gm.set_clustering_min_evals(0.0, 0.2)
    pcm = synth.make_audio(20000 + 7 * S, seed=S + 1)
    fused, n = capi.run_utterance(ft, gm, pcm, lnabytes=nbytes)
    import torch
    d_fea = torch.from_numpy(ft.run(pcm, 0, n)).cuda()
    d_scr = torch.empty(gm.score_scratch_floats(n), dtype=torch.float32, device="cuda")
    d_by = torch.empty((n, S * nbytes), dtype=torch.uint8, device="cuda")
    gm.score_lna_dev(d_fea, d_scr, d_by, True, nbytes)
    torch.cuda.synchronize()
    assert d_by.cpu().numpy().tobytes() == fused[5:]
    ch = oracle.FeatureChain(cfg)
    om = oracle.DiagModel(*model)
    om.set_clustering(C, pairs, 0.0, 0.2)
    ll = om.score_clustered(ch.generate(pcm, 0, n))
    lp_ref, by_ref = oracle.lna_encode(np.maximum(np.exp(ll), 1e-50), True, nbytes)
    body = np.frombuffer(fused[5:], np.uint8).reshape(n, -1)
    if nbytes == 4:
        smooth = (ll > -87.0) | (ll < -104.5)
        assert np.abs(body.view("<f4") - lp_ref)[smooth].max() <= 1e-4
    else:
        a = body.reshape(n, S, 2).astype(int)
        b = np.asarray(by_ref).reshape(n, S, 2).astype(int)
        d = np.abs((a[..., 0] * 256 + a[..., 1]) - (b[..., 0] * 256 + b[..., 1]))
        assert d.max() <= 1 and (d == 0).mean() >= CODES_EQUAL_MIN


@pytest.mark.parametrize("D", [20, 80])
def test_clustering_under_a_global_cmllr_transform(capi, oracle, D):
    """phone_probs -C ... -S ... with a UNIT_NO (global) model transform: the pool's Gaussians are
    AdaptedGaussians -- members evaluated on A f + b and scaled by |prod diag A|
    (aku/ModelModules.hh:164-173) -- while the cluster centres are plain Gaussians ranked on the
    frame itself (aku/Distributions.cc:2688-2691).  Scores and per-frame exact counts against the
    oracle, whichever of clustering / transform is set first."""
    model = synth.make_model(D=D, G=1200, S=100, comps=12, seed=31)   # D = 80: the model as dimension parts
    mean, var, off, idx, w = model
    g2c = synth.make_clustering(mean, 48)
    pairs = _pairs(g2c)
    rng = np.random.default_rng(8)
    A = np.eye(D) * rng.uniform(0.9, 1.1, D) + 0.03 * rng.standard_normal((D, D))
    b = 0.2 * rng.standard_normal(D)
    W = np.hstack([b[:, None], A])
    frames = synth.make_frames(400, D=D, seed=9)
    om = oracle.DiagModel(mean, var, off, idx, w)
    om.set_clustering(48, pairs, 0.05, 0.2)
    want, want_n = om.score_clustered_adapted(frames.astype(np.float64), W, want_counts=True)
    plain = om.score_clustered(frames.astype(np.float64))
    assert np.abs(want - plain).max() > 0.05          # the transform matters
    g2t = np.zeros(1200, np.int32)
    for order in ("cluster_first", "transform_first"):
        gm = capi.Gmm.from_arrays(mean, var, off, idx, w)
        if order == "transform_first":
            gm.set_cmllr(g2t, W[None])
        gm.set_clustering(48, pairs)
        gm.set_clustering_min_evals(0.05, 0.2)
        if order == "cluster_first":
            gm.set_cmllr(g2t, W[None])
        for prec in (0, 3, 4):
            gm.set_precision(prec)
            got = gm.score(frames)
            assert np.array_equal(gm.cluster_exact_counts(len(frames)), want_n), (order, prec)
            assert_ll(got, want, "%s, precision %d" % (order, prec))
        gm.set_cmllr()                                  # back to the unadapted model
        assert_ll(gm.score(frames), plain, "back to the unadapted model")
        gm.close()


@pytest.mark.parametrize("D", [20, 80])
def test_clustering_under_per_class_cmllr_transforms(capi, oracle, D):
    """Regression-class transforms (UNIT_GAUSSIAN / UNIT_MIX / UNIT_PHONE speaker files) with -C: the
    Gaussians of one mixture -- and of one cluster -- may belong to different classes.  Each member is
    evaluated on its own class's A f + b and scaled by that class's |det|, Gaussians without a
    transform on the frame itself, the centres are plain (aku/ModelModules.hh:164-173,
    aku/Distributions.cc:2684-2722).  Scores and exact counts against the oracle in either order of
    set_clustering / set_cmllr, with a class change in between, a class whose determinant is 0, and
    back to the unadapted model."""
    mean, var, off, idx, w = synth.make_model(D=D, G=1200, S=100, comps=12, seed=31)
    g2c = synth.make_clustering(mean, 48)
    g2c[5] = g2c[77] = -1
    pairs = _pairs(g2c)
    rng = np.random.default_rng(18)

    def transform(scale=1.0):
        A = np.eye(D) * rng.uniform(0.9, 1.1, D) * scale + 0.03 * rng.standard_normal((D, D))
        return np.hstack([0.2 * rng.standard_normal(D)[:, None], A])

    W3 = np.stack([transform(), transform(), transform()])
    g2t = rng.integers(-1, 3, 1200).astype(np.int32)          # -1: unadapted Gaussians
    frames = synth.make_frames(400, D=D, seed=9)
    om = oracle.DiagModel(mean, var, off, idx, w)
    om.set_clustering(48, pairs, 0.05, 0.2)
    plain = om.score_clustered(frames.astype(np.float64))
    want, want_n = om.score_clustered_classes(frames.astype(np.float64), g2t, W3, want_counts=True)
    assert np.abs(want - plain).max() > 0.05
    for order in ("cluster_first", "transform_first"):
        gm = capi.Gmm.from_arrays(mean, var, off, idx, w)
        if order == "transform_first":
            gm.set_cmllr(g2t, W3)
        gm.set_clustering(48, pairs)
        gm.set_clustering_min_evals(0.05, 0.2)
        if order == "cluster_first":
            gm.set_cmllr(g2t, W3)
        for prec in (0, 3, 4):
            gm.set_precision(prec)
            got = gm.score(frames)
            assert np.array_equal(gm.cluster_exact_counts(len(frames)), want_n), (order, prec)
            assert_ll(got, want, "%s, precision %d" % (order, prec))
        # another speaker: new matrices, a different class membership, one singular class
        g2t2 = rng.integers(0, 2, 1200).astype(np.int32)
        W2 = np.stack([transform(), transform()])
        W2[1, 3, 1 + 3] = 0.0                                # diag product 0: that class's exact values vanish
        want2, n2 = om.score_clustered_classes(frames.astype(np.float64), g2t2, W2, want_counts=True)
        gm.set_cmllr(g2t2, W2)
        got2 = gm.score(frames)
        assert np.array_equal(gm.cluster_exact_counts(len(frames)), n2)
        assert_ll(got2, want2, "second speaker")
        gm.set_cmllr()
        assert_ll(gm.score(frames), plain, "back to the unadapted model")
        gm.close()


def test_clustering_with_outlier_routed_and_ill_conditioned_models(capi, oracle):
    """Production recognisers run clustered, and real models hold a few variance-floored Gaussians
    (kappa > 600: taken out of the matrix layouts, gmm.h).  The outliers' exact values come from the
    centred kernel under the same per-(cluster, frame) selection bits; a model that is ill-conditioned
    as a whole runs the centred kernel for every component.  Scores and exact-evaluation counts
    against the oracle's cluster branch, several thresholds, frames sitting on the outliers, a state
    made of outliers only, an outlier in no cluster."""
    import ctypes as C
    rng = np.random.default_rng(31)
    mean, var, off, idx, w = synth.make_model(D=39, G=512, S=48, comps=8, seed=13)
    bad = rng.choice(512, 24, replace=False)
    var[bad] *= 2e-3
    idx[off[5]:off[6]] = bad[:off[6] - off[5]]
    idx[off[9]] = bad[3]
    frames = synth.make_frames(300, seed=8)
    frames[:24] = (mean[bad] + np.sqrt(var[bad]) * rng.standard_normal((24, 39))).astype(np.float32)
    g2c = synth.make_clustering(mean, 32)
    g2c[bad[0]] = -1                                   # an outlier that belongs to no cluster
    g2c[7] = -1
    L = capi.lib()
    L.aasr_debug_kappa.restype = C.c_double
    L.aasr_debug_kappa.argtypes = [C.c_void_p]
    for minc, ming in ((0.0, 0.25), (0.2, 0.0), (0.0, 0.0), (1.0, 1.0)):
        gm, om, got, want = _check(capi, oracle, (mean, var, off, idx, w), g2c, 32, minc, ming, frames)
        assert L.aasr_debug_kappa(gm._h) > 600 and gm.active_layout() in (1, 2)   # outlier routing, not all-centred
        if minc == 1.0:
            assert np.abs(got - om.score(frames.astype(np.float64))).max() <= TOL
        gm.close()
    # the same model under one global CMLLR transform (|det| folded into the weights of rows and
    # centred records): adapted members, plain centres
    A = np.eye(39) * rng.uniform(0.95, 1.05, 39) + 0.01 * rng.standard_normal((39, 39))
    W = np.hstack([0.05 * rng.standard_normal(39)[:, None], A])
    om = oracle.DiagModel(mean, var, off, idx, w)
    om.set_clustering(32, _pairs(g2c), 0.0, 0.25)
    want, want_n = om.score_clustered_adapted(frames.astype(np.float64), W, want_counts=True)
    gm = capi.Gmm.from_arrays(mean, var, off, idx, w)
    gm.set_clustering(32, _pairs(g2c))
    gm.set_clustering_min_evals(0.0, 0.25)
    gm.set_cmllr(np.zeros(512, np.int32), W[None])
    for prec in (0, 3, 4):
        gm.set_precision(prec)
        got = gm.score(frames)
        assert np.array_equal(gm.cluster_exact_counts(len(frames)), want_n)
        assert_ll(got, want, "clustered + adapted, prec %d" % prec)
    gm.close()
    # a majority of tight Gaussians: the whole model in the centred form, still clustered
    var2 = var.copy()
    var2[rng.choice(512, 300, replace=False)] *= 2e-3
    om = oracle.DiagModel(mean, var2, off, idx, w)
    gm = capi.Gmm.from_arrays(mean, var2, off, idx, w)
    assert gm.active_layout() == 4
    gm.set_clustering(32, _pairs(g2c))
    for minc, ming in ((0.0, 0.25), (0.3, 0.1), (0.0, 0.0)):
        om.set_clustering(32, _pairs(g2c), minc, ming)
        want, want_n = om.score_clustered(frames.astype(np.float64), want_counts=True)
        gm.set_clustering_min_evals(minc, ming)
        got = gm.score(frames)
        assert np.array_equal(gm.cluster_exact_counts(len(frames)), want_n)
        assert_ll(got, want, "centred form, clustered (%g, %g)" % (minc, ming))
    gm.close()
