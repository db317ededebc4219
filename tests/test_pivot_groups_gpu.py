"""Multi-pivot engine parts (gmm_plan_engine_parts): models FITTED to data -- the shape
HmmSet::read_all (aku/HmmSet.cc:351-357) loads in production -- whose Gaussians sit far from the pool's one pivot in units
of their own standard deviations.  The states are sorted into pivot groups, every group expanded around its own pivot
inside ONE launch of the scoring kernel; parity against the oracle (aku/Distributions.cc:1040-1062, 2078-2086,
aku/HmmSet.cc:484-501) on the public layout, on the engine's own layout through the LNA pass
(aku/phone_probs.cc:224-262), and the structure of the column map."""
import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-4
VISIBLE = -103.0


def blobs(F, D=39, seed=1, n_blobs=6, spread=3.0):
    """Frames from a few well separated clusters with different scales: what makes a fitted model need several pivots."""
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((n_blobs, D)) * spread
    cent[0] = 0
    sc = np.exp(rng.uniform(np.log(0.3), np.log(1.5), (n_blobs, D)))
    which = rng.integers(0, n_blobs, F)
    return (cent[which] + rng.standard_normal((F, D)) * sc[which]).astype(np.float32)


def visible_err(got, ref):
    vis = ref > VISIBLE
    return float(np.abs(got - ref)[vis].max()), int(vis.sum())


@pytest.fixture(scope="module")
def fitted():
    X = blobs(30000)
    return X, synth.fit_model(X, S=200, comps=16)


def test_fitted_model_gets_pivot_groups_and_matches_the_oracle(capi, oracle, fitted, monkeypatch):
    X, model = fitted
    monkeypatch.setenv("AASR_PG_PIVOT_COST", "64")   # (test hook: a small model would otherwise not pay for more pivots)
    g = capi.Gmm.from_arrays(*model)
    parts = g.engine_parts()
    assert parts is not None, g.engine_plan_note()
    p0 = parts["parts"][0]
    assert p0["arith"] == 2 and p0["pivot_groups"] >= 3, parts
    assert sum(p["states"] for p in parts["parts"]) == 200
    n16, moved = g.precision_states()
    assert n16 >= p0["states"] and g.effective_precision() == 4
    k, k2 = synth.conditioning(model[0], model[1])
    assert k2.max() > 80.0     # around the pool's one pivot the model is over the two-term limits
    fr = np.ascontiguousarray(X[:2048])
    ref = oracle.DiagModel(*model).score(fr.astype(np.float64))
    err, n = visible_err(g.score(fr), ref)
    assert n > 10000 and err <= TOL, (err, n)
    g.close()


def test_column_map_is_a_permutation_into_whole_lines(capi, fitted, monkeypatch):
    X, model = fitted
    monkeypatch.setenv("AASR_PG_PIVOT_COST", "64")
    g = capi.Gmm.from_arrays(*model)
    parts = g.engine_parts()
    colmap, col0, begin, real_end, piv = g.engine_layout(0)
    assert col0 == 0 and len(set(colmap.tolist())) == 200 and colmap.max() < parts["cols"]
    assert (begin % 32 == 0).all() and (real_end > begin).all() and (real_end[:-1] <= begin[1:]).all()
    in0 = colmap < real_end[-1]
    assert in0.sum() == parts["parts"][0]["states"]
    # every state of part 0 sits in a real column of exactly one group, and is well conditioned around THAT group's pivot
    mean, var, off, idx, w = model
    for s in np.flatnonzero(in0):
        grp = np.flatnonzero((begin <= colmap[s]) & (colmap[s] < real_end))
        assert grp.size == 1
        gi = idx[off[s]:off[s + 1]]
        kk, kk2 = synth.conditioning(mean[gi], var[gi], piv[grp[0]].astype(np.float64))
        assert kk.max() <= 250.0 * 1.0001 and kk2.max() <= 80.0 * 1.0001
    assert g.score_scratch_floats(100) >= 100 * parts["cols"]
    g.close()


def test_engine_layout_through_the_lna_pass(capi, oracle, fitted, monkeypatch):
    """aasr_gmm_score_lna_dev: the parts score into their own column ranges, the LNA pass reads through the column map."""
    import torch
    X, model = fitted
    monkeypatch.setenv("AASR_PG_PIVOT_COST", "64")
    g = capi.Gmm.from_arrays(*model)
    fr = np.ascontiguousarray(X[5000:5000 + 700])
    S = g.num_states
    d_f = torch.from_numpy(fr).cuda()
    d_scr = torch.empty(g.score_scratch_floats(len(fr)), dtype=torch.float32, device="cuda")
    ref_ll, ref_lik = oracle.DiagModel(*model).score(fr.astype(np.float64), want_lik=True)
    for nbytes in (4, 2):
        d_by = torch.empty((len(fr), S * nbytes), dtype=torch.uint8, device="cuda")
        g.score_lna_dev(d_f, d_scr, d_by, True, nbytes)
        torch.cuda.synchronize()
        lp_ref, by_ref = oracle.lna_encode(ref_lik, True, nbytes)
        by = d_by.cpu().numpy()
        if nbytes == 4:
            lp = by.view("<f4").reshape(len(fr), S)
            m = ref_ll > -87.0   # (below: the reference stores a DENORMAL float likelihood, conftest.assert_lp_denormal_band)
            assert np.abs(lp.astype(np.float64) - lp_ref)[m].max() <= TOL
        else:
            a = by.reshape(len(fr), S, 2).astype(np.int32)
            b = by_ref.reshape(len(fr), S, 2).astype(np.int32)
            ca, cb = a[..., 0] * 256 + a[..., 1], b[..., 0] * 256 + b[..., 1]
            assert np.abs(ca - cb).max() <= 1 and (ca == cb).mean() > 0.99
    g.close()


def test_single_group_many_groups_and_odd_shapes(capi, oracle, monkeypatch):
    """Groups of one state, an odd state count per group, ragged component counts are refused by the grouped layout and fall
    back; dimensions other than 39; many frames (8-wave workgroups, two-level cut plans) and few (4-wave)."""
    monkeypatch.setenv("AASR_PG_PIVOT_COST", "16")
    for D, S, comps, F in ((39, 37, 8, 300), (13, 96, 4, 9000), (24, 130, 16, 1500), (39, 64, 16, 20000)):
        X = blobs(12000, D=D, seed=3 + D, n_blobs=5, spread=4.0)
        model = synth.fit_model(X, S=S, comps=comps, seed=11 + S)
        g = capi.Gmm.from_arrays(*model)
        fr = np.ascontiguousarray(np.tile(X, (F // len(X) + 1, 1))[:F])
        sub = np.arange(0, F, max(1, F // 256))
        ref = oracle.DiagModel(*model).score(fr[sub].astype(np.float64))
        got = g.score(fr)[sub]
        err, n = visible_err(got, ref)
        assert err <= TOL, (D, S, comps, err, g.engine_parts(), g.engine_plan_note())
        g.close()


def test_probe_rejects_move_to_the_next_part():
    """The load-time probe (gmm_probe_f16x2) runs on the two-term part; states it rejects are taken out and the part is
    built again.  With the test hook's tolerance most probed states are rejected: the model must still be scored right
    (a process of its own: the library reads the hook once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from aaltoasr_amd import capi, synth
from oracle import oracle as O
from test_pivot_groups_gpu import blobs
capi.check(capi.lib().aasr_set_device(0))
X = blobs(30000)
model = synth.fit_model(X, S=200, comps=16)
g = capi.Gmm.from_arrays(*model)
n16, moved = g.precision_states()
fr = np.ascontiguousarray(X[:600])
ref = O.DiagModel(*model).score(fr.astype(np.float64))
err = float(np.abs(g.score(fr) - ref)[ref > -103.0].max())
print("RESULT", n16, moved, err)
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, AASR_PG_PIVOT_COST="64", AASR_F16_PROBE_TOL="1.5e-5")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    n16, moved, err = r.stdout.split("RESULT")[1].split()
    assert int(moved) > 0 and float(err) <= TOL, r.stdout


def test_global_transform_and_other_precisions_on_a_model_with_parts(capi, oracle, fitted, monkeypatch):
    """One constrained-MLLR transform for the whole pool (aku/ModelModules.hh:164-212) is applied to the frames in front of
    the parts; the verification precisions run on the model's own layouts."""
    X, model = fitted
    monkeypatch.setenv("AASR_PG_PIVOT_COST", "64")
    g = capi.Gmm.from_arrays(*model)
    fr = np.ascontiguousarray(X[100:400])
    om = oracle.DiagModel(*model)
    ref = om.score(fr.astype(np.float64))
    for prec in (0, 3, 4):
        g.set_precision(prec)
        err, n = visible_err(g.score(fr), ref)
        assert err <= TOL, (prec, err)
    rng = np.random.default_rng(8)
    D = fr.shape[1]
    A = np.eye(D) + 0.05 * rng.standard_normal((D, D))
    b = 0.1 * rng.standard_normal(D)
    W = np.concatenate([b[:, None], A], 1)[None]
    g.set_cmllr(np.zeros(g.num_gaussians, np.int32), W)
    ref_t = om.score(fr.astype(np.float64) @ A.T + b) + np.log(np.abs(np.prod(np.diag(A))))
    err, n = visible_err(g.score(fr), ref_t)
    assert g.engine_parts() is not None and err <= TOL, (err, g.engine_parts())
    g.set_cmllr()
    err, n = visible_err(g.score(fr), ref)
    assert err <= TOL
    g.close()


def test_gaussian_clustering_over_engine_parts(capi, oracle, fitted, monkeypatch):
    """The clustered pass (aku/Distributions.cc:2684-2722; what pyrectool always runs) on a model with engine parts:
    every part gives the exact part of its states under the same selection bits, the merge adds the centres' share through
    the column map.  Scores within 1e-4 of the oracle's cluster branch, exact-evaluation counts bit-equal; also through
    the LNA pass and with padded rows."""
    import torch
    X, model = fitted
    monkeypatch.setenv("AASR_PG_PIVOT_COST", "64")
    mean, var, off, idx, w = model
    C = 48
    g2c = synth.make_clustering(mean, C)
    pairs = [(int(a), int(c)) for a, c in enumerate(g2c)]
    fr = np.ascontiguousarray(X[2000:2000 + 900])
    for minc, ming in ((0.0, 0.25), (0.2, 0.0), (1.0, 1.0)):
        om = oracle.DiagModel(*model)
        om.set_clustering(C, pairs, minc, ming)
        want, want_n = om.score_clustered(fr.astype(np.float64), want_counts=True)
        g = capi.Gmm.from_arrays(*model)
        assert g.engine_parts() is not None
        g.set_clustering(C, pairs)
        g.set_clustering_min_evals(minc, ming)
        got = g.score(fr)
        assert np.array_equal(g.cluster_exact_counts(len(fr)), want_n)
        vis = want > VISIBLE
        assert np.abs(got - want)[vis].max() <= TOL, (minc, ming, np.abs(got - want)[vis].max())
        if minc == 0.0:
            assert g.score_pitch_ok()
            d_fr = torch.from_numpy(fr).cuda()
            padded = torch.full((len(fr), 224), -7.0, dtype=torch.float32, device="cuda")
            g.score_dev_pitched(d_fr, padded, 224)
            torch.cuda.synchronize()
            p = padded.cpu().numpy()
            assert np.array_equal(p[:, :200], got) and (p[:, 200:] == -7.0).all()
            d_scr = torch.empty(g.score_scratch_floats(len(fr)), dtype=torch.float32, device="cuda")
            d_by = torch.empty((len(fr), 200 * 2), dtype=torch.uint8, device="cuda")
            g.score_lna_dev(d_fr, d_scr, d_by, True, 2)
            torch.cuda.synchronize()
            _, by_ref = oracle.lna_encode(np.exp(want), True, 2)
            a = d_by.cpu().numpy().reshape(len(fr), 200, 2).astype(np.int32)
            b = by_ref.reshape(len(fr), 200, 2).astype(np.int32)
            assert np.abs((a[..., 0] * 256 + a[..., 1]) - (b[..., 0] * 256 + b[..., 1])).max() <= 1
        g.close()


def test_lna_path_at_the_verification_precisions_on_a_model_with_parts(capi, oracle, fitted, monkeypatch):
    """with engine parts planned but not active (a precision other than the default), the engine's row
    pitch must be the model's own -- the LNA entry point (aasr_gmm_score_lna_dev) used to ask outlier-routed / centred
    models for a pitched launch they do not have."""
    import torch
    X, model = fitted
    monkeypatch.setenv("AASR_PG_PIVOT_COST", "64")
    g = capi.Gmm.from_arrays(*model)
    assert g.engine_parts() is not None
    fr = np.ascontiguousarray(X[7000:7000 + 300])
    S = g.num_states
    d_f = torch.from_numpy(fr).cuda()
    d_scr = torch.empty(g.score_scratch_floats(len(fr)), dtype=torch.float32, device="cuda")
    ref_ll, ref_lik = oracle.DiagModel(*model).score(fr.astype(np.float64), want_lik=True)
    _, by_ref = oracle.lna_encode(ref_lik, True, 2)
    b = by_ref.reshape(len(fr), S, 2).astype(np.int32)
    cb = b[..., 0] * 256 + b[..., 1]
    ok = (ref_ll.max(1, keepdims=True) > -80.0) & (ref_ll > -87.0)
    for prec in (0, 3, 2, 4):
        g.set_precision(prec)
        d_by = torch.empty((len(fr), S * 2), dtype=torch.uint8, device="cuda")
        g.score_lna_dev(d_f, d_scr, d_by, True, 2)
        torch.cuda.synchronize()
        a = d_by.cpu().numpy().reshape(len(fr), S, 2).astype(np.int32)
        ca = a[..., 0] * 256 + a[..., 1]
        assert np.abs(ca - cb)[ok].max() <= 1, prec
    g.close()


def test_a_second_clustering_with_the_same_cluster_count_reaches_the_parts(capi, oracle, fitted, monkeypatch):
    """aasr_gmm_set_clustering with another Gaussian -> cluster assignment but the same number of clusters:
    the engine parts' views of the clustering (rows' clusters, masks) must be rebuilt, not kept."""
    X, model = fitted
    monkeypatch.setenv("AASR_PG_PIVOT_COST", "64")
    mean = model[0]
    C = 40
    fr = np.ascontiguousarray(X[3000:3000 + 400])
    g = capi.Gmm.from_arrays(*model)
    assert g.engine_parts() is not None
    for seed in (5, 6):
        g2c = synth.make_clustering(mean, C, seed=seed, iters=1 + seed % 2)
        pairs = [(int(a), int(c)) for a, c in enumerate(g2c)]
        om = oracle.DiagModel(*model)
        om.set_clustering(C, pairs, 0.0, 0.25)
        want, want_n = om.score_clustered(fr.astype(np.float64), want_counts=True)
        g.set_clustering(C, pairs)
        g.set_clustering_min_evals(0.0, 0.25)
        got = g.score(fr)
        assert np.array_equal(g.cluster_exact_counts(len(fr)), want_n), seed
        vis = want > VISIBLE
        assert np.abs(got - want)[vis].max() <= TOL, (seed, np.abs(got - want)[vis].max())
    g.close()


def test_probe_verdicts_are_cached_and_a_rejected_state_stays_rejected():
    """Round-5 review item 8.  (a) The load-time probe's verdict is cached per model content: building the same model a second
    time takes the verdicts instead of scoring the probe frames again (aasr_debug_probe_counts).  (b) With the test hook's
    tolerance the probe takes the whole-model two-term rows away; a layout built LATER (aasr_debug_set_layouts builds the
    independent tracks on demand) must not pack them again: under that layout the default precision then runs the
    three-term rows, bit for bit what AASR_PREC_BF16X3 gives.  (A process of its own: the library reads the hook once.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, ctypes, numpy as np
sys.path.insert(0, %r)
from aaltoasr_amd import capi, synth
capi.check(capi.lib().aasr_set_device(0))
L = capi.lib()
L.aasr_debug_probe_counts.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
L.aasr_debug_probe_counts.restype = None
def counts():
    r, h = ctypes.c_int64(0), ctypes.c_int64(0)
    L.aasr_debug_probe_counts(ctypes.byref(r), ctypes.byref(h))
    return r.value, h.value
model = synth.make_model(D=39, G=64 * 8, S=64, comps=8, seed=77)
r0, h0 = counts()
g = capi.Gmm.from_arrays(*model)
r1, h1 = counts()
g2 = capi.Gmm.from_arrays(*model)
r2, h2 = counts()
n16, moved = g.precision_states()
fr = synth.make_frames(200, seed=78)
g.set_layouts(2)
g.set_precision(4)
s4 = g.score(fr)
g.set_precision(3)
s3 = g.score(fr)
print("RESULT", r1 - r0, h1 - h0, r2 - r1, h2 - h1, moved, int(np.array_equal(s4, s3)), g.active_layout())
""" % root
    env = dict(os.environ, AASR_F16_PROBE_TOL="1.0e-6")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    runs1, hits1, runs2, hits2, moved, same, layout = [int(x) for x in r.stdout.split("RESULT")[1].split()]
    assert runs1 > 0 and runs2 == 0 and hits2 > 0, r.stdout      # the second build ran no probe on the device
    assert moved > 0 and layout == 2 and same == 1, r.stdout       # the late layout did not re-admit the rejected rows


@pytest.mark.parametrize("D", [13, 24, 39, 47, 55])
def test_slab_constant_part_in_other_dimensions(capi, oracle, D):
    """The slab-constant K layout (TrackLayout::sc: seven dimensions + their share of the constant per slab of 16 K slots)
    picks another kernel instance per dimension (2 slabs at 13 dimensions ... 8 at 55).  Half of the states hold one Gaussian
    over the plain two-term limits -- too many for outlier routing -- so they form the second engine part; frames on and
    off the model; the clustered pass over the parts as well."""
    S = 128
    base = synth.make_model(D=D, G=S * 8, S=S, comps=8, seed=600 + D)
    bad = list(range(0, S, 2))
    model = synth.push_states_over_the_f16_limits(base, bad, kappa2=100.0)
    g = capi.Gmm.from_arrays(*model)
    parts = g.engine_parts()
    assert parts is not None and any(p["arith"] == 4 for p in parts["parts"]), (parts, g.engine_plan_note())
    rng = np.random.default_rng(D)
    fr = synth.make_frames(600, D=D, seed=601 + D)
    fr[300:] = (fr[300:] * 2.5).astype(np.float32)          # far from every Gaussian
    om = oracle.DiagModel(*model)
    ref = om.score(fr.astype(np.float64))
    err, n = visible_err(g.score(fr), ref)
    assert n > 1000 and err <= TOL, (D, err, parts)
    C = 16
    g2c = synth.make_clustering(model[0], C, seed=3)
    pairs = [(int(a), int(c)) for a, c in enumerate(g2c)]
    om.set_clustering(C, pairs, 0.0, 0.25)
    want, want_n = om.score_clustered(fr.astype(np.float64), want_counts=True)
    g.set_clustering(C, pairs)
    g.set_clustering_min_evals(0.0, 0.25)
    got = g.score(fr)
    assert np.array_equal(g.cluster_exact_counts(len(fr)), want_n), D
    vis = want > VISIBLE
    assert np.abs(got - want)[vis].max() <= TOL, (D, np.abs(got - want)[vis].max())
    g.close()
