"""Per-state precision routing: a model whose Gaussians do not ALL satisfy the plain two-term layout's conditioning limits
keeps the fastest rows for the states that qualify and gives the others what their conditioning needs -- since round 6
through ONE mechanism, the engine parts (gmm_plan_engine_parts: plain rows around a group's pivot, the slab-constant layout,
the centred form), or, for a handful of far-out Gaussians, outlier routing on the model's own layout; rounds 4-5 had a
"mixed" layout of two sections for it, whose tests these were.  States are independent output columns
(Mixture::compute_likelihood, aku/Distributions.cc:2078-2086), so the parity bar is the usual one: every state
log-likelihood within 1e-4 of the oracle's (aku/HmmSet.cc:484-501)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import assert_ll

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _routed(capi, oracle, model, bad, frames, grouped=None, clustered=False):
    """Scores `frames` in every precision; returns the default precision's scores.  `bad`: the states over the plain limits."""
    S = len(model[2]) - 1
    ref = oracle.DiagModel(*model).score(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(*model)
    assert g.get_precision() == 4 and g.effective_precision() == 4
    n16, moved = g.precision_states()
    # (states of `bad` that the engine's planner moved to the remainder are models of their own around THEIR pivot and may
    # qualify for the plain two-term layout there)
    assert S - len(bad) <= n16 <= S, (n16, moved, S, len(bad))
    if grouped is not None:
        assert g.active_layout() == (1 if grouped else 2)
    got4 = g.score(frames)
    assert_ll(got4, ref, "routed model, default precision")
    for prec in (3, 0):
        g.set_precision(prec)
        assert g.effective_precision() == prec and g.precision_states()[0] == 0
        assert_ll(g.score(frames), ref, "precision %d on the routed model" % prec)
    g.set_precision(4)
    assert np.array_equal(g.score(frames), got4)       # back on the default precision: the same bits
    g.close()
    return got4, ref


@pytest.mark.parametrize("share", ["one", 0.01, 0.1, 0.4])
def test_routed_states_match_the_oracle(capi, oracle, share):
    """configs[1]'s layout in miniature (16 components per state, disjoint pool, grouped tracks) with one state, 1 %, 10 %
    and 40 % of the states holding a Gaussian over the fp16 limits; frame counts on both sides of the 8-wave form's
    threshold and not a multiple of a wave's 64 frames."""
    S = 300
    base = synth.make_model(D=39, G=S * 16, S=S, comps=16, seed=411)
    rng = np.random.default_rng(5)
    if share == "one":
        bad = [137]
    else:
        bad = sorted(rng.choice(S, max(1, int(round(share * S))), replace=False).tolist())
    model = synth.push_states_over_the_f16_limits(base, bad)
    for F in (70, 9001):
        _routed(capi, oracle, model, bad, synth.make_frames(F, seed=412 + F), grouped=True)


def test_routed_states_at_group_edges_and_small_models(capi, oracle):
    """Pairs are formed inside groups of 16 output columns and flushed per group of 32 (or 16): states next to the group
    boundaries, whole groups on one side, an odd state count, fewer states than one group."""
    for S, bad in ((97, [0, 15, 16, 31, 32, 33, 63, 64, 96]), (96, list(range(64, 96))), (96, list(range(0, 32)) + [40]),
                   (5, [2]), (33, [32]), (3125 // 25, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])):
        base = synth.make_model(D=39, G=S * 8, S=S, comps=8, seed=420 + S)
        model = synth.push_states_over_the_f16_limits(base, bad)
        _routed(capi, oracle, model, bad, synth.make_frames(333, seed=421), grouped=True)


def test_routed_states_on_independent_tracks_and_other_dimensions(capi, oracle):
    """Ragged mixtures take the independent-track layout (states stored one by one): its sections need no pair table.
    Other feature dimensions pick other kernel instances."""
    base = synth.make_model(D=39, G=4000, S=150, comps_range=(1, 40), seed=430)
    bad = [3, 4, 77, 149]
    model = synth.push_states_over_the_f16_limits(base, bad)
    _routed(capi, oracle, model, bad, synth.make_frames(700, seed=431), grouped=False)
    for D in (13, 24, 47):
        base = synth.make_model(D=D, G=96 * 8, S=96, comps=8, seed=432 + D)
        bad = [0, 50, 51, 95]
        model = synth.push_states_over_the_f16_limits(base, bad, kappa2=100.0)
        _routed(capi, oracle, model, bad, synth.make_frames(400, D=D, seed=433), grouped=True)


def test_a_model_with_most_states_over_the_limits_is_split_into_engine_parts(capi, oracle):
    """Beyond 45 % of the states the two launches of the MIXED layout cost what the whole model costs on three bf16 terms
    (the second section's scattered stores), so no mixed layout is built.  Since round 5 such a model is scored through
    its engine parts instead (gmm_plan_engine_parts): the states that qualify keep two fp16 terms, the others take three
    bf16 terms as a model of their own, public-layout calls get the columns gathered back."""
    S = 64
    bad = list(range(0, S, 2)) + [1]
    model = synth.push_states_over_the_f16_limits(synth.make_model(D=39, G=S * 8, S=S, comps=8, seed=470), bad)
    g = capi.Gmm.from_arrays(*model)
    parts = g.engine_parts()
    assert parts is not None and parts["parts"][0]["arith"] == 2, g.engine_plan_note()
    assert g.effective_precision() == 4 and g.precision_states()[0] >= S - len(bad)
    fr = synth.make_frames(300, seed=471)
    ref = oracle.DiagModel(*model).score(fr.astype(np.float64))
    assert_ll(g.score(fr), ref, "engine parts, public layout")
    g.set_precision(3)   # the other arithmetics stay on the model's own layouts
    assert g.effective_precision() == 3 and g.precision_states() == (0, 0)
    assert_ll(g.score(fr), ref, "unrouted model, three terms")
    g.close()


def test_routed_model_with_padded_rows_and_device_pointers(capi, oracle):
    """aasr_gmm_score_dev_pitched on a routed model: rows padded to whole 128-byte lines, bit for bit the dense values,
    the padding untouched (the two sections write disjoint columns of the same lines)."""
    import torch
    S = 100
    bad = [7, 8, 40, 99]
    model = synth.push_states_over_the_f16_limits(synth.make_model(D=39, G=S * 16, S=S, comps=16, seed=440), bad)
    g = capi.Gmm.from_arrays(*model)
    assert S - len(bad) <= g.precision_states()[0] <= S and g.score_pitch_ok()
    fr = synth.make_frames(9000, seed=441)
    dense = g.score(fr)
    d_fr = torch.from_numpy(fr).cuda()
    for pitch in (128, 103):
        padded = torch.full((len(fr), pitch), -7.0, dtype=torch.float32, device="cuda")
        g.score_dev_pitched(d_fr, padded, pitch)
        torch.cuda.synchronize()
        out = padded.cpu().numpy()
        assert np.array_equal(out[:, :S], dense) and np.all(out[:, S:] == -7.0), pitch
    g.close()


def test_routed_model_under_gaussian_clustering(capi, oracle):
    """The masked (clustered) pass on the mixed layout: scores and exact-evaluation counts as the oracle's cluster branch
    (aku/Distributions.cc:2684-2722)."""
    S = 128
    bad = [5, 6, 64, 100, 127]
    model = synth.push_states_over_the_f16_limits(synth.make_model(D=39, G=S * 16, S=S, comps=16, seed=450), bad)
    g2c = synth.make_clustering(model[0], 64)
    pairs = [(int(a), int(c)) for a, c in enumerate(g2c) if c >= 0]
    frames = synth.make_frames(500, seed=451)
    om = oracle.DiagModel(*model)
    om.set_clustering(64, pairs, 0.0, 0.25)
    want, want_n = om.score_clustered(frames.astype(np.float64), want_counts=True)
    gm = capi.Gmm.from_arrays(*model)
    assert S - len(bad) <= gm.precision_states()[0] <= S
    gm.set_clustering(64, pairs)
    gm.set_clustering_min_evals(0.0, 0.25)
    got = gm.score(frames)
    assert np.array_equal(gm.cluster_exact_counts(len(frames)), want_n)
    assert np.abs(got - want).max() <= 1e-4
    gm.close()


def test_probe_guard_moves_states_that_fail_it(capi, oracle, tmp_path):
    """The load-time probe (gmm_probe_f16x2): with its tolerance forced to nothing (AASR_F16_PROBE_TOL, a test hook)
    every probed state fails and is scored with three terms instead -- the model still matches the oracle, and
    aasr_gmm_precision_states reports the move.  (In-process the guard runs on every model of the suite at 5e-5.)"""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from aaltoasr_amd import capi, synth
from oracle import oracle as O
capi.check(capi.lib().aasr_set_device(0))
model = synth.make_model(D=39, G=64 * 16, S=64, comps=16, seed=460)
g = capi.Gmm.from_arrays(*model)
n16, moved = g.precision_states()
fr = synth.make_frames(300, seed=461)
err = float(np.abs(g.score(fr) - O.DiagModel(*model).score(fr.astype(np.float64))).max())
print("RESULT", n16, moved, g.effective_precision(), err)
''' % ROOT
    out = {}
    for tol in ("", "1e-9"):
        env = dict(os.environ)
        if tol:
            env["AASR_F16_PROBE_TOL"] = tol
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tol] = r.stdout.split("RESULT")[1].split()
    assert out[""][:3] == ["64", "0", "4"] and float(out[""][3]) <= 1e-4
    n16, moved = int(out["1e-9"][0]), int(out["1e-9"][1])
    assert moved > 32 and n16 == 64 - moved and float(out["1e-9"][3]) <= 1e-4, out


def test_routed_model_on_the_engines_own_score_layout(capi, oracle):
    """aasr_gmm_score_lna_dev / aasr_run_utterance on a routed model: the second section is scored as a model of its own
    into spare columns behind the state columns (whole-line stores) and the LNA pass reads the rows through a column map.
    The packed bytes must be the ones the public layout gives -- aasr_gmm_score (columns = states) followed by
    aasr_lna_encode -- except where the sub-model's own pivot moves a value by rounding (2-byte codes at most one step,
    > 99 % identical), and within the usual bounds of the oracle's."""
    import torch
    S = 200
    bad = sorted(np.random.default_rng(3).choice(S, 20, replace=False).tolist())
    model = synth.push_states_over_the_f16_limits(synth.make_model(D=39, G=S * 16, S=S, comps=16, seed=480), bad)
    g = capi.Gmm.from_arrays(*model)
    assert S - len(bad) <= g.precision_states()[0] <= S
    F = 9000
    fr = synth.make_frames(F, seed=481)
    n_scratch = g.score_scratch_floats(F)
    parts = g.engine_parts()
    assert parts is not None and parts["cols"] == 192 + 32 and parts["parts"][0]["states"] == S - len(bad)
    assert n_scratch >= F * 224        # 180 states on two terms -> 192 columns, + one line of columns for the 20 others
    d_fr = torch.from_numpy(fr).cuda()
    d_scr = torch.empty(n_scratch, dtype=torch.float32, device="cuda")
    public = g.score(fr)
    ref_ll, lik = oracle.DiagModel(*model).score(fr.astype(np.float64), want_lik=True)
    assert_ll(public, ref_ll, "public layout")
    for nb in (2, 4):
        d_by = torch.empty((F, S * nb), dtype=torch.uint8, device="cuda")
        g.score_lna_dev(d_fr, d_scr, d_by, True, nb)
        torch.cuda.synchronize()
        got = d_by.cpu().numpy()
        _, want = capi.lna_encode(public, True, nb)
        lp_ref, by_ref = oracle.lna_encode(lik, True, nb)
        if nb == 2:
            a = got.reshape(F, S, 2).astype(np.int32)
            b = want.reshape(F, S, 2).astype(np.int32)
            ca, cb = a[..., 0] * 256 + a[..., 1], b[..., 0] * 256 + b[..., 1]
            assert np.abs(ca - cb).max() <= 1 and (ca == cb).mean() > 0.99
            # (the engine parts expand around pivots of their own, the public layout -- the mixed layout here -- around the
            # pool's: rounding-level differences, a code step at most)
            d_raw = torch.empty((F, S * 2), dtype=torch.uint8, device="cuda")
            g.score_lna_dev(d_fr, d_scr, d_raw, False, 2)
            torch.cuda.synchronize()
            r = d_raw.cpu().numpy().reshape(F, S, 2).astype(np.int32)
            w = capi.lna_encode(public, False, 2)[1].reshape(F, S, 2).astype(np.int32)
            cr, cw = r[..., 0] * 256 + r[..., 1], w[..., 0] * 256 + w[..., 1]
            assert np.abs(cr - cw).max() <= 1 and (cr == cw).mean() > 0.99
            o = by_ref.reshape(F, S, 2).astype(np.int32)
            co = o[..., 0] * 256 + o[..., 1]
            assert np.abs(ca - co).max() <= 1
        else:
            lp = got.view("<f4").reshape(F, S)
            lp_pub = want.view("<f4").reshape(F, S)
            vis = ref_ll > -85
            assert np.abs(lp - lp_pub)[vis].max() <= 8e-5    # (the sub-model has its own pivot and may run two fp16 terms where the public layout runs three bf16 terms)
            assert np.abs(lp - lp_ref)[vis].max() <= 1e-4
    # clustering merges by state column: the engine layout steps aside, the bytes stay right
    g2c = synth.make_clustering(model[0], 64)
    g.set_clustering(64, [(int(a), int(c)) for a, c in enumerate(g2c)])
    g.set_clustering_min_evals(1.0, 1.0)
    d_by = torch.empty((F, S * 2), dtype=torch.uint8, device="cuda")
    g.score_lna_dev(d_fr, d_scr, d_by, True, 2)
    torch.cuda.synchronize()
    a = d_by.cpu().numpy().reshape(F, S, 2).astype(np.int32)
    o = oracle.lna_encode(lik, True, 2)[1].reshape(F, S, 2).astype(np.int32)
    assert np.abs((a[..., 0] * 256 + a[..., 1]) - (o[..., 0] * 256 + o[..., 1])).max() <= 1
    g.close()


def test_routed_model_under_a_global_transform_and_through_the_model_cache(capi, oracle, tmp_path):
    """A global constrained-MLLR transform set on a routed model (updated in place: adapted frames, log|det| at the kernels'
    output -- both sections), taken off again; and the same model through the binary model cache (aasr_gmm_write_cache /
    create_from_cache): the routing is rebuilt from the parsed model, the scores are the same bits."""
    S, D = 96, 39
    bad = [3, 40, 41, 77]
    model = synth.push_states_over_the_f16_limits(synth.make_model(D=D, G=S * 8, S=S, comps=8, seed=490), bad)
    fr = synth.make_frames(500, seed=491)
    g = capi.Gmm.from_arrays(*model)
    assert S - len(bad) <= g.precision_states()[0] <= S
    plain = g.score(fr)
    rng = np.random.default_rng(492)
    A = np.eye(D) * rng.uniform(0.95, 1.05, D) + 0.01 * rng.standard_normal((D, D))
    b = 0.1 * rng.standard_normal(D)
    W = np.hstack([b[:, None], A])
    g.set_cmllr(np.zeros(S * 8, np.int32), W[None])
    want = oracle.score_adapted(oracle.DiagModel(*model), fr.astype(np.float64), np.zeros(S * 8, np.int32), W[None])
    assert_ll(g.score(fr), want, "routed model under a global transform")
    g.set_cmllr(None, None)
    assert np.array_equal(g.score(fr), plain)
    base = str(tmp_path / "m")
    oracle.write_gk(base + ".gk", model[0], model[1])
    oracle.write_mc(base + ".mc", model[2], model[3], model[4])
    oracle.write_ph(base + ".ph", S)
    g1 = capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")
    n16 = g1.precision_states()[0]
    assert 0 < n16 < S
    ref = oracle.DiagModel(*oracle.read_gk(base + ".gk"), model[2], model[3], model[4]).score(fr.astype(np.float64))
    got1 = g1.score(fr)
    assert_ll(got1, ref, "routed model from files")
    cache = str(tmp_path / "m.cache")
    g1.write_cache(cache)
    g2 = capi.Gmm.from_cache(cache)
    assert g2.precision_states()[0] == n16 and np.array_equal(g2.score(fr), got1)
    for h in (g, g1, g2):
        h.close()
