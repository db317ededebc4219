"""GPU parity: aasr_gmm_score / aasr_gmm_gauss_loglik (HIP, through the C ABI)
against the CPU oracle (oracle/aasr_oracle.c restating HmmSet.cc:484-501,
Distributions.cc:1040-1062, 2078-2086).  Tolerance 1e-4 absolute on natural
log-likelihoods (BASELINE.json north_star)."""
import numpy as np
import pytest

from conftest import assert_ll

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-4


# every scoring kernel is exercised: the launcher's own choice (7), the
# independent-track layout (2), the general LDS-staged MFMA kernel (0) and the
# centred-form vector kernel (4)
LAYOUTS = [7, 2, 0, 4]


def _check(capi, oracle, model, frames, layouts=LAYOUTS):
    mean, var, off, idx, w = model
    ref = oracle.DiagModel(mean, var, off, idx, w).score(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    g.set_precision(0)      # the f32 kernels first (the default is the bf16x3 split)
    worst = 0.0
    used = set()
    for mask in layouts:
        g.set_layouts(mask)
        used.add(g.active_layout())
        got = g.score(frames)
        worst = max(worst, assert_ll(got, ref, "layout %d (kernel %d)" % (mask, g.active_layout())))
    # the split-operand kernels (three bf16 terms; two fp16 terms where the model is eligible, else the
    # setting falls back to bf16x3) on whichever track layout the model has
    g.set_layouts(7)
    if g.active_layout() in (1, 2):
        for prec in (3, 4):
            g.set_precision(prec)
            got = g.score(frames)
            worst = max(worst, assert_ll(got, ref, "precision %d (runs as %d) on layout %d" % (
                prec, g.effective_precision(), g.active_layout())))
        g.set_precision(0)
    g.close()
    return worst, used


@pytest.mark.parametrize("F", [1, 63, 64, 65, 255, 256, 257, 1000])
def test_config1_shape_ragged_frames(capi, oracle, F):
    """BASELINE config 1 model: D=39, G=256, S=32x8; ragged frame counts."""
    model = synth.make_model(D=39, G=256, S=32, comps=8)
    _check(capi, oracle, model, synth.make_frames(F))


def test_sixteen_components_disjoint(capi, oracle):
    """cfg-2 layout in miniature: 16 comps/state, continuous-density pool."""
    model = synth.make_model(D=39, G=2048, S=128, comps=16)
    _check(capi, oracle, model, synth.make_frames(300))


def test_tied_pool(capi, oracle):
    """Mixtures point at arbitrary (shared) pool Gaussians."""
    model = synth.make_model(D=39, G=512, S=96, comps=16, tied=True)
    _check(capi, oracle, model, synth.make_frames(200))


def test_variable_components_and_long_states(capi, oracle):
    """n_s from 1 to 150: segments span 32-row chunks and 64-row tiles."""
    model = synth.make_model(D=39, G=8192, S=64, comps_range=(1, 150), seed=5, tied=True)
    _check(capi, oracle, model, synth.make_frames(130))


@pytest.mark.parametrize("D", [1, 7, 8, 13, 25, 26, 39, 40, 47, 63])
def test_dimensions(capi, oracle, D):
    model = synth.make_model(D=D, G=96, S=12, comps=8, seed=D)
    _check(capi, oracle, model, synth.make_frames(70, D=D, seed=100 + D))


def test_empty_state_and_zero_weight(capi, oracle):
    mean, var, off, idx, w = synth.make_model(D=13, G=64, S=8, comps=8, seed=3)
    # state 2 loses all components; one component of state 5 gets weight 0
    n = np.diff(off)
    n[2] = 0
    keep = np.ones(len(idx), bool)
    keep[off[2]:off[3]] = False
    idx, w = idx[keep], w[keep]
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
    w[off[5]] = 0.0
    model = (mean, var, off, idx, w)
    frames = synth.make_frames(40, D=13)
    _check(capi, oracle, model, frames)
    g = capi.Gmm.from_arrays(*model)
    got = g.score(frames)
    assert np.allclose(got[:, 2], np.log(1e-50), atol=1e-5)


def test_floor_and_far_frames(capi, oracle):
    """Frames far from every Gaussian: state likelihood < 1e-50 clamps to
    log(1e-50) (HmmSet.cc:497-498); mid-range values stay within tolerance."""
    model = synth.make_model(D=39, G=256, S=32, comps=8)
    fr = synth.make_frames(64)
    fr[:16] *= 3.0
    fr[16:32] *= 6.0
    fr[32:48] += 4.0
    _check(capi, oracle, model, fr)


def test_zero_variance_dimension(capi, oracle):
    """var <= 0 -> precision 0 -> 'invalid' Gaussian with constant 0
    (Distributions.cc:1144-1147, 1276-1287)."""
    mean, var, off, idx, w = synth.make_model(D=8, G=32, S=4, comps=8, seed=9)
    var[3, 2] = 0.0
    var[7, :] = -1.0
    _check(capi, oracle, (mean, var, off, idx, w), synth.make_frames(50, D=8))


def test_gauss_loglik_pool_view(capi, oracle):
    mean, var, off, idx, w = synth.make_model(D=39, G=200, S=25, comps=8)
    frames = synth.make_frames(90)
    ref = oracle.DiagModel(mean, var, off, idx, w).gauss_loglik(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    got = g.gauss_loglik(frames)
    assert np.abs(got - ref).max() <= TOL


def test_model_files_roundtrip(capi, oracle, tmp_path):
    mean, var, off, idx, w = synth.make_model(D=13, G=64, S=8, comps=8, seed=11)
    base = str(tmp_path / "m")
    oracle.write_gk(base + ".gk", mean, var)
    oracle.write_mc(base + ".mc", off, idx, w)
    oracle.write_ph(base + ".ph", 8)
    g = capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")
    assert (g.dim, g.num_states, g.num_gaussians) == (13, 8, 64)
    frames = synth.make_frames(33, D=13)
    ref = oracle.read_model(base).score(frames.astype(np.float64))
    assert np.abs(g.score(frames) - ref).max() <= TOL


def test_layout_selection(capi):
    """Uniform models take the grouped-track kernel, ragged ones the independent
    tracks, models whose peak likelihood leaves no exponent headroom the general
    kernel."""
    g = capi.Gmm.from_arrays(*synth.make_model(D=39, G=2048, S=128, comps=16))
    assert g.active_layout() == 1
    g = capi.Gmm.from_arrays(*synth.make_model(D=39, G=8192, S=64, comps_range=(1, 150), seed=5, tied=True))
    assert g.active_layout() == 2
    mean, var, off, idx, w = synth.make_model(D=39, G=64, S=8, comps=8)
    g = capi.Gmm.from_arrays(mean, var * 1e-4, off, idx, w)   # sigma ~ 0.01: ill-conditioned
    assert g.active_layout() == 4


def test_ill_conditioned_model_takes_the_centred_kernel(capi, oracle):
    """sigma ~ 0.03 with means ~ N(0,1): the expanded (GEMM) form would lose
    ~1e-2 to cancellation; the launcher must pick the centred form by itself."""
    mean, var, off, idx, w = synth.make_model(D=39, G=64, S=8, comps=8, seed=21)
    var = var * 1e-3
    frames = (mean[np.arange(40) % 64] + 0.03 * synth.make_frames(40, seed=5)).astype(np.float32)
    worst, used = _check(capi, oracle, (mean, var, off, idx, w), frames, layouts=[7, 4])
    assert used == {4}


def test_centred_kernel_keeps_double_means(capi, oracle):
    """Means arrive as doubles (a .gk file's digits); rounded to one float a mean costs
    p |x - mu| ulp(mu)/2 -- 1e-4 at 14 sigma from a sigma = 0.012 Gaussian around 2.3 (fuzz_parity
    seed 811) -- so the centred records carry mu as a float pair and only roundings relative to
    x - mu are left."""
    rng = np.random.default_rng(811)
    D, G, S = 2, 96, 12
    mean = rng.standard_normal((G, D)) * 2.0 + np.array([2.0, -3.0])     # not float32-representable
    var = np.exp(rng.uniform(np.log(5e-5), np.log(5e-4), (G, D)))
    off = np.arange(0, G + 1, G // S).astype(np.int32)
    idx = np.arange(G, dtype=np.int32)
    w = rng.uniform(0.1, 1.0, G)
    # frames 8..15 sigma away from one Gaussian each
    pick = rng.integers(0, G, 600)
    frames = (mean[pick] + np.sqrt(var[pick]) * rng.uniform(8, 15, (600, 1)) * rng.choice([-1.0, 1.0], (600, D))
              / np.sqrt(D)).astype(np.float32)
    want = oracle.DiagModel(mean, var, off, idx, w).score(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    assert g.active_layout() == 4
    got = g.score(frames)
    vis = want > -103.97
    assert vis.sum() > 300
    err = np.abs(got - want)[vis].max()
    # with float means the same comparison gives ~2e-4 (the oracle on rounded means differs by that much)
    rounded = oracle.DiagModel(mean.astype(np.float32).astype(np.float64), var, off, idx, w).score(frames.astype(np.float64))
    assert np.abs(rounded - want)[vis].max() > 1e-4
    assert err <= 6e-5, err


def test_block_partition_invariance(capi):
    """Scoring is per-frame: any split of the frame block gives identical bits."""
    model = synth.make_model(D=39, G=256, S=32, comps=8)
    g = capi.Gmm.from_arrays(*model)
    fr = synth.make_frames(700)
    whole = g.score(fr)
    parts = np.vstack([g.score(fr[:1]), g.score(fr[1:300]), g.score(fr[300:])])
    assert np.array_equal(whole.view(np.uint32), parts.view(np.uint32))


def test_device_pointer_entry(capi, oracle):
    import torch
    model = synth.make_model(D=39, G=256, S=32, comps=8)
    g = capi.Gmm.from_arrays(*model)
    fr = synth.make_frames(513)
    d_fr = torch.from_numpy(fr).cuda()
    d_out = torch.empty((513, 32), dtype=torch.float32, device="cuda")
    g.score_dev(d_fr, d_out)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint32), g.score(fr).view(np.uint32))


def test_model_cache_round_trip(capi, oracle, tmp_path):
    """aasr_gmm_write_cache / create_from_cache: same scores bit for bit, HMM inventory
    kept, corruption detected."""
    mean, var, off, idx, w = synth.make_model(D=20, G=300, S=25, tied=True, comps_range=(1, 17))
    base = str(tmp_path / "m")
    oracle.write_gk(base + ".gk", mean, var)
    oracle.write_mc(base + ".mc", off, idx, w)
    oracle.write_ph(base + ".ph", 25)
    g1 = capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")
    cache = str(tmp_path / "m.aasr")
    g1.write_cache(cache)
    g2 = capi.Gmm.from_cache(cache)
    frames = synth.make_frames(200, D=20)
    assert (g2.dim, g2.num_states, g2.num_gaussians) == (20, 25, 300)
    assert np.array_equal(g1.score(frames), g2.score(frames))
    g2.write_cache(str(tmp_path / "again.aasr"))          # a cache of a cache is the same file
    assert open(cache, "rb").read() == open(tmp_path / "again.aasr", "rb").read()
    blob = bytearray(open(cache, "rb").read())
    blob[len(blob) // 2] ^= 0x40
    open(tmp_path / "bad.aasr", "wb").write(bytes(blob))
    with pytest.raises(capi.AasrError, match="checksum mismatch"):
        capi.Gmm.from_cache(str(tmp_path / "bad.aasr"))
    with pytest.raises(capi.AasrError, match="not a model cache"):
        capi.Gmm.from_cache(base + ".gk")
    # the cache remembers the text files it was written from: it is refused once they change
    # (retraining, another -b next to the same cache path) and for a model built in memory
    g3 = capi.Gmm.from_cache_checked(cache, base + ".gk", base + ".mc", base + ".ph")
    assert np.array_equal(g1.score(frames), g3.score(frames))
    oracle.write_gk(base + ".gk", mean + 0.25, var)
    with pytest.raises(capi.AasrError, match="stale model cache"):
        capi.Gmm.from_cache_checked(cache, base + ".gk", base + ".mc", base + ".ph")
    capi.Gmm.from_arrays(mean, var, off, idx, w).write_cache(str(tmp_path / "mem.aasr"))
    with pytest.raises(capi.AasrError, match="stale model cache"):
        capi.Gmm.from_cache_checked(str(tmp_path / "mem.aasr"), base + ".gk", base + ".mc", base + ".ph")
    # index tables of a foreign file are validated, not trusted (checksum recomputed here)
    good = bytearray(open(cache, "rb").read())
    hdr = 8 + 4 + 4 + 8 * 3 + 4 + 8 + 6 * 8
    pos = hdr + 2 * 300 * 20 * 8 + 4          # mix_off[1]
    good[pos:pos + 4] = (10 ** 6).to_bytes(4, "little")
    h = 1469598103934665603
    for byte in good[:-8]:
        h = ((h ^ byte) * 1099511628211) & (2 ** 64 - 1)
    good[-8:] = h.to_bytes(8, "little")
    open(tmp_path / "crafted.aasr", "wb").write(bytes(good))
    with pytest.raises(capi.AasrError, match="inconsistent mixture tables"):
        capi.Gmm.from_cache(str(tmp_path / "crafted.aasr"))


def test_outlier_routing_keeps_the_model_on_the_matrix_path(capi, oracle):
    """A few Gaussians with sigma ~ 0.03 (kappa >> 600) in an otherwise well-conditioned model: they
    are taken out of the matrix layouts and scored in the centred form, merged per state; the model
    as a whole stays on the track kernels (both precisions), ragged / empty states included."""
    import ctypes as C
    rng = np.random.default_rng(31)
    mean, var, off, idx, w = synth.make_model(D=39, G=512, S=48, comps=8, seed=13)
    bad = rng.choice(512, 24, replace=False)
    var[bad] *= 2e-3
    # a state made of outliers only, one with a single outlier, the rest mixed by the draw
    idx[off[5]:off[6]] = bad[:off[6] - off[5]]
    idx[off[9]] = bad[3]
    frames = synth.make_frames(300, seed=8)
    # some frames close to the outliers so that their terms dominate
    frames[:24] = (mean[bad] + np.sqrt(var[bad]) * rng.standard_normal((24, 39))).astype(np.float32)
    ref = oracle.DiagModel(mean, var, off, idx, w).score(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    L = capi.lib()
    L.aasr_debug_kappa.restype = C.c_double
    L.aasr_debug_kappa.argtypes = [C.c_void_p]
    assert L.aasr_debug_kappa(g._h) > 600
    assert g.active_layout() in (1, 2)                 # not the centred kernel
    for prec in (4, 3, 0):
        g.set_precision(prec)
        for mask in (7, 2, 0):
            g.set_layouts(mask)
            got = g.score(frames)
            err = np.abs(got - ref)
            assert err.max() <= 1e-4, "prec %d layout %d: max |dll| %.3g at %s" % (
                prec, mask, err.max(), np.unravel_index(err.argmax(), err.shape))
    g.set_layouts(4)                                   # the whole model in the centred form agrees too
    assert_ll(g.score(frames), ref, "whole model in the centred form")
    g.close()
    # a majority of outliers: no routing, the centred kernel takes the model
    var2 = var.copy()
    var2[rng.choice(512, 300, replace=False)] *= 2e-3
    g2 = capi.Gmm.from_arrays(mean, var2, off, idx, w)
    assert g2.active_layout() == 4
    ref2 = oracle.DiagModel(mean, var2, off, idx, w).score(frames.astype(np.float64))
    assert_ll(g2.score(frames), ref2, "majority of outliers")
    g2.close()


def test_wide_and_narrow_workgroups_give_identical_bits(capi):
    """Batches of 8192+ frames take the 8-wave bf16x3 kernel, smaller ones the 4-wave one; a file's
    scores must not depend on how the recipe driver happened to batch it."""
    model = synth.make_model(D=39, G=1024, S=77, comps=12, seed=3)
    g = capi.Gmm.from_arrays(*model)
    fr = synth.make_frames(9000, seed=12)
    for prec in (3, 4):
        g.set_precision(prec)
        assert g.effective_precision() == prec
        whole = g.score(fr)                                   # 8-wave workgroups
        parts = np.vstack([g.score(fr[:4000]), g.score(fr[4000:8100]), g.score(fr[8100:])])   # 4-wave
        assert np.array_equal(whole.view(np.uint32), parts.view(np.uint32))
    g.close()


@pytest.mark.parametrize("D", [1, 3, 7, 13, 20, 24, 31, 40, 47, 55, 63])
def test_wide_kernel_every_instance(capi, oracle, D):
    """Every K/16 instance of the 8-wave kernel (batches of 8192+ frames), grouped and independent
    track layouts, against the oracle and bit-for-bit against the 4-wave kernel (smaller batches);
    dimension 63 has no 8-wave form (LDS) and must still agree."""
    rng = np.random.default_rng(100 + D)
    for ragged in (False, True):
        S = 9
        n = rng.integers(1, 7, S) if ragged else np.full(S, 4)
        off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
        G = int(off[-1])
        mean = rng.standard_normal((G, D))
        var = np.exp(rng.uniform(np.log(0.3), np.log(3.0), (G, D)))
        idx = np.arange(G, dtype=np.int32)
        w = rng.uniform(0.1, 1.0, G)
        g = capi.Gmm.from_arrays(mean, var, off, idx, w)
        fr = synth.make_frames(8200, D=D, seed=D)
        ref = oracle.DiagModel(mean, var, off, idx, w).score(fr[:300].astype(np.float64))
        for prec in (3, 4):
            g.set_precision(prec)
            whole = g.score(fr)
            parts = np.vstack([g.score(fr[:4100]), g.score(fr[4100:])])
            assert np.array_equal(whole.view(np.uint32), parts.view(np.uint32))
            assert np.abs(whole[:300] - ref).max() <= 1e-4, (prec, g.effective_precision())
        g.close()


def test_gauss_loglik_of_tight_gaussians(capi, oracle):
    """PDFPool::compute_likelihood(f, index) (aku/Distributions.hh:145) for variance-floored Gaussians:
    the expanded (matrix) form loses ~eps * kappa there, so per-Gaussian values of an ill-conditioned
    or outlier-routed pool come from the centred kernel as well -- no 1e-50 floor on this view."""
    rng = np.random.default_rng(4)
    mean, var, off, idx, w = synth.make_model(D=39, G=256, S=32, comps=8, seed=3)
    bad = rng.choice(256, 12, replace=False)
    var[bad] *= 2e-3
    frames = synth.make_frames(200, seed=2)
    frames[:12] = (mean[bad] + np.sqrt(var[bad]) * rng.standard_normal((12, 39))).astype(np.float32)
    ref = oracle.DiagModel(mean, var, off, idx, w).gauss_loglik(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    got = g.gauss_loglik(frames)
    # relative to the magnitude: values reach -1e6 for the tight Gaussians far from a frame
    err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    assert err.max() <= 2e-6, err.max()
    near = ref > -150
    assert np.abs(got - ref)[near].max() <= 1e-4, np.abs(got - ref)[near].max()
    assert near[:12, bad].diagonal().all()          # the frames drawn from the tight Gaussians see them


def test_f16x2_is_chosen_by_conditioning_and_clamps_far_frames(capi, oracle):
    """AASR_PREC_F16X2 (the default): the two-term fp16 form runs only for models whose conditioning estimate is
    below its own, tighter limits -- others fall back to the three-term bf16 form under the same setting -- and a
    frame beyond the fp16 clamp (|x - pivot| > 240, its square would overflow) comes out at the floor as in the oracle."""
    import ctypes as C
    L = capi.lib()
    L.aasr_debug_kappa.restype = C.c_double
    L.aasr_debug_kappa.argtypes = [C.c_void_p]
    model = synth.make_model(D=39, G=512, S=32, comps=16, seed=71)
    g = capi.Gmm.from_arrays(*model)
    assert g.effective_precision() == 4 and L.aasr_debug_kappa(g._h) < 330
    fr = synth.make_frames(200, seed=72)
    fr[5, 7] = 300.0
    fr[6, :] = -1000.0
    fr[7, 38] = 239.0
    ref = oracle.DiagModel(*model).score(fr.astype(np.float64))
    assert_ll(g.score(fr), ref, "f16x2 with frames beyond the clamp")
    assert np.all(g.score(fr)[5:7] == np.float32(np.log(1e-50)))
    g.close()
    # tighter Gaussians: kappa between the f16x2 and the bf16x3 limits -> same setting, bf16x3 kernel
    mean, var, off, idx, w = synth.make_model(D=39, G=512, S=32, comps=16, seed=71, var_lo=0.08, var_hi=0.5)
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    assert 330 < L.aasr_debug_kappa(g._h) < 600 and g.active_layout() in (1, 2)
    # ... for the states that hold such a Gaussian (per-state precision routing, tests/test_mixed_gpu.py): here nearly all
    n16 = g.precision_states()[0]
    assert n16 < 8 and g.effective_precision() == (4 if n16 else 3)
    fr = synth.make_frames(200, seed=73)
    assert_ll(g.score(fr), oracle.DiagModel(mean, var, off, idx, w).score(fr.astype(np.float64)), "fallback")
    g.close()


@pytest.mark.parametrize("D", [64, 65, 80, 100, 126, 127, 200])
def test_dimensions_beyond_63(capi, oracle, D):
    """Feature dimension > 63 (the reference reads any `dim`, aku/Distributions.cc:1131-1150; ConcatModule,
    aku/FeatureModules.cc:1461-1501, produces wide vectors): the diagonal density factorises over the dimensions,
    so the model is scored as parts of <= 63 dimensions and the parts are added per mixture component."""
    rng = np.random.default_rng(300 + D)
    mean, var, off, idx, w = synth.make_model(D=D, G=160, S=14, comps_range=(0, 14), seed=D, tied=True)
    w[off[3]] = 0.0                                            # a zero-weight component
    frames = synth.make_frames(90, D=D, seed=400 + D)
    frames[:8] = (mean[idx[:8]] + 0.3 * rng.standard_normal((8, D))).astype(np.float32)   # some frames on Gaussians
    frames[8:12] *= 4.0                                        # some far out (the 1e-50 floor)
    om = oracle.DiagModel(mean, var, off, idx, w)
    ref = om.score(frames.astype(np.float64))
    g = capi.Gmm.from_arrays(mean, var, off, idx, w)
    for prec in (4, 3, 0):
        g.set_precision(prec)
        assert_ll(g.score(frames), ref, "D = %d, precision %d" % (D, prec))
    # per-Gaussian log-likelihoods: the parts added, no floor
    gl = g.gauss_loglik(frames[:20])
    want = om.gauss_loglik(frames[:20].astype(np.float64))
    vis = want > -200
    assert np.abs(gl - want)[vis].max() <= 1e-4 * np.maximum(1.0, np.abs(want[vis]) / 100).max()
    # Gaussian clustering on the wide model: centres from the vector kernel, the selection masks applied where the
    # parts are added per component, log-domain merge; scores and exact-evaluation counts as the oracle's
    if D <= 200:
        g.set_precision(4)
        g2c = synth.make_clustering(mean, 12, seed=3)
        g2c[::19] = -1
        pairs = [(int(i), int(c)) for i, c in enumerate(g2c) if c >= 0]
        for minc, ming in ((0.0, 0.25), (0.3, 0.0), (1.0, 1.0)):
            om.set_clustering(12, pairs, minc, ming)
            want_c, want_n = om.score_clustered(frames.astype(np.float64), want_counts=True)
            g.set_clustering(12, pairs)
            g.set_clustering_min_evals(minc, ming)
            got_c = g.score(frames)
            assert np.array_equal(g.cluster_exact_counts(len(frames)), want_n), (D, minc, ming)
            assert_ll(got_c, want_c, "D = %d clustered (%g, %g)" % (D, minc, ming))
        g.set_clustering(0, [])
    # AASR_PREC_F64 (the reference's arithmetic in double) has instances up to 192 dimensions
    if D <= 192:
        g.set_precision(1)
        got = g.score(frames)
        vis = ref > np.log(1e-50)
        assert np.abs(got - ref)[vis].max() <= 2e-6 * np.maximum(1.0, np.abs(ref[vis])).max()   # rounded to float once
        assert_ll(got, ref, "D = %d in double" % D)
        got64 = g.score_f64(frames.astype(np.float64))
        assert np.abs(got64 - ref).max() <= 1e-9 * np.maximum(1.0, np.abs(ref)).max()
    else:
        with pytest.raises(capi.AasrError):
            g.set_precision(1)
    g.close()
