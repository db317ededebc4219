"""tests/golden/oracle_vectors.npz (written by tools/make_golden.py): the oracle must
still reproduce its committed outputs (CPU), and the engine must match the committed
outputs directly (GPU) -- a regression pin for the parts of the path that have no
golden vectors in the reference."""
import os

import numpy as np
import pytest

from conftest import CODES_EQUAL_MIN, observed

from aaltoasr_amd import synth


@pytest.fixture(scope="module")
def vec(golden_dir):
    return np.load(os.path.join(golden_dir, "oracle_vectors.npz"))


def test_oracle_reproduces_committed_vectors(vec):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(root, "tools", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    now = mod.vectors()
    assert sorted(now) == sorted(vec.files)
    for k in vec.files:
        if now[k].dtype.kind in "iu":
            assert np.array_equal(now[k], vec[k]), k
        else:
            assert np.allclose(now[k], vec[k], rtol=0, atol=1e-9), k   # libm may differ by an ulp


@pytest.mark.gpu
def test_engine_matches_committed_vectors(capi, vec, golden_dir, oracle):
    model = synth.make_model(D=39, G=256, S=32, comps=8)
    frames = synth.make_frames(200)
    frames[::25] *= 6.0
    g = capi.Gmm.from_arrays(*model)
    ll = g.score(frames)
    assert np.abs(ll - vec["cfg0_state_loglik"]).max() <= 1e-4
    assert np.abs(g.gauss_loglik(frames[1:4]) - vec["cfg0_gauss_loglik_f1_4"]).max() <= 1e-4
    for nb in (2, 4):
        lp, by = capi.lna_encode(ll, True, nb)
        ok = vec["cfg0_state_loglik"] > -85
        assert np.abs(lp - vec["cfg0_lna%d_lp" % nb])[ok].max() <= 1e-4
        if nb == 2:
            code = by.reshape(200, 32, 2).astype(np.int32)
            ref = vec["cfg0_lna2_bytes"].reshape(200, 32, 2).astype(np.int32)
            d = np.abs((code[..., 0] * 256 + code[..., 1]) - (ref[..., 0] * 256 + ref[..., 1]))[ok]
            assert d.max() <= 1
            observed('golden_vectors codes equal', float((d == 0).mean()), CODES_EQUAL_MIN)  # observed 0.9917
    g = capi.Gmm.from_arrays(*synth.make_model(D=39, G=2048, S=128, tied=True, comps_range=(1, 23)))
    assert np.abs(g.score(synth.make_frames(120, seed=77)) - vec["tied_state_loglik"]).max() <= 1e-4
    model = synth.make_model(D=39, G=2048, S=128, comps=16)
    g = capi.Gmm.from_arrays(*model)
    g.set_clustering(64, [(i, int(c)) for i, c in enumerate(vec["cluster_g2c"])])
    g.set_clustering_min_evals(0.0, 0.25)
    fr = synth.make_frames(150, seed=78)
    assert np.abs(g.score(fr) - vec["cluster_state_loglik"]).max() <= 1e-4
    assert np.array_equal(g.cluster_exact_counts(150), vec["cluster_exact_counts"])
    rng = np.random.default_rng(synth.SEED + 5)
    D, G = 8, 64
    mean = rng.standard_normal((G, D))
    a = rng.standard_normal((G, D, D)) * 0.4
    cov = a @ a.transpose(0, 2, 1) + 0.1 * np.eye(D)
    _, _, off, idx, w = synth.make_model(D=D, G=G, S=8, comps=8)
    gf = capi.Gmm.from_full(mean, cov, off, idx, w)
    assert np.abs(gf.score(synth.make_frames(100, D=D, seed=79)) - vec["full_state_loglik"]).max() <= 1e-4
    pcm, _ = oracle.read_wav_pcm16(os.path.join(golden_dir, "short.wav"))
    for name in ("mfcc_p_dd", "mfcc_cms_norm"):
        ft = capi.Feat(open(os.path.join(golden_dir, name + ".feaconf")).read())
        assert np.abs(ft.run(pcm, -15, 106, dtype=np.float64) - vec["fea_" + name]).max() <= 5e-6
