"""End-to-end GPU parity of the whole path (BASELINE config 1): one 5-s 16 kHz
WAV -> MFCC chain -> 256-Gaussian diagonal HmmSet -> LNA, against the oracle's
restatement of phone_probs (aku/phone_probs.cc:145-267)."""
import os
import struct
import wave

import numpy as np
import pytest

from conftest import CODES_EQUAL_MIN, assert_lp_denormal_band, observed

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu


def _write_wav(path, pcm, rate=16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(pcm.astype("<i2").tobytes())


def _oracle_lna(oracle, ch, om, pcm, nbytes, normalize=True, start=0, end=None):
    n = ch.num_frames(len(pcm))
    stop = n if end is None else min(end, n)
    fea = ch.generate(pcm, start, stop - start)
    _, lik = om.score(fea, want_lik=True)
    lp, by = oracle.lna_encode(lik, normalize, nbytes)
    return lp, by


@pytest.fixture(scope="module")
def setup(capi, oracle, golden_dir):
    cfg = open(os.path.join(golden_dir, "mfcc_cms_norm.feaconf")).read()
    model = synth.make_model(D=39, G=256, S=32, comps=8)
    # move the Gaussians into the region real features occupy so that the
    # posteriors are not all at the floor
    return dict(cfg=cfg, model=model, ch=oracle.FeatureChain(cfg), om=oracle.DiagModel(*model),
                ft=capi.Feat(cfg), gm=capi.Gmm.from_arrays(*model))


def test_float_denormal_band_end_to_end(capi, oracle, setup):
    """The band the other end-to-end tests step around, checked for what the contract allows inside it.  The reference
    keeps the LINEAR state likelihood in a float before it normalises (aku/phone_probs.cc:224-236): for
    ln 2^-149 < ll < ln 2^-126 that float is a denormal, a whole number of quanta of 2^-149, and 1e-4 on ll can move the
    rounding by one quantum -- up to 0.69 in the logarithm at one quantum.  A model whose Gaussians lie 2.6x further out
    than the features puts a quarter of all (frame, state) values into that band: audio -> MFCC chain -> scoring ->
    normalised 4-byte LNA values in one call, every band value within one quantum of the oracle's float, everything
    outside it within 1e-4, in every scoring arithmetic."""
    mean, var, off, idx, w = synth.make_model(D=39, G=1024, S=128, comps=8, seed=517)
    model = (mean * 2.6, var, off, idx, w)
    om = oracle.DiagModel(*model)
    pcm = synth.make_audio(80000, seed=518)
    fea = setup["ch"].generate(pcm, 0, 623)
    ll_ref, lik = om.score(fea, want_lik=True)
    lp_ref, _ = oracle.lna_encode(lik, True, 4)
    band = (ll_ref > np.log(2.0 ** -149)) & (ll_ref < np.log(2.0 ** -126))
    assert band.mean() > 0.1, band.mean()
    gm = capi.Gmm.from_arrays(*model)
    for prec in (4, 3, 0):
        gm.set_precision(prec)
        data, frames = capi.run_utterance(setup["ft"], gm, pcm, lnabytes=4)
        assert frames == 623
        lp = np.frombuffer(data[5:], "<f4").reshape(623, 128)
        n_band = assert_lp_denormal_band(lp, lik, "precision %d" % prec)
        assert n_band >= int(band.sum())
        smooth = (ll_ref > -87.0) | (ll_ref < -104.5)
        assert np.abs(lp - lp_ref)[smooth].max() <= 1e-4, prec
    gm.close()


@pytest.mark.parametrize("nbytes", [2, 4])
def test_config1_utterance_lna(capi, oracle, setup, nbytes):
    pcm = synth.make_audio(80000)
    data, frames = capi.run_utterance(setup["ft"], setup["gm"], pcm, lnabytes=nbytes)
    assert frames == 623
    assert data[:5] == oracle.lna_header(32, nbytes)
    lp_ref, by_ref = _oracle_lna(oracle, setup["ch"], setup["om"], pcm, nbytes)
    body = np.frombuffer(data[5:], np.uint8).reshape(623, -1)
    if nbytes == 4:
        lp = body.view("<f4")
        # log-likelihoods within 1e-4 of the reference path (north_star); states
        # in the float-denormal band of the reference's float storage are exempt
        ll_ref = setup["om"].score(setup["ch"].generate(pcm, 0, 623))
        smooth = (ll_ref > -87.0) | (ll_ref < -104.5)
        assert np.abs(lp - lp_ref)[smooth].max() <= 1e-4
    else:
        code = body.reshape(623, 32, 2).astype(int)
        code = code[..., 0] * 256 + code[..., 1]
        cref = by_ref.reshape(623, 32, 2).astype(int)
        cref = cref[..., 0] * 256 + cref[..., 1]
        assert np.abs(code - cref).max() <= 1
        observed('pipeline config1 codes equal', float((code == cref).mean()), CODES_EQUAL_MIN)  # observed 0.9965: f32-class scores in front of the packing
    dec = oracle.lna_decode(data)
    assert dec.shape == (623, 32)


def test_frame_window_and_no_normalization(capi, oracle, setup):
    pcm = synth.make_audio(32000, seed=11)
    data, frames = capi.run_utterance(setup["ft"], setup["gm"], pcm, start_frame=10, end_frame=60,
                                      normalize=False, lnabytes=4)
    assert frames == 50
    lp = np.frombuffer(data[5:], "<f4").reshape(50, 32)
    lp_ref, _ = _oracle_lna(oracle, setup["ch"], setup["om"], pcm, 4, normalize=False, start=10, end=60)
    ll_ref = setup["om"].score(setup["ch"].generate(pcm, 10, 50))
    smooth = (ll_ref > -87.0) | (ll_ref < -104.5)
    assert np.abs(lp - lp_ref)[smooth].max() <= 1e-4


def test_recipe_run_and_batches(capi, oracle, setup, tmp_path):
    lens = [16000, 24000, 9506, 40000, 12345]
    lines = []
    pcms = []
    for i, n in enumerate(lens):
        pcm = synth.make_audio(n, seed=20 + i)
        pcms.append(pcm)
        _write_wav(str(tmp_path / ("u%d.wav" % i)), pcm)
        lines.append("audio=%s lna=%s" % (tmp_path / ("u%d.wav" % i), "u%d.lna" % i))
    recipe = str(tmp_path / "r.recipe")
    open(recipe, "w").write("# synthetic recipe\n" + "\n".join(lines) + "\n")
    out = str(tmp_path / "out")
    os.makedirs(out)
    st = capi.run_recipe(setup["ft"], setup["gm"], recipe, lnabytes=2, out_dir=out)
    assert st.utterances == 5
    assert st.frames == sum(setup["ch"].num_frames(n) for n in lens)
    whole = {}
    for i, pcm in enumerate(pcms):
        data = open(os.path.join(out, "u%d.lna" % i), "rb").read()
        single, _ = capi.run_utterance(setup["ft"], setup["gm"], pcm, lnabytes=2)
        assert data == single           # blocked run == per-utterance run, byte for byte
        whole[i] = data
    # two "ranks" with the reference's -B 2 -I k slicing reproduce the same files
    out2 = str(tmp_path / "out2")
    os.makedirs(out2)
    s1 = capi.run_recipe(setup["ft"], setup["gm"], recipe, num_batches=2, batch_index=1, out_dir=out2)
    s2 = capi.run_recipe(setup["ft"], setup["gm"], recipe, num_batches=2, batch_index=2, out_dir=out2)
    assert (s1.utterances, s2.utterances) == (3, 2)
    for i in range(5):
        assert open(os.path.join(out2, "u%d.lna" % i), "rb").read() == whole[i]
    # --no-overwrite skips existing outputs
    s3 = capi.run_recipe(setup["ft"], setup["gm"], recipe, no_overwrite=True, out_dir=out)
    assert s3.utterances == 0


def test_recipe_upload_ring_and_its_fallback_write_the_same_files(capi, setup, tmp_path):
    """The recipe driver's reader copies every file's samples into a pinned upload ring (asynchronous uploads, one copy
    per run of utterances); a file that finds no room keeps its pageable buffer.  With the ring cut down to two seconds of
    audio (a diagnostic entry point) a recipe of 40 files of 0.3-2.4 s -- one block, so nothing is returned to the ring
    before the end -- goes partly through the ring (incl. its wrap-free placement) and partly around it, and a ring smaller
    than every file takes none: the LNA files are the same bytes as with the whole ring (aku/phone_probs.cc:145-267 has
    one way to read a file)."""
    import ctypes
    L = capi.lib()
    L.aasr_debug_set_upload_ring_samples.argtypes = [ctypes.c_int64]
    L.aasr_debug_set_upload_ring_samples.restype = None
    rng = np.random.default_rng(5)
    lines = []
    for i in range(40):
        pcm = synth.make_audio(int(rng.integers(4800, 38400)), seed=300 + i)
        _write_wav(str(tmp_path / ("v%d.wav" % i)), pcm)
        lines.append("audio=%s lna=%s" % (tmp_path / ("v%d.wav" % i), "v%d.lna" % i))
    recipe = str(tmp_path / "v.recipe")
    open(recipe, "w").write("\n".join(lines) + "\n")
    outs = {}
    try:
        for name, limit in (("whole", 0), ("two_seconds", 32000 * 4), ("none", 1000)):   # a file takes at most a quarter
            L.aasr_debug_set_upload_ring_samples(limit)
            out = str(tmp_path / name)
            os.makedirs(out)
            st = capi.run_recipe(setup["ft"], setup["gm"], recipe, lnabytes=2, out_dir=out)
            assert st.utterances == 40
            outs[name] = [open(os.path.join(out, "v%d.lna" % i), "rb").read() for i in range(40)]
    finally:
        L.aasr_debug_set_upload_ring_samples(0)
    assert outs["two_seconds"] == outs["whole"] and outs["none"] == outs["whole"]
    assert len(outs["whole"][0]) > 1000


def test_dimension_mismatch_is_reported(capi, setup):
    g13 = capi.Gmm.from_arrays(*synth.make_model(D=13, G=16, S=2, comps=8))
    with pytest.raises(capi.AasrError, match="Gaussian dimension is 13 but feature dimension is 39"):
        capi.run_utterance(setup["ft"], g13, synth.make_audio(8000))


def test_pitched_scores_and_lna_equal_the_dense_path(capi):
    """aasr_gmm_score_dev_pitched / aasr_lna_encode_dev_pitched (score rows padded to whole cache
    lines, what the recipe driver and the full-chain bench keep on the device) give the bits of the
    dense entry points, for the 4-wave and the 8-wave kernel."""
    import torch
    from aaltoasr_amd import synth
    model = synth.make_model(D=39, G=808, S=101, comps=8, seed=9)
    g = capi.Gmm.from_arrays(*model)
    assert g.score_pitch_ok()
    for F in (700, 9001):
        fr = torch.from_numpy(synth.make_frames(F, seed=F)).cuda()
        dense = torch.empty((F, 101), dtype=torch.float32, device="cuda")
        g.score_dev(fr, dense)
        for pitch in (128, 103, 112):        # whole 128-byte lines, odd, half lines
            padded = torch.full((F, pitch), -7.0, dtype=torch.float32, device="cuda")
            g.score_dev_pitched(fr, padded, pitch)
            torch.cuda.synchronize()
            assert torch.equal(padded[:, :101], dense) and bool((padded[:, 101:] == -7.0).all())
        by_d = torch.empty((F, 202), dtype=torch.uint8, device="cuda")
        by_p = torch.empty((F, 202), dtype=torch.uint8, device="cuda")
        capi.lna_encode_dev(dense, True, 2, None, by_d)
        capi.lna_encode_dev(padded, True, 2, None, by_p, num_states=101)
        torch.cuda.synchronize()
        assert torch.equal(by_d, by_p)
    g.set_precision(0)                       # the f32 track kernel takes a pitch too
    assert g.score_pitch_ok()
    g.score_dev(fr, dense)
    padded = torch.full((F, 128), -7.0, dtype=torch.float32, device="cuda")
    g.score_dev_pitched(fr, padded, 128)
    torch.cuda.synchronize()
    assert torch.equal(padded[:, :101], dense) and bool((padded[:, 101:] == -7.0).all())
    g.set_precision(2)                       # the centred kernel writes dense rows only
    assert not g.score_pitch_ok()
    with pytest.raises(capi.AasrError):
        g.score_dev_pitched(fr, padded, 128)


def test_recipe_segment_times_use_float_frame_limits(capi, oracle, setup, tmp_path):
    """start-time=2.008 end-time=8.008 at 125 frames/s: the reference multiplies float by float
    (aku/phone_probs.cc:199-206, aku/Recipe.hh:48-49) and writes frames 250..1000; double
    arithmetic would give 251..999."""
    pcm = synth.make_audio(160000, seed=5)
    _write_wav(str(tmp_path / "seg.wav"), pcm)
    recipe = str(tmp_path / "seg.recipe")
    open(recipe, "w").write("audio=%s lna=seg.lna start-time=2.008 end-time=8.008\n" % (tmp_path / "seg.wav"))
    st = capi.run_recipe(setup["ft"], setup["gm"], recipe, lnabytes=4, out_dir=str(tmp_path))
    info = oracle.recipe_read(open(recipe).read())[0]
    start, end = oracle.recipe_frame_limits(info, setup["ch"].frame_rate)
    assert (start, end) == (250, 1001) and st.frames == 751
    data = open(tmp_path / "seg.lna", "rb").read()
    single, n = capi.run_utterance(setup["ft"], setup["gm"], pcm, start_frame=start, end_frame=end, lnabytes=4)
    assert n == 751 and data == single
    lp = np.frombuffer(data[5:], "<f4").reshape(751, 32)
    lp_ref, _ = _oracle_lna(oracle, setup["ch"], setup["om"], pcm, 4, start=start, end=end)
    ll_ref = setup["om"].score(setup["ch"].generate(pcm, start, end - start))
    smooth = (ll_ref > -87.0) | (ll_ref < -104.5)
    assert np.abs(lp - lp_ref)[smooth].max() <= 1e-4
