"""GPU parity of the feature chain (aasr_feat_*, HIP kernels through the C ABI)
against the CPU oracle and the reference's golden files.

Tolerances: the device restates every float32/float64 island of the reference
operation by operation (tables come from host libm) -- including logf, which is
glibc's own algorithm on the device (csrc/feat_kernels.hip glibc_logf; the oracle
calls the host's) -- so the modules up to and including mel / dct / merge agree
bit for bit and the rest to a few 1e-14 (summation orders in double: the mean
subtractor's window sums).  FEAT_TOL is what remains as the bound; VTLN variants,
sr_norm and quanteq (double pow / sinc tables) keep looser ones below.  The
reference's own golden
files only resolve 0.005."""
import os

import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu

FEAT_TOL = 1e-10


def _cfg(golden_dir, name):
    return open(os.path.join(golden_dir, name + ".feaconf")).read()


@pytest.fixture(scope="module")
def short_wav(oracle, golden_dir):
    pcm, _ = oracle.read_wav_pcm16(os.path.join(golden_dir, "short.wav"))
    return pcm


@pytest.mark.parametrize("name,lo,hi", [("mfcc_p_dd", -10, 80), ("mfcc_cms_norm", -15, 90)])
def test_reference_goldens(capi, golden_dir, short_wav, name, lo, hi):
    """aku/tests/{mfcc_p_dd,mfcc_cms_norm}.script incl. negative and post-EOF frames."""
    ref = np.loadtxt(os.path.join(golden_dir, name + ".ref"))
    ft = capi.Feat(_cfg(golden_dir, name))
    assert ft.dim == 39 and ft.last_frame(len(short_wav)) == 72
    n = hi - lo + 1
    got = ft.run(short_wav, lo, n, dtype=np.float64)
    assert np.abs(got - ref[:n]).max() <= 0.005 + 1e-9


@pytest.mark.parametrize("name", ["mfcc_p_dd", "mfcc_cms_norm"])
def test_every_module_matches_oracle(capi, oracle, golden_dir, short_wav, name):
    cfg = _cfg(golden_dir, name)
    ch = oracle.FeatureChain(cfg)
    ft = capi.Feat(cfg)
    for m in ch.mods:
        want = ch.generate(short_wav, -12, 100, module=m.name)
        got = ft.run(short_wav, -12, 100, module=m.name, dtype=np.float64)
        assert got.shape == want.shape
        err = np.abs(got - want).max()
        if m.type in ("audiofile", "fft", "mel", "dct"):
            assert np.array_equal(got, want), (m.name, err)      # bit for bit (mel: glibc's logf on the device)
        elif m.type == "power":
            assert err <= 1e-12 * max(1.0, np.abs(want).max()), (m.name, err)
        else:
            assert err <= FEAT_TOL, (m.name, err)


def test_synthetic_audio_and_float_output(capi, oracle, golden_dir):
    pcm = synth.make_audio(80000)
    cfg = _cfg(golden_dir, "mfcc_cms_norm")
    ch = oracle.FeatureChain(cfg)
    ft = capi.Feat(cfg)
    n = ft.last_frame(len(pcm)) + 1
    # the graph's look-around (55, 30) + the 7 rows the mean subtractor's kernel reads before its window (kCmsLead)
    assert n == 623 and ft.halo() == (55 + 7, 30) and ch.halo() == (55, 30)
    want = ch.generate(pcm, 0, n)
    got64 = ft.run(pcm, 0, n, dtype=np.float64)
    got32 = ft.run(pcm, 0, n)
    assert np.abs(got64 - want).max() <= FEAT_TOL
    assert np.array_equal(got32, got64.astype(np.float32))


def test_block_partition_invariance(capi, golden_dir):
    """random_feature_test.cc analogue: any partition of the frame range gives
    bit-identical features (each frame is a pure function of its index)."""
    pcm = synth.make_audio(40000, seed=5)
    ft = capi.Feat(_cfg(golden_dir, "mfcc_cms_norm"))
    whole = ft.run(pcm, -20, 360, dtype=np.float64)
    parts = np.vstack([ft.run(pcm, -20, 7, dtype=np.float64), ft.run(pcm, -13, 200, dtype=np.float64),
                       ft.run(pcm, 187, 153, dtype=np.float64)])
    assert np.array_equal(whole.view(np.uint64), parts.view(np.uint64))
    rng = np.random.default_rng(0)
    for f in rng.integers(-20, 340, 10):
        one = ft.run(pcm, int(f), 1, dtype=np.float64)
        assert np.array_equal(one[0].view(np.uint64), whole[f + 20].view(np.uint64))


def test_batch_equals_per_utterance(capi, golden_dir):
    import torch
    ft = capi.Feat(_cfg(golden_dir, "mfcc_cms_norm"))
    utts = [synth.make_audio(n, seed=i) for i, n in enumerate([16000, 9506, 48000, 257 + 128 * 3])]
    frames = [ft.last_frame(len(u)) + 1 for u in utts]
    pcm_off = np.concatenate([[0], np.cumsum([len(u) for u in utts])])
    frame_off = np.concatenate([[0], np.cumsum(frames)])
    d_pcm = torch.from_numpy(np.concatenate(utts)).cuda()
    d_out = torch.empty((int(frame_off[-1]), 39), dtype=torch.float32, device="cuda")
    ft.run_batch_dev(d_pcm, pcm_off, frame_off, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for u, (a, b) in zip(utts, zip(frame_off[:-1], frame_off[1:])):
        assert np.array_equal(got[a:b], ft.run(u, 0, int(b - a)))


def test_batch_of_very_unequal_utterances(capi, golden_dir):
    """The kernels find a workgroup's utterance from an equal-length first guess + a short walk, a bisection when that
    fails (find_utt_near): one long utterance among many short ones, at either end, puts most workgroups on the
    bisection."""
    import torch
    ft = capi.Feat(_cfg(golden_dir, "mfcc_cms_norm"))
    short = [synth.make_audio(1200 + 37 * i, seed=100 + i) for i in range(40)]
    long_one = synth.make_audio(160000, seed=99)
    for utts in ([long_one] + short, short + [long_one], short[:20] + [long_one] + short[20:]):
        frames = [ft.last_frame(len(u)) + 1 for u in utts]
        pcm_off = np.concatenate([[0], np.cumsum([len(u) for u in utts])])
        frame_off = np.concatenate([[0], np.cumsum(frames)])
        d_pcm = torch.from_numpy(np.concatenate(utts)).cuda()
        d_out = torch.empty((int(frame_off[-1]), 39), dtype=torch.float32, device="cuda")
        ft.run_batch_dev(d_pcm, pcm_off, frame_off, d_out)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        for k in (0, 1, 19, 20, 21, 39, 40):
            a, b = int(frame_off[k]), int(frame_off[k + 1])
            assert np.array_equal(got[a:b], ft.run(utts[k], 0, b - a)), k


@pytest.mark.parametrize("dim_cep,out_dim", [(12, 39), (12, 26), (12, 40), (10, 33), (7, 25), (4, 1)])
def test_temporal_kernel_shapes(capi, oracle, dim_cep, out_dim):
    """k_temporal_fused forms the transform in blocks of three output rows x two frames: output dimensions that are
    not multiples of three, non-square transforms and other source widths, against the oracle (and the fused path
    against the module-by-module one)."""
    rng = np.random.default_rng(dim_cep * 100 + out_dim)
    cfg = synth.make_feature_config(dim_cep=dim_cep)
    d = 3 * (dim_cep + 1)
    a = 0.2 * rng.standard_normal((out_dim, d)) + np.eye(out_dim, d)
    bias = 0.1 * rng.standard_normal(out_dim)
    lines = cfg.split("\n")
    out = []
    for ln in lines:
        t = ln.strip()
        if t.startswith("matrix "):
            ln = "  matrix " + " ".join("%.6g" % x for x in a.ravel())
        elif t.startswith("bias "):
            ln = "  bias " + " ".join("%.6g" % x for x in bias)
        elif t == "dim %d" % d:
            ln = "  dim %d" % out_dim
        out.append(ln)
    cfg = "\n".join(out)
    pcm = synth.make_audio(30000, seed=out_dim)
    ch = oracle.FeatureChain(cfg)
    ft = capi.Feat(cfg)
    assert ft.dim == ch.dim == out_dim
    n = ft.last_frame(len(pcm)) + 1
    want = ch.generate(pcm, -3, n + 6)
    got = ft.run(pcm, -3, n + 6, dtype=np.float64)
    assert np.abs(got - want).max() <= FEAT_TOL
    capi.debug_feat_fusion(False)
    try:
        unfused = ft.run(pcm, -3, n + 6, dtype=np.float64)
    finally:
        capi.debug_feat_fusion(True)
    # the temporal kernel is bit-identical to the chain of module kernels; the tiled mean subtractor groups its window
    # sums differently from the frame-order loop (1e-15 relative)
    assert np.abs(got - unfused).max() <= 1e-12 * max(1.0, np.abs(unfused).max())


@pytest.mark.parametrize("left,right", [(75, 75), (3, 200), (400, 400), (0, 0)])
def test_mean_subtractor_windows(capi, oracle, left, right):
    """The mean subtractor's kernel picks its tile by the window (128 rows x 512 threads, 64 x 256, 128 x 512 alone on a CU
    for the reference's default 75 + 75, the untiled kernel beyond): each against the oracle."""
    cfg = synth.make_feature_config().replace("left 50", "left %d" % left).replace("right 25", "right %d" % right)
    assert "left %d" % left in cfg
    pcm = synth.make_audio(56000, seed=3)
    ch = oracle.FeatureChain(cfg)
    ft = capi.Feat(cfg)
    n = ft.last_frame(len(pcm)) + 1
    want = ch.generate(pcm, -5, n + 9)
    got = ft.run(pcm, -5, n + 9, dtype=np.float64)
    assert np.abs(got - want).max() <= FEAT_TOL
    # a frame's value does not depend on the partition of the range
    part = np.vstack([ft.run(pcm, -5, 133, dtype=np.float64), ft.run(pcm, 128, n + 9 - 133, dtype=np.float64)])
    assert np.array_equal(got.view(np.uint64), part.view(np.uint64))


def test_options_copy_borders_window_magnitude(capi, oracle):
    cfg = """module
{
  name audio
  type audiofile
  sample_rate 8000
  frame_rate 100
  window_width 128
  copy_borders 0
  pre_emph_coef 0.95
}
module
{
  name fft
  type fft
  magnitude 1
  log 1
  sources audio
}
module
{
  name mel
  type mel
  root 1
  sources fft
}
module
{
  name c
  type dct
  dim 8
  zeroth 1
  sources mel
}
module
{
  name d
  type delta
  width 3
  normalization 7.5
  sources c
}
module
{
  name n
  type normalization
  var 4 4 4 4 4 4 4 4
  mean 1 0 0 0 0 0 0 0
  sources d
}
module
{
  name t
  type lin_transform
  dim 8
  bias 1 2 3 4 5 6 7 8
  sources n
}
"""
    pcm = synth.make_audio(8000, seed=3, sample_rate=8000)
    ch = oracle.FeatureChain(cfg)
    ft = capi.Feat(cfg)
    assert ft.sample_rate == 8000 and abs(ft.frame_rate - 100) < 1e-6
    for name in ["audio", "fft", "mel", "c", "d", "n", "t"]:
        want = ch.generate(pcm, -3, 110, module=name)
        got = ft.run(pcm, -3, 110, module=name, dtype=np.float64)
        scale = max(1.0, np.abs(want[np.isfinite(want)]).max())
        both = np.isfinite(want) & np.isfinite(got)
        # the non-finite pattern (log of an empty mel bin) is asserted equal, the finite entries to FEAT_TOL
        assert (np.isfinite(want) == np.isfinite(got)).all(), name
        assert np.array_equal(want[~both], got[~both], equal_nan=True), name
        assert np.abs(got[both] - want[both]).max() <= FEAT_TOL * scale, name


@pytest.mark.parametrize("rate,frame_rate,width", [(16000, 100, 0), (16000, 100, 400), (8000, 100, 200),
                                                   (16000, 125, 240), (44100, 100, 1000),
                                                   (11025, 100, 220), (22050, 100, 442), (16000, 100, 364),
                                                   (16000, 125, 276)])
def test_non_power_of_two_windows(capi, oracle, rate, frame_rate, width):
    """25-ms style windows need KissFFT's radix-3/5 butterflies (kf_bfly3 / kf_bfly5); half lengths
    with prime factors 7, 11, 13, 17, 23 (220, 442, 364, 276 samples) its generic butterfly
    (kf_bfly_generic) -- the oracle's FFT is pinned bit for bit on the vendored KissFFT for all of
    them (tests/test_oracle_golden.py)."""
    cfg = "module\n{\n name a\n type audiofile\n sample_rate %d\n frame_rate %d\n%s}\n" % (
        rate, frame_rate, (" window_width %d\n" % width) if width else "")
    cfg += "module\n{\n name f\n type fft\n magnitude 0\n sources a\n}\n"
    cfg += "module\n{\n name m\n type mel\n sources f\n}\n"
    pcm = synth.make_audio(rate, seed=9, sample_rate=rate)
    ch = oracle.FeatureChain(cfg)
    ft = capi.Feat(cfg)
    n = min(60, ft.last_frame(len(pcm)) + 1)
    for mod in ("f", "m"):
        want = ch.generate(pcm, -2, n, module=mod)
        got = ft.run(pcm, -2, n, module=mod, dtype=np.float64)
        if mod == "f":
            assert np.array_equal(got, want)          # spectrum bit-identical
        else:
            assert np.abs(got - want).max() <= FEAT_TOL


def test_set_parameters(capi, oracle, golden_dir, short_wav):
    cfg = _cfg(golden_dir, "mfcc_cms_norm")
    ft = capi.Feat(cfg)
    base = ft.run(short_wav, 0, 50, dtype=np.float64)
    ft.set_parameters("normalization", "{\n mean " + " ".join(["0"] * 39) + "\n scale " + " ".join(["1"] * 39) + "\n}\n")
    changed = ft.run(short_wav, 0, 50, dtype=np.float64)
    assert np.abs(changed - base).max() > 1e-3
    with pytest.raises(capi.AasrError, match="Invalid mean dimension"):
        ft.set_parameters("normalization", "{\n mean 1 2 3\n}\n")


def test_config_errors(capi):
    def err(text):
        with pytest.raises(capi.AasrError) as ei:
            capi.Feat(text)
        return ei.value
    assert "Unknown module type" in err("module\n{\n name a\n type nosuch\n}\n").msg
    assert "PreModule: Must set dimension" in err("module\n{\n name a\n type pre\n}\n").msg
    assert "pwlin_vtln and slapt" in err("module\n{\n name a\n type audiofile\n sample_rate 16000\n}\nmodule\n{\n name f\n type fft\n sources a\n}\n"
                                         "module\n{\n name v\n type vtln\n pwlin_vtln 1\n slapt 1\n sources f\n}\n").msg
    assert "first module should be a base module" in err("module\n{\n name a\n type fft\n}\n").msg
    assert "Must set sample rate" in err("module\n{\n name a\n type audiofile\n}\n").msg
    assert "value redefined" in err("module\n{\n name a\n name b\n type audiofile\n}\n").msg
    ft = capi.Feat("module\n{\n name a\n type audiofile\n sample_rate 16000\n}\n")
    with pytest.raises(capi.AasrError) as ei:
        ft.run(np.zeros(100, np.int16), 0, 1)
    assert ei.value.code == capi.AASR_ERR_SHORT_AUDIO


ADAPT_CFG = """module
{
  name audiofile
  type audiofile
  sample_rate 16000
}
module
{
  name fft
  type fft
  magnitude 0
  sources audiofile
}
module
{
  name vtln
  type vtln
  %s
  sources fft
}
module
{
  name mel
  type mel
  sources vtln
}
module
{
  name mel_power
  type mel_power
  sources mel
}
module
{
  name dct
  type dct
  sources mel
}
module
{
  name qe
  type quanteq
  sources mel
}
module
{
  name concat
  type concat
  left 3
  right 3
  sources dct
}
module
{
  name sr_norm
  type sr_norm
  in_frames 7
  out_frames 5
  sources concat
}
module
{
  name merge
  type merge
  sources sr_norm mel_power qe
}
"""


def _block(opts):
    return "{\n" + "".join(" %s %s\n" % kv for kv in opts.items()) + "}\n"


@pytest.mark.parametrize("vtln_opts,params", [
    ("", {"warp_factor": "1.12"}),                                   # bilinear + Lanczos sinc
    ("pwlin_vtln 1\n  pwlin_turnpoint 0.85", {"warp_factor": "0.91"}),
    ("slapt 1", {"slapt_coef": "0.02 -0.007"}),
    ("sinc_interpolation_rad 0", {"warp_factor": "1.07"}),           # linear interpolation
    ("lanczos_window 0\n  sinc_interpolation_rad 5", {"warp_factor": "0.95"}),
    ("all-pass 1", {"warp_factor": "1.05"}),
    ("all-pass 1\n  slapt 1", {"slapt_coef": "0.015 -0.004"}),     # aku/FeatureModules.cc:1758-1868
])
def test_adaptation_modules_match_oracle(capi, oracle, vtln_opts, params):
    """vtln / sr_norm / quanteq / concat / mel_power (the modules SpeakerConfig
    drives) before and after set_parameters."""
    cfg = ADAPT_CFG % vtln_opts
    pcm = synth.make_audio(24000, seed=11)
    ch = oracle.FeatureChain(cfg)
    ft = capi.Feat(cfg)
    assert ft.dim == ch.dim == 5 * 12 + 1 + 21 and ft.halo() == (3, 3)

    def compare(tag):
        for m in ch.mods:
            want = ch.generate(pcm, -6, 70, module=m.name)
            got = ft.run(pcm, -6, 70, module=m.name, dtype=np.float64)
            assert got.shape == want.shape, (tag, m.name)
            scale = max(1.0, np.abs(want).max())
            if m.type in ("audiofile", "fft", "vtln"):
                assert np.abs(got - want).max() <= 1e-12 * scale, (tag, m.name)
            else:
                assert np.abs(got - want).max() <= FEAT_TOL * max(1.0, scale / 30), (tag, m.name)
        return ft.run(pcm, -6, 70, dtype=np.float64)

    base = compare("unit warp")
    if "all-pass" not in vtln_opts:
        # no warp: the vtln module is the identity on the spectrum
        assert np.array_equal(ft.run(pcm, 0, 30, module="vtln", dtype=np.float64),
                              ft.run(pcm, 0, 30, module="fft", dtype=np.float64)) or "lanczos_window 0" in vtln_opts
    speaker = {"vtln": params, "sr_norm": {"speech_rate": "1.25"},
               "qe": {"alpha": " ".join(["0.6"] * 21), "gamma": " ".join(["0.8"] * 21),
                      "quant_max": " ".join(["12.5"] * 21)}}
    for mod, opts in speaker.items():
        ch.set_parameters(mod, opts)
        ft.set_parameters(mod, _block(opts))
    adapted = compare("speaker parameters")
    assert np.abs(adapted - base).max() > 1e-2
    # back to the defaults (what SpeakerConfig does for a speaker without entries)
    for mod in speaker:
        ch.set_parameters(mod, {})
        ft.set_parameters(mod, "{\n}\n")
    assert np.array_equal(ft.run(pcm, -6, 70, dtype=np.float64), base)


def test_block_requests_upload_only_their_samples(capi):
    """aasr_feat_run sends only the samples a block of frames can reach to the device.  A handle whose
    device buffer holds another recording everywhere else must give the bits of a handle that saw the
    whole file: blocks at the start, in the middle, across the end, negative and post-EOF frames."""
    from aaltoasr_amd import synth
    cfg = os.path.join(os.path.dirname(__file__), "golden", "mfcc_cms_norm.feaconf")
    pcm = synth.make_audio(16000 * 40, seed=41)
    other = synth.make_audio(16000 * 40, seed=42)
    whole = capi.Feat.from_file(cfg)
    last = whole.last_frame(len(pcm))
    ref = whole.run(pcm, -60, last + 1 + 120, dtype=np.float64)      # frames -60 .. last+59
    part = capi.Feat.from_file(cfg)
    for first, n in ((-60, 70), (0, 1), (1000, 256), (2499, 3), (last - 100, 160), (last + 5, 20), (-3, last + 10)):
        part.run(other, 0, last + 1, dtype=np.float64)               # poison the device copy
        got = part.run(pcm, first, n, dtype=np.float64)
        assert np.array_equal(got, ref[first + 60:first + 60 + n]), (first, n)
        for name in ("fft", "mel"):
            a = part.run(other, 0, 8, module=name)
            g2 = part.run(pcm, first, min(n, 40), module=name)
            w2 = whole.run(pcm, first, min(n, 40), module=name)
            assert np.array_equal(g2, w2), (name, first)


def _fusion_graph(window=0, magnitude=0, root=0, zeroth=0, dct_dim=12, w1=2, w2=3, with_matrix=True,
                  with_cms=True, sample_rate=16000, frame_rate=125):
    rng = np.random.default_rng(window + 7 * dct_dim + w1)
    d = 3 * (dct_dim + 1)

    def vec(v):
        return " ".join("%.6g" % x for x in np.asarray(v).ravel())

    def mod(name, typ, src=None, **kw):
        t = "module\n{\n  name %s\n  type %s\n" % (name, typ)
        for k, v in kw.items():
            t += "  %s %s\n" % (k, v)
        return t + ("  sources %s\n" % src if src else "") + "}\n"

    a = dict(sample_rate=sample_rate, frame_rate=frame_rate)
    if window:
        a["window_width"] = window
    cfg = mod("a", "audiofile", **a) + mod("f", "fft", "a", magnitude=magnitude)
    cfg += mod("m", "mel", "f", root=root) + mod("p", "power", "f")
    cfg += mod("c", "dct", "m", dim=dct_dim, zeroth=zeroth) + mod("cp", "merge", "c p")
    cfg += mod("d1", "delta", "cp", width=w1) + mod("d2", "delta", "d1", width=w2) + mod("all", "merge", "cp d1 d2")
    cfg += mod("nrm", "normalization", "all", mean=vec(rng.normal(0, 2, d)), scale=vec(rng.uniform(0.05, 0.5, d)))
    lt = dict(dim=d)
    if with_matrix:
        lt["matrix"] = vec(np.eye(d) + 0.1 * rng.standard_normal((d, d)))
        lt["bias"] = vec(0.1 * rng.standard_normal(d))
    cfg += mod("lt", "lin_transform", "nrm", **lt)
    if with_cms:
        cfg += mod("cms", "mean_subtractor", "lt", left=int(rng.integers(0, 60)), right=int(rng.integers(0, 40)))
    return cfg


@pytest.mark.parametrize("kw", [dict(), dict(window=400, magnitude=1, root=1), dict(window=512, zeroth=1, dct_dim=9),
                                dict(window=320, w1=1, w2=1, with_matrix=False), dict(with_cms=False, w1=3, w2=2),
                                dict(window=200, sample_rate=8000, frame_rate=100, dct_dim=5)])
def test_fused_kernels_equal_the_module_by_module_path(capi, oracle, kw):
    """The fused spectral (audiofile..merge) and temporal (delta..lin_transform) kernels and the
    mean subtractor writing the float output itself, against every module running its own kernel:
    identical bits for the output and for every module tap, on a batch of utterances of very
    different lengths, incl. frames before 0 and past the end; and both agree with the oracle."""
    import torch
    cfg = _fusion_graph(**kw)
    ft = capi.Feat(cfg)
    sr = ft.sample_rate
    lens = [sr * 2, 700, sr // 2 + 13, sr * 3 + 7, 513, sr]
    utts = [synth.make_audio(n, seed=300 + i, sample_rate=sr) for i, n in enumerate(lens)]
    frames = [ft.last_frame(len(u)) + 1 for u in utts]
    pcm_off = np.concatenate([[0], np.cumsum([len(u) for u in utts])]).astype(np.int64)
    frame_off = np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)
    d_pcm = torch.from_numpy(np.concatenate(utts)).cuda()
    outs = []
    for fused in (True, False):
        capi.debug_feat_fusion(fused)
        try:
            d_out = torch.zeros((int(frame_off[-1]), ft.dim), dtype=torch.float32, device="cuda")
            ft.run_batch_dev(d_pcm, pcm_off, frame_off, d_out)
            torch.cuda.synchronize()
            taps = {name: ft.run(utts[0], -9, frames[0] + 15, module=name, dtype=np.float64) for name, _ in ft.modules()}
            taps["__f32"] = ft.run(utts[3], -30, 90)
            outs.append((d_out.cpu().numpy(), taps))
        finally:
            capi.debug_feat_fusion(True)
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    for name in outs[0][1]:
        assert np.array_equal(outs[0][1][name], outs[1][1][name]), name
    ch = oracle.FeatureChain(cfg)
    for u in (0, 1, 4):
        want = ch.generate(utts[u], 0, frames[u])
        got = outs[0][0][int(frame_off[u]):int(frame_off[u + 1])]
        # float32 output: one rounding of the double chain
        assert np.abs(got - want).max() <= 6e-8 * max(1.0, np.abs(want).max()) + FEAT_TOL


def test_eof_frame_matches_the_sequential_reader(capi, oracle):
    """aasr_feat_eof_frame = the frame at which AudioFileModule::generate meets the end of the file
    (aku/FeatureModules.cc:399-413), i.e. how many frames phone_probs emits: equal to the oracle's
    brute-force-checked orc_eof_frame for integral and fractional window advances and for lengths
    beyond 2^24 samples, where last_frame()'s float formula is one off; a whole-file run emits
    exactly that many frames."""
    rng = np.random.default_rng(3)
    parted = 0
    for rate, fr, width in ((16000, 125, 0), (11025, 100, 220), (22050, 100, 442), (11025, 125, 200), (44100, 100, 1000)):
        cfg = "module\n{\n name a\n type audiofile\n sample_rate %d\n frame_rate %d\n%s}\n" % (
            rate, fr, (" window_width %d\n" % width) if width else "")
        cfg += "module\n{\n name f\n type fft\n magnitude 0\n sources a\n}\n"
        ft = capi.Feat(cfg)
        ch = oracle.FeatureChain(cfg)
        for n in list(rng.integers(2000, 300000, 40)) + list(rng.integers((1 << 24) - 2000, (1 << 24) + 200000, 20)) + [57600000]:
            n = int(n)
            assert ft.eof_frame(n) == ch.num_frames(n), (rate, fr, width, n)
            assert ft.last_frame(n) == ch.last_frame(n)
            parted += ft.eof_frame(n) != ft.last_frame(n) + 1
    assert parted > 0
    # a fractional advance where the two differ: the run emits eof_frame frames, the last one whole
    cfg = "module\n{\n name a\n type audiofile\n sample_rate 11025\n frame_rate 125\n window_width 200\n}\n"
    cfg += "module\n{\n name f\n type fft\n magnitude 0\n sources a\n}\n"
    ft = capi.Feat(cfg)
    ch = oracle.FeatureChain(cfg)
    n = next(int(k) for k in range(30000, 31000) if ft.eof_frame(int(k)) != ft.last_frame(int(k)) + 1)
    pcm = synth.make_audio(n, seed=2, sample_rate=11025)
    T = ft.eof_frame(n)
    assert np.array_equal(ft.run(pcm, 0, T + 3, dtype=np.float64), ch.generate(pcm, 0, T + 3))
