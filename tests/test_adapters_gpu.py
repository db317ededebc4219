"""The aku-shaped C++ surface on the GPU: the reference tool's command line
(phone_probs) and the reference's own per-frame calling sequence written
against aku::FeatureGenerator / aku::HmmSet adapters."""
import math
import os
import subprocess
import wave

import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aaltoasr_amd", "lib", "bin")


def _write_wav(path, pcm, rate=16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(pcm.astype("<i2").tobytes())


@pytest.fixture(scope="module")
def world(capi, oracle, golden_dir, tmp_path_factory):
    d = tmp_path_factory.mktemp("aku")
    cfg_path = os.path.join(golden_dir, "mfcc_cms_norm.feaconf")
    model = synth.make_model(D=39, G=256, S=32, comps=8)
    base = str(d / "model")
    oracle.write_gk(base + ".gk", model[0], model[1])
    oracle.write_mc(base + ".mc", model[2], model[3], model[4])
    oracle.write_ph(base + ".ph", 32)
    pcms = [synth.make_audio(n, seed=40 + i) for i, n in enumerate([80000, 16000, 30000])]
    lines = []
    for i, p in enumerate(pcms):
        _write_wav(str(d / ("a%d.wav" % i)), p)
        lines.append("audio=%s lna=%s" % (d / ("a%d.wav" % i), d / ("a%d.lna" % i)))
    recipe = str(d / "test.recipe")
    open(recipe, "w").write("\n".join(lines) + "\n")
    return dict(dir=d, cfg=cfg_path, base=base, pcms=pcms, recipe=recipe, model=model,
                ft=capi.Feat.from_file(cfg_path), gm=capi.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph"))


@pytest.mark.parametrize("nbytes", [2, 4])
def test_phone_probs_cli(capi, world, nbytes):
    out = world["dir"] / ("cli%d" % nbytes)
    os.makedirs(out)
    r = subprocess.run([os.path.join(BIN, "phone_probs"), "-b", world["base"], "-c", world["cfg"],
                        "-r", world["recipe"], "-a", "-o", str(out), "--lnabytes=%d" % nbytes, "-i", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Processing file 1/3" in r.stdout
    for i, pcm in enumerate(world["pcms"]):
        want, _ = capi.run_utterance(world["ft"], world["gm"], pcm, lnabytes=nbytes)
        assert open(out / ("a%d.lna" % i), "rb").read() == want


def test_phone_probs_cli_batches_and_errors(world):
    exe = os.path.join(BIN, "phone_probs")
    out = world["dir"] / "clib"
    os.makedirs(out)
    for k in (1, 2):
        r = subprocess.run([exe, "-b", world["base"], "-c", world["cfg"], "-r", world["recipe"], "-a",
                            "-o", str(out), "-B", "2", "-I", str(k)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(out)) == ["a0.lna", "a1.lna", "a2.lna"]
    r = subprocess.run([exe, "-b", world["base"], "-c", world["cfg"], "-r", world["recipe"], "-B", "2"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Must give both --batch and --bindex" in r.stderr
    r = subprocess.run([exe, "-b", world["base"], "-c", world["cfg"], "-r", world["recipe"], "-C", "x.gcl"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "could not open x.gcl" in r.stderr
    r = subprocess.run([exe, "-b", world["base"], "-c", world["cfg"], "-r", world["recipe"], "-S", "x.spkc"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "could not open x.spkc" in r.stderr
    # --sort-recipe: stable sort by speaker id (Recipe::sort_infos); processing order changes
    rec = str(world["dir"] / "spk.recipe")
    lines = open(world["recipe"]).read().split("\n")[:3]
    open(rec, "w").write("\n".join(l + " speaker=%s" % s for l, s in zip(lines, "bab")) + "\n")
    r = subprocess.run([exe, "-b", world["base"], "-c", world["cfg"], "-r", rec, "-a", "-o", str(out),
                        "--sort-recipe", "-i", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    order = [l.split("/")[-1] for l in r.stdout.split("\n") if l.startswith("Input:")]
    assert order == ["a1.wav", "a0.wav", "a2.wav"]
    r = subprocess.run([exe, "-c", world["cfg"], "-r", world["recipe"]], capture_output=True, text=True)
    assert r.returncode != 0 and "Must give either --base" in r.stderr


@pytest.mark.parametrize("nbytes", [2, 4])
def test_reference_style_frame_loop_on_adapters(capi, oracle, world, nbytes):
    """aku_adapter_check is phone_probs.cc's frame loop verbatim on the adapter
    classes: generate(f) / eof() / reset_cache / precompute_likelihoods /
    state_likelihood, float normalisation on the host."""
    out = str(world["dir"] / ("loop%d.lna" % nbytes))
    r = subprocess.run([os.path.join(BIN, "aku_adapter_check"), world["cfg"], world["base"],
                        str(world["dir"] / "a0.wav"), out, str(nbytes)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = open(out, "rb").read()
    eng, frames = capi.run_utterance(world["ft"], world["gm"], world["pcms"][0], lnabytes=nbytes)
    assert frames == 623 and len(got) == len(eng) and got[:5] == eng[:5]
    a = oracle.lna_decode(got)
    b = oracle.lna_decode(eng)
    ch = oracle.FeatureChain(open(world["cfg"]).read())
    ll_ref = oracle.DiagModel(*world["model"]).score(ch.generate(world["pcms"][0], 0, 623))
    smooth = (ll_ref > -87.0) | (ll_ref < -104.5)
    tol = 1.0 / 1820 + 1e-6 if nbytes == 2 else 2e-5
    assert np.abs(a - b)[smooth].max() <= tol


def test_clustering_through_cli_and_adapters(capi, oracle, world):
    """phone_probs -C GCL --eval-ming R (aku/phone_probs.cc:112-117) and the same
    two HmmSet calls on the adapter class give the engine's clustered scores."""
    gcl = str(world["dir"] / "model.gcl")
    oracle.write_gcl(gcl, 16, synth.make_clustering(world["model"][0], 16))
    gm = capi.Gmm.from_files(world["base"] + ".gk", world["base"] + ".mc", world["base"] + ".ph")
    gm.read_clustering(gcl)
    gm.set_clustering_min_evals(0.0, 0.25)
    want, frames = capi.run_utterance(world["ft"], gm, world["pcms"][0], lnabytes=4)
    plain, _ = capi.run_utterance(world["ft"], world["gm"], world["pcms"][0], lnabytes=4)
    assert want != plain                      # the approximation is visible
    out = world["dir"] / "clicl"
    os.makedirs(out)
    r = subprocess.run([os.path.join(BIN, "phone_probs"), "-b", world["base"], "-c", world["cfg"],
                        "-r", world["recipe"], "-a", "-o", str(out), "--lnabytes=4", "-C", gcl,
                        "--eval-ming", "0.25"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(out / "a0.lna", "rb").read() == want
    loop = str(world["dir"] / "loopcl.lna")
    r = subprocess.run([os.path.join(BIN, "aku_adapter_check"), world["cfg"], world["base"],
                        str(world["dir"] / "a0.wav"), loop, "4", gcl, "0", "0.25"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    a, b = oracle.lna_decode(open(loop, "rb").read()), oracle.lna_decode(want)
    smooth = (b > -80.0)
    assert a.shape == b.shape and np.abs(a - b)[smooth].max() <= 2e-5


@pytest.mark.parametrize("nbytes,block", [(4, 100), (2, 4096), (4, 1)])
def test_engine_acoustics_equals_lna_reader_view(capi, oracle, world, nbytes, block):
    """the decoder's Acoustics interface (go_to / log_prob) served from
    the GPU returns exactly what LnaReaderCircular reads from the LNA file phone_probs writes
    (decoder/src/LnaReaderCircular.cc:129-209)."""
    out = str(world["dir"] / ("ac_%d_%d.f32" % (nbytes, block)))
    r = subprocess.run([os.path.join(BIN, "acoustics_check"), world["cfg"], world["base"],
                        str(world["dir"] / "a1.wav"), str(nbytes), str(block), out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lna, frames = capi.run_utterance(world["ft"], world["gm"], world["pcms"][1], lnabytes=nbytes)
    assert r.stdout.split()[0] == str(frames) and r.stdout.split()[-1] == str(frames)
    got = np.fromfile(out, np.float32).reshape(frames, 32)
    body = np.frombuffer(lna[5:], np.uint8)
    if nbytes == 4:
        want = body.view("<f4").reshape(frames, 32)
    else:
        b = body.reshape(-1, 2).astype(np.int64)
        want = ((b[:, 0] * 256 + b[:, 1]) / -1820.0).astype(np.float32).reshape(frames, 32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nbytes", [2, 4])
def test_engine_lna_files_through_the_reference_decoder_reader(capi, oracle, world, nbytes):
    """Files written by this engine's phone_probs, opened with the reference recogniser's own
    LnaReaderCircular (compiled in place into oracle/_ref/liblna_ref.so; the prebuilt library
    travels to the GPU box): frame count, state count and every log_prob() are what
    EngineAcoustics serves in memory."""
    if oracle.ref_lna() is None:
        pytest.skip("oracle/_ref/liblna_ref.so not built")
    lna, frames = capi.run_utterance(world["ft"], world["gm"], world["pcms"][1], lnabytes=nbytes)
    path = str(world["dir"] / ("reader_%d.lna" % nbytes))
    with open(path, "wb") as f:
        f.write(lna)
    got = oracle.ref_lna_read(path, frames + 10, 32, buf_size=16, order=1)
    assert got.shape == (frames, 32)
    out = str(world["dir"] / ("ac_reader_%d.f32" % nbytes))
    r = subprocess.run([os.path.join(BIN, "acoustics_check"), world["cfg"], world["base"],
                        str(world["dir"] / "a1.wav"), str(nbytes), "256", out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(got, np.fromfile(out, np.float32).reshape(frames, 32))


def test_audio_containers_and_raw_endian_options_through_phone_probs(capi, world):
    """AudioFileModule's `raw` / `endian` options (aku/FeatureModules.cc:345-356) and the
    containers libsndfile opens for the reference: the same samples as big-endian headerless
    PCM (`raw 1`, `endian big`), Sun AU, AIFF and NIST SPHERE files give byte-identical LNA
    files to the WAV input, from the CLI (recipe path) and the per-frame adapters (stream open)."""
    import sunau
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        import aifc
    d = world["dir"]
    pcm = world["pcms"][1]
    want, _ = capi.run_utterance(world["ft"], world["gm"], pcm)
    be = pcm.astype(">i2").tobytes()
    open(str(d / "c.raw"), "wb").write(be)
    with sunau.open(str(d / "c.au"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.setcomptype("NONE", "")
        w.writeframes(be)
    with aifc.open(str(d / "c.aiff"), "wb") as w:
        w.aiff(); w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(be)
    head = ("NIST_1A\n   1024\nsample_count -i %d\nsample_n_bytes -i 2\nchannel_count -i 1\n"
            "sample_byte_format -s2 10\nsample_rate -i 16000\nsample_coding -s3 pcm\nend_head\n" % len(pcm))
    open(str(d / "c.sph"), "wb").write(head.encode().ljust(1024, b" ") + be)
    cfg_text = open(world["cfg"]).read()
    raw_cfg = str(d / "raw_be.feaconf")
    assert "sample_rate 16000" in cfg_text
    open(raw_cfg, "w").write(cfg_text.replace("sample_rate 16000", "sample_rate 16000\n  raw 1\n  endian big", 1))
    for name, cfg in (("c.raw", raw_cfg), ("c.au", world["cfg"]), ("c.aiff", world["cfg"]), ("c.sph", world["cfg"])):
        rec = str(d / "cont.recipe")
        out = str(d / (name + ".lna"))
        open(rec, "w").write("audio=%s lna=%s\n" % (d / name, out))
        r = subprocess.run([os.path.join(BIN, "phone_probs"), "-b", world["base"], "-c", cfg, "-r", rec],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert open(out, "rb").read() == want, name
    # a headerless little-endian file needs no option at all (AudioReader::open's fallback), and
    # `raw 1` makes a WAV header part of the signal, as enforce_raw does
    open(str(d / "c_le.raw"), "wb").write(pcm.astype("<i2").tobytes())
    got, rate = capi.audio_read(str(d / "c_le.raw"), world["ft"])
    assert rate == 16000 and np.array_equal(got, pcm)
    ft_raw = capi.Feat(open(raw_cfg).read().replace("endian big", "endian little"))
    got, _ = capi.audio_read(str(d / "a1.wav"), ft_raw)
    assert len(got) == len(pcm) + 22 and np.array_equal(got[22:], pcm)
    # sample-rate mismatch: the reference's message (aku/FeatureModules.cc:254-260)
    _write_wav(str(d / "c8k.wav"), pcm, rate=8000)
    with pytest.raises(capi.AasrError) as ei:
        capi.audio_read(str(d / "c8k.wav"), world["ft"])
    assert "Audio file sample rate (8000 Hz) and model configuration (16000 Hz) don't agree." in str(ei.value)


def test_phone_probs_takes_the_reference_option_grammar(world):
    """Grouped short options with deferred arguments, --name=value, and `-i10` being three options
    (aku/conf.hh:12-40); the grammar itself is pinned against aku/conf.cc in tests/test_conf_cli.py."""
    d = world["dir"]
    out = d / "grammar"
    out.mkdir(exist_ok=True)
    exe = os.path.join(BIN, "phone_probs")
    r = subprocess.run([exe, "-bcr", world["base"], world["cfg"], world["recipe"], "-aN", "--lnabytes=4",
                        "--output-dir", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    os.makedirs(str(d / "grammar2"), exist_ok=True)
    r2 = subprocess.run([exe, "-b", world["base"], "-c", world["cfg"], "-r", world["recipe"], "-a", "-N",
                         "--lnabytes", "4", "-o", str(d / "grammar2") + "/"], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr
    for i in range(3):
        a = open(str(out / ("a%d.lna" % i)), "rb").read()
        assert a == open(str(d / "grammar2" / ("a%d.lna" % i)), "rb").read() and a[4] == 4
    r = subprocess.run([exe, "-b", world["base"], "-c", world["cfg"], "-r", world["recipe"], "-i10"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and r.stderr.startswith("invalid option -1")


def test_feature_module_view_of_the_graph(capi, golden_dir, world):
    """aku::FeatureGenerator::module(name) -> aku::FeatureModule (name, type_str, dim, at(frame),
    get_parameters / set_parameters through aku::ModuleConfig): what a caller of the reference's
    plugin API sees is what the C ABI computes for the same module, including border frames, and a
    parameter change made through the view reaches the generator's output."""
    import wave
    cfg = os.path.join(golden_dir, "mfcc_cms_norm.feaconf")
    wav = os.path.join(golden_dir, "short.wav")
    out = str(world["dir"] / "modules.txt")
    r = subprocess.run([os.path.join(BIN, "aku_adapter_check"), "modules", cfg, wav, out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with wave.open(wav, "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), "<i2")
    ft = capi.Feat.from_file(cfg)
    lines = open(out).read().splitlines()
    pos = 0
    for name, typ in ft.modules():
        head = lines[pos].split()
        assert head[:3] == ["module", name, typ] and int(head[3]) == ft.module_dim(name)
        want = ft.run(pcm, -3, 24, module=name, dtype=np.float64)
        got = np.array([[float(x) for x in lines[pos + 1 + k].split()] for k in range(24)])
        assert np.array_equal(got, want), name
        pos += 25
    assert lines[pos] == "unknown 1"
    pos += 1
    # the normalization module's parameters, shifted by the view and read back ("%g" round trip)
    norm = [n for n, t in ft.modules() if t == "normalization"]
    assert norm and lines[pos].startswith("params " + norm[0])
    block = ft.get_parameters(norm[0])
    mean = [np.float32(x) + np.float32(0.25) for x in block.split("mean", 1)[1].split("\n")[0].split()]
    shifted = "{\n  mean " + " ".join("%g" % m for m in mean) + "\n" + block.split("\n", 2)[2]
    ft.set_parameters(norm[0], shifted)
    assert "  mean " + " ".join("%g" % m for m in mean) in "\n".join(lines[pos:pos + 4])
    end = [i for i in range(pos, len(lines)) if lines[i] == "}"][0]
    got = np.array([[float(x) for x in l.split()] for l in lines[end + 1:end + 7]])
    assert np.array_equal(got, ft.run(pcm, 0, 6, dtype=np.float64))


def test_pool_likelihood_view(capi, oracle, world):
    """PDFPool::compute_likelihood / compute_log_likelihood for single Gaussians through the HmmSet
    adapter (one device pass per frame) against the oracle's per-Gaussian log-likelihoods."""
    out = str(world["dir"] / "pool.txt")
    r = subprocess.run([os.path.join(BIN, "aku_adapter_check"), "pool", world["cfg"], world["base"],
                        str(world["dir"] / "a1.wav"), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().split("\n")
    G = int(lines[0])
    assert G == 256
    vals = np.array([[float(x) for x in l.split()] for l in lines[1:1 + 2 * G]])
    fea = world["ft"].run(world["pcms"][1], 0, 8, dtype=np.float64)
    om = oracle.DiagModel(*world["model"])
    for k, f in enumerate((0, 7)):
        ref = om.gauss_loglik(fea[f:f + 1])[0][::-1]
        got = vals[k * G:(k + 1) * G]
        assert np.abs(got[:, 0] - ref).max() <= 1e-4
        assert np.allclose(got[:, 1], np.exp(got[:, 0]), rtol=1e-6)


def test_distributions_surface(capi, oracle, world):
    """HmmSet::get_emission_pdf / get_pool_pdf / get_pool and PDF / Mixture / PDFPool::
    compute_likelihood(const Vector&) (aku/HmmSet.hh:216-228, aku/Distributions.hh:66-69,145,
    872-873) called the way aku/MllrTrainer.cc:40-56 and aku/logl.cc:58-60 do: mixture value =
    state_likelihood = weighted sum of its Gaussians' values = the oracle's, for a block frame
    and for a free-standing copy of it."""
    out = str(world["dir"] / "dist.txt")
    r = subprocess.run([os.path.join(BIN, "aku_adapter_check"), "dist", world["cfg"], world["base"],
                        str(world["dir"] / "a1.wav"), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().split("\n")
    S, G = (int(x) for x in lines[0].split())
    assert (S, G) == (32, 256)
    mean, var, off, idx, w = world["model"]
    fea = world["ft"].run(world["pcms"][1], 0, 12, dtype=np.float64)
    om = oracle.DiagModel(mean, var, off, idx, w)
    pos = 1
    for f in (3, 11):
        vals = np.array([[float(x) for x in l.split()] for l in lines[pos:pos + S]])
        ll_ref, lik_ref = om.score(fea[f:f + 1], want_lik=True)
        lik_ref = np.maximum(lik_ref[0], 1e-50)
        assert np.allclose(vals[:, 0], lik_ref, rtol=1.0001e-4, atol=0)  # = 1e-4 on the logarithm; Mixture::compute_likelihood
        assert np.abs(vals[:, 1] - np.log(lik_ref)).max() <= 1e-4     # compute_log_likelihood
        assert np.array_equal(vals[:, 0], vals[:, 2])                 # == HmmSet::state_likelihood
        assert np.allclose(vals[:, 3], lik_ref, rtol=1.0001e-4, atol=0)  # sum_k w_k N_k by hand
        assert np.allclose(vals[:, 4], vals[:, 0], rtol=1e-5)         # a copy of the vector, scored alone
        g0, m0, c_last, lik_g0 = lines[pos + S].split()
        assert int(g0) == idx[off[0]] and float(m0) == mean[int(g0), 0] and float(c_last) == var[int(g0), -1]
        assert abs(np.log(float(lik_g0)) - om.gauss_loglik(fea[f:f + 1])[0][int(g0)]) <= 1e-4
        pos += S + 1


@pytest.mark.gpu
def test_python_pptoolbox_module(capi, world, tmp_path):
    """The reference's SWIG module (aku/swig/PPToolbox.i:57-75) as a Python module on the C ABI:
    `import PPToolbox` from the package directory, the reference's call sequence, files and
    descriptors; the LNA image equals the engine's own run_utterance bytes (2-byte, normalised, as
    aku/PhoneProbsToolbox.cc:84-131 hard-wires -- that path is compared with the oracle in
    test_pipeline_gpu.py); errors surface as RuntimeError."""
    import importlib, sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aaltoasr_amd")
    sys.path.insert(0, pkg)
    try:
        PPToolbox = importlib.import_module("PPToolbox")
    finally:
        sys.path.remove(pkg)
    wav = str(world["dir"] / "a1.wav")
    t = PPToolbox.PPToolbox()
    with pytest.raises(RuntimeError):
        t.generate(wav, str(tmp_path / "x.lna"), False)      # nothing loaded yet
    t.read_configuration(world["cfg"])
    t.read_models(world["base"])
    out = str(tmp_path / "a.lna")
    t.generate(wav, out, False)
    want, frames = capi.run_utterance(world["ft"], world["gm"], world["pcms"][1], 0, 0, True, 2)
    assert open(out, "rb").read() == want and frames > 50
    # descriptors: audio in through one, LNA out through another; neither is closed by the call
    fd_in = os.open(wav, os.O_RDONLY)
    out2 = str(tmp_path / "b.lna")
    fd_out = os.open(out2, os.O_WRONLY | os.O_CREAT, 0o644)
    try:
        t.generate_to_fd(fd_in, fd_out, False)
        os.fstat(fd_in), os.fstat(fd_out)
    finally:
        os.close(fd_in)
        os.close(fd_out)
    assert open(out2, "rb").read() == want
    # clustering through the facade (aku/PhoneProbsToolbox.cc:50-53)
    gcl = str(tmp_path / "c.gcl")
    from oracle import oracle as O
    O.write_gcl(gcl, 16, synth.make_clustering(world["model"][0], 16))
    t.set_clustering(gcl, 0.2, 0.2)
    t.generate(wav, out, False)
    g2 = capi.Gmm.from_files(world["base"] + ".gk", world["base"] + ".mc", world["base"] + ".ph")
    g2.read_clustering(gcl)
    g2.set_clustering_min_evals(0.2, 0.2)
    want_c, _ = capi.run_utterance(world["ft"], g2, world["pcms"][1], 0, 0, True, 2)
    assert open(out, "rb").read() == want_c and want_c != want
    with pytest.raises(RuntimeError):
        t.read_models(str(tmp_path / "missing"))
    with pytest.raises(RuntimeError):
        t.generate(str(tmp_path / "missing.wav"), out, False)
    with pytest.raises(RuntimeError):
        t.read_configuration(str(tmp_path / "missing.cfg"))


@pytest.mark.gpu
def test_stream_input_is_consumed_incrementally(capi, world, tmp_path):
    """decode-stream.cc's reading loop (gen.open(stdin, true, true); generate(f) until eof(),
    decoder/decode-stream.cc:81, 238-276) on the adapter: raw PCM16 arrives through a pipe in two
    parts; frames of the first part come out while the second has not been written yet (the
    reference's AudioReader reads a non-seekable stream as frames ask for samples,
    aku/AudioReader.cc:112-142, 170-213), and every frame equals the whole-file result bit for bit,
    eof() on the same frame."""
    import select, time
    pcm = world["pcms"][0]                      # 5 s
    out = str(tmp_path / "stream.f64")
    p = subprocess.Popen([os.path.join(BIN, "aku_adapter_check"), "stream", world["cfg"], out],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        half = 16000 * 2
        p.stdin.write(pcm[:half].astype("<i2").tobytes())
        p.stdin.flush()
        seen = b""
        deadline = time.time() + 240
        # frames whose window and look-ahead lie inside the first 2 s must appear without more input
        while seen.count(b"\n") < 100 and time.time() < deadline:
            r, _, _ = select.select([p.stdout], [], [], 1.0)
            if r:
                chunk = os.read(p.stdout.fileno(), 65536)
                if not chunk:
                    break
                seen += chunk
        first_part = seen.count(b"\n")
        assert first_part >= 100, (first_part, p.poll())
        assert p.poll() is None                 # still waiting for the rest of the stream
        p.stdin.write(pcm[half:].astype("<i2").tobytes())
        p.stdin.close()
        rest, err = p.stdout.read(), p.stderr.read()
        assert p.wait(timeout=120) == 0, err.decode()
    finally:
        if p.poll() is None:
            p.kill()
    total = (seen + rest).count(b"\n")
    ft = world["ft"]
    last = ft.last_frame(len(pcm))
    assert total == last + 1                    # eof() turned true on frame last_frame + 1
    assert first_part < total
    got = np.fromfile(out, np.float64).reshape(total, ft.dim)
    want = ft.run(pcm, 0, total, dtype=np.float64)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("seed,block", [(1, 7), (2, 64), (3, 1), (4, 2048)])
def test_per_frame_api_in_random_order(capi, world, seed, block):
    """The adapters' per-frame surface driven out of order (aku_adapter_check random): frames forwards,
    backwards, far apart, before 0 and past the end of the file; eager and lazy likelihood calls; a module
    tap in between -- with adapter blocks of 1 / 7 / 64 / 2048 frames.  Every value equals the batch entry
    points' (features bit for bit; likelihood = exp of the batch log-likelihood as the adapter returns it)."""
    out = str(world["dir"] / ("random_%d.txt" % seed))
    wav = str(world["dir"] / "a2.wav")                     # 30 000 samples: 233 frames
    r = subprocess.run([os.path.join(BIN, "aku_adapter_check"), "random", world["cfg"], world["base"], wav, out,
                        str(seed), "600", str(block)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    pcm = world["pcms"][2]
    ft, gm = world["ft"], world["gm"]
    eof = ft.eof_frame(len(pcm))
    rows = [line.split() for line in open(out)]
    lo, hi = min(int(t[1]) for t in rows), max(int(t[1]) for t in rows) + 1
    assert hi > eof + 2
    fea = ft.run(pcm, lo, hi - lo, dtype=np.float64)
    tap = ft.run(pcm, lo, hi - lo, module="mfcc", dtype=np.float64)
    ll = gm.score(fea.astype(np.float32))
    seen = {"F": 0, "E": 0, "L": 0, "T": 0}
    for t in rows:
        seen[t[0]] += 1
        f = int(t[1])
        if t[0] == "F":
            assert int(t[2]) == (1 if f >= eof else 0), t[:3]
            got = np.array([float.fromhex(x) for x in t[3:]])
            assert np.array_equal(got, fea[f - lo]), (f, np.abs(got - fea[f - lo]).max())
        elif t[0] == "T":
            got = np.array([float.fromhex(x) for x in t[2:]])
            assert np.array_equal(got, tap[f - lo]), f
        elif t[0] == "E":
            got = np.array([float.fromhex(x) for x in t[2:]])
            assert np.array_equal(got, np.array([math.exp(float(v)) for v in ll[f - lo]])), f   # libm's exp, as std::exp
        else:
            s = int(t[2])
            assert float.fromhex(t[3]) == math.exp(float(ll[f - lo, s])), (f, s)
    assert seen["F"] == 600 and min(seen.values()) > 100
