"""Speaker / utterance configuration: aasr_spkc_* and
phone_probs -S against the oracle's restatement of aku::SpeakerConfig
(aku/SpeakerConfig.cc) driving the oracle feature chain and AdaptedGaussian
scoring.  Same tolerances as the rest: 5e-6 on features, 1e-4 on state
log-likelihoods."""
import os
import subprocess
import wave

import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aaltoasr_amd", "lib", "bin")

CFG = """module
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
  name mfcc
  type dct
  dim 12
  sources mel
}
module
{
  name d1
  type delta
  sources mfcc
}
module
{
  name merged
  type merge
  sources mfcc d1
}
module
{
  name norm
  type normalization
  sources merged
}
module
{
  name mllr
  type lin_transform
  sources norm
}
"""
D = 24


def _nums(rng, n, lo, hi, digits=6):
    return " ".join(("%%.%dg" % digits) % v for v in rng.uniform(lo, hi, n))


def _matrix_text(rng, digits=6):
    a = np.eye(D) * rng.uniform(0.95, 1.05, D) + 0.01 * rng.standard_normal((D, D))
    b = 0.2 * rng.standard_normal(D)
    w = np.hstack([b[:, None], a])
    return " ".join(("%%.%dg" % digits) % v for v in w.ravel()), w


def _spkc(digits=6):
    rng = np.random.default_rng(77)
    spk = {}
    text = "speaker default\n{\n  vtln\n  {\n  }\n  feature norm\n  {\n  }\n  feature mllr\n  {\n  }\n}\n"
    for name, warp in (("anna", 1.08), ("bert", 0.93)):
        wtxt, _ = _matrix_text(rng, digits)
        mean = _nums(rng, D, -1, 1, digits)
        scale = _nums(rng, D, 0.9, 1.1, digits)
        lin = " ".join(("%%.%dg" % digits) % v for v in (np.eye(D) + 0.01 * rng.standard_normal((D, D))).ravel())
        text += ("speaker %s\n{\n  feature vtln\n  {\n    warp_factor %g\n  }\n  norm\n  {\n    mean %s\n    scale %s\n  }\n"
                 "  feature mllr\n  {\n    matrix %s\n    bias %s\n  }\n  model cmllr\n  {\n    unitmode UNIT_NO\n    w1 %s\n  }\n}\n"
                 % (name, warp, mean, scale, lin, _nums(rng, D, -0.1, 0.1, digits), wtxt))
    # regression-class style speaker: two mixture groups + one phone group
    w1, _ = _matrix_text(rng)
    w2, _ = _matrix_text(rng)
    text += ("speaker carl\n{\n  model cmllr\n  {\n    unitmode UNIT_MIX\n    w1 0 1 2 3 4 5 6 7 %s\n    w2 6 7 8 9 10 %s\n  }\n}\n"
             % (w1, w2))
    text += "utterance default\n{\n  vtln\n  {\n  }\n}\nutterance u7\n{\n  vtln\n  {\n    warp_factor 1.02\n  }\n}\n"
    return text


def _write_wav(path, pcm, rate=16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(pcm.astype("<i2").tobytes())


@pytest.fixture(scope="module")
def world(capi, oracle, tmp_path_factory):
    d = tmp_path_factory.mktemp("spk")
    model = list(synth.make_model(D=D, G=128, S=16, comps=8, var_lo=1.0, var_hi=6.0))
    # put the synthetic pool where this chain's features live
    fea = oracle.FeatureChain(CFG).generate(synth.make_audio(24000, seed=90), 0, 180)
    mu, sd = fea.mean(0), fea.std(0) + 1e-3
    model[0] = mu + 0.7 * sd * model[0]
    model[1] = sd * sd * model[1]
    base = str(d / "m")
    oracle.write_gk(base + ".gk", model[0], model[1])
    oracle.write_mc(base + ".mc", model[2], model[3], model[4])
    oracle.write_ph(base + ".ph", 16)
    mean, var = oracle.read_gk(base + ".gk")
    cfg_path = str(d / "f.cfg")
    open(cfg_path, "w").write(CFG)
    spkc_path = str(d / "s.spkc")
    open(spkc_path, "w").write(_spkc())
    pcms = [synth.make_audio(n, seed=90 + i) for i, n in enumerate([24000, 20000, 30000, 18000, 26000])]
    # (speaker, utterance) per recipe line: repeated and interleaved speakers, an unknown
    # speaker (gets the defaults), an utterance entry
    who = [("anna", ""), ("anna", ""), ("bert", "u7"), ("zoe", ""), ("anna", "")]
    lines = []
    for i, p in enumerate(pcms):
        _write_wav(str(d / ("a%d.wav" % i)), p)
        l = "audio=%s lna=%s speaker=%s" % (d / ("a%d.wav" % i), d / ("a%d.lna" % i), who[i][0])
        if who[i][1]:
            l += " utterance=" + who[i][1]
        lines.append(l)
    recipe = str(d / "r.recipe")
    open(recipe, "w").write("\n".join(lines) + "\n")
    return dict(dir=d, base=base, cfg=cfg_path, spkc=spkc_path, pcms=pcms, who=who, recipe=recipe,
                model=(mean, var, model[2], model[3], model[4]))


def _oracle_lna(oracle, world, spkc_text=None, recipe=None):
    """What the reference loop produces for every recipe line (4-byte LNA values)."""
    ch = oracle.FeatureChain(CFG)
    om = oracle.DiagModel(*world["model"])
    sc = oracle.SpeakerConfig(ch, om)
    sc.read_text(spkc_text if spkc_text is not None else open(world["spkc"]).read())
    outs = []
    # Recipe::read keeps a key's value on later lines that do not repeat it
    # (aku/Recipe.cc:31,82-90): the utterance=u7 of line 3 also applies to lines 4 and 5
    infos = oracle.recipe_read(open(recipe or world["recipe"]).read())
    assert [i.utterance_id for i in infos] == (["", "", "u7", "u7", "u7"] if recipe is None else [""] * 5)
    for pcm, info in zip(world["pcms"], infos):
        spk, utt = info.speaker_id, info.utterance_id
        sc.set_speaker(spk)
        if utt:
            sc.set_utterance(utt)
        n = ch.num_frames(len(pcm))
        fea = ch.generate(pcm, 0, n)
        ll = oracle.score_adapted(om, fea, sc.g2t, sc.W)
        outs.append((fea, ll))
    return outs


def test_recipe_with_speakers_matches_oracle(capi, oracle, world):
    want = _oracle_lna(oracle, world)
    ft = capi.Feat.from_file(world["cfg"])
    gm = capi.Gmm.from_files(world["base"] + ".gk", world["base"] + ".mc", world["base"] + ".ph")
    sc = capi.SpeakerConfig(ft, gm)
    sc.read_file(world["spkc"])
    out = world["dir"] / "eng"
    os.makedirs(out)
    st = capi.run_recipe(ft, gm, world["recipe"], lnabytes=4, normalize=False, afname=True,
                         out_dir=str(out), speakers=sc)
    assert st.utterances == 5
    for i, (fea, ll) in enumerate(want):
        got = oracle.lna_decode(open(out / ("a%d.lna" % i), "rb").read())
        assert got.shape == ll.shape
        ok = ll > -85
        assert ok.mean() > 0.2 and np.abs(got - ll)[ok].max() <= 1e-4, i
    # lines 0 and 1 share every setting (the file is written with 6 digits, so the "%g"
    # read-back changes nothing): they stayed in one device batch
    assert sc.num_changes > 0


def test_direct_calls_and_feature_parity(capi, oracle, world):
    ch = oracle.FeatureChain(CFG)
    om = oracle.DiagModel(*world["model"])
    osc = oracle.SpeakerConfig(ch, om)
    osc.read_text(open(world["spkc"]).read())
    ft = capi.Feat.from_file(world["cfg"])
    gm = capi.Gmm.from_files(world["base"] + ".gk", world["base"] + ".mc", world["base"] + ".ph")
    sc = capi.SpeakerConfig(ft)          # phone_probs reads the speaker file before the model
    sc.read_file(world["spkc"])
    sc.set_model(gm)
    pcm = world["pcms"][0]
    for spk, utt in [("bert", ""), ("carl", ""), ("carl", "u7"), ("", ""), ("anna", "x-unknown")]:
        osc.set_speaker(spk)
        sc.set_speaker(spk)
        if utt:
            osc.set_utterance(utt)
            sc.set_utterance(utt)
        fea = ch.generate(pcm, -3, 60)
        got = ft.run(pcm, -3, 60, dtype=np.float64)
        assert np.abs(got - fea).max() <= 5e-6, (spk, utt)
        ll = oracle.score_adapted(om, fea.astype(np.float32), osc.g2t, osc.W)
        assert np.abs(gm.score(fea.astype(np.float32)) - ll).max() <= 1e-4, (spk, utt)
    # carl: mixtures 6 and 7 are claimed by both transforms; the later key ("6 7 8 9 10")
    # sorts after ("0 1 ... 7") and wins
    osc.set_speaker("carl")
    m = om
    assert set(osc.g2t[m.mix_idx[m.mix_off[6]:m.mix_off[8]]]) == {1}
    assert set(osc.g2t[m.mix_idx[m.mix_off[0]:m.mix_off[6]]]) == {0}
    assert (osc.g2t[m.mix_idx[m.mix_off[11]:]] == -1).all()


def test_percent_g_round_trip_of_a_repeated_speaker(capi, oracle, world):
    """Parameters with more than 6 significant digits: the second set_speaker of the same
    speaker applies the "%g" read-back (aku/SpeakerConfig.cc:242-243, 322-340)."""
    text = _spkc(digits=9)
    ch = oracle.FeatureChain(CFG)
    osc = oracle.SpeakerConfig(ch, oracle.DiagModel(*world["model"]))
    osc.read_text(text)
    ft = capi.Feat.from_file(world["cfg"])
    gm = capi.Gmm.from_files(world["base"] + ".gk", world["base"] + ".mc", world["base"] + ".ph")
    sc = capi.SpeakerConfig(ft, gm)
    sc.read_text(text)
    pcm = world["pcms"][1]
    feats = []
    for k in range(3):
        osc.set_speaker("anna")
        sc.set_speaker("anna")
        want = ch.generate(pcm, 0, 40)
        got = ft.run(pcm, 0, 40, dtype=np.float64)
        assert np.abs(got - want).max() <= 5e-6, k
        feats.append(got)
    assert np.abs(feats[1] - feats[0]).max() > 1e-7     # the round trip moved the parameters
    assert np.array_equal(feats[2], feats[1])            # and is idempotent afterwards


def test_errors_follow_the_reference(capi, world):
    ft = capi.Feat.from_file(world["cfg"])
    sc = capi.SpeakerConfig(ft)
    with pytest.raises(capi.AasrError, match="Syntax error on line 1"):
        sc.read_text("speakers anna\n{\n}\n")
    with pytest.raises(capi.AasrError, match="unknown module requested: nosuch"):
        sc.read_text("speaker a\n{\n  nosuch\n  {\n  }\n}\n")
    with pytest.raises(capi.AasrError, match="Unknown module namespace"):
        sc.read_text("speaker a\n{\n  decoder x\n  {\n  }\n}\n")
    with pytest.raises(capi.AasrError, match="unknown model module requested"):
        sc.read_text("speaker a\n{\n  model mllr\n  {\n  }\n}\n")
    sc2 = capi.SpeakerConfig(ft)
    sc2.read_text("speaker a\n{\n  norm\n  {\n    mean %s\n    var %s\n  }\n}\n"
                  % (" ".join(["0"] * D), " ".join(["4"] * D)))
    with pytest.raises(capi.AasrError, match="needs a default speaker"):
        sc2.set_speaker("")
    with pytest.raises(capi.AasrError, match="Unknown speaker b, and default speaker settings are missing"):
        sc2.set_speaker("b")
    sc2.set_speaker("a")
    # the read-back added `scale` next to the file's `var`: the reference refuses the block
    with pytest.raises(capi.AasrError, match="Both scale and var"):
        sc2.set_speaker("a")
    with pytest.raises(capi.AasrError, match="Default utterance is required"):
        capi.SpeakerConfig(ft).set_utterance("")


def test_phone_probs_cli_speakers(capi, oracle, world):
    out = world["dir"] / "cli"
    os.makedirs(out)
    r = subprocess.run([os.path.join(BIN, "phone_probs"), "-b", world["base"], "-c", world["cfg"],
                        "-r", world["recipe"], "-a", "-o", str(out), "--lnabytes=4", "-N",
                        "-S", world["spkc"]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = _oracle_lna(oracle, world)
    for i, (fea, ll) in enumerate(want):
        got = oracle.lna_decode(open(out / ("a%d.lna" % i), "rb").read())
        ok = ll > -85
        assert np.abs(got - ll)[ok].max() <= 1e-4, i


def test_reference_style_loop_with_speaker_config(capi, oracle, world):
    """phone_probs.cc's calling sequence (read_speaker_file before the model, set_speaker /
    set_utterance before gen.open, then the per-frame loop) on the aku adapter classes."""
    out = str(world["dir"] / "loop.lna")
    r = subprocess.run([os.path.join(BIN, "aku_adapter_check"), world["cfg"], world["base"],
                        str(world["dir"] / "a2.wav"), out, "4", "spkc", world["spkc"], "bert", "u7"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ch = oracle.FeatureChain(CFG)
    om = oracle.DiagModel(*world["model"])
    sc = oracle.SpeakerConfig(ch, om)
    sc.read_text(open(world["spkc"]).read())
    sc.set_speaker("bert")
    sc.set_utterance("u7")
    pcm = world["pcms"][2]
    fea = ch.generate(pcm, 0, ch.num_frames(len(pcm)))
    lik = np.exp(oracle.score_adapted(om, fea, sc.g2t, sc.W))
    want, _ = oracle.lna_encode(lik, True, 4)
    got = oracle.lna_decode(open(out, "rb").read())
    ok = want > -60
    assert got.shape == want.shape and np.abs(got - want)[ok].max() <= 1e-4


def test_speakers_with_gaussian_clustering(capi, oracle, world, tmp_path):
    """phone_probs -S SPKC -C GCL, the adapted recognition pass of pyrectool (rectool.py:655-666):
    speakers with a global (UNIT_NO) model transform over a clustered pool, against the oracle's
    clustered + adapted scoring; a speaker with per-class transforms (regression classes) too."""
    mean, var, off, idx, w = world["model"]
    g2c = synth.make_clustering(mean, 12)
    pairs = [(int(g), int(c)) for g, c in enumerate(g2c)]
    ch = oracle.FeatureChain(CFG)
    om = oracle.DiagModel(mean, var, off, idx, w)
    om.set_clustering(12, pairs, 0.0, 0.25)
    osc = oracle.SpeakerConfig(ch, om)
    osc.read_text(open(world["spkc"]).read())
    ft = capi.Feat.from_file(world["cfg"])
    gm = capi.Gmm.from_files(world["base"] + ".gk", world["base"] + ".mc", world["base"] + ".ph")
    gm.set_clustering(12, pairs)
    gm.set_clustering_min_evals(0.0, 0.25)
    sc = capi.SpeakerConfig(ft, gm)
    sc.read_file(world["spkc"])
    lines = []
    for i, spk in enumerate(["anna", "bert", "anna"]):
        lines.append("audio=%s lna=c%d.lna speaker=%s" % (world["dir"] / ("a%d.wav" % i), i, spk))
    recipe = str(tmp_path / "c.recipe")
    open(recipe, "w").write("\n".join(lines) + "\n")
    st = capi.run_recipe(ft, gm, recipe, lnabytes=4, normalize=False, out_dir=str(tmp_path), speakers=sc)
    assert st.utterances == 3
    for i, spk in enumerate(["anna", "bert", "anna"]):
        osc.set_speaker(spk)
        pcm = world["pcms"][i]
        n = ch.num_frames(len(pcm))
        fea = ch.generate(pcm, 0, n)
        assert len(osc.W) == 1 and (np.asarray(osc.g2t) == 0).all()
        ll = om.score_clustered_adapted(fea, np.asarray(osc.W[0]))
        got = oracle.lna_decode(open(tmp_path / ("c%d.lna" % i), "rb").read())
        ok = ll > -85
        assert ok.mean() > 0.2 and np.abs(got - ll)[ok].max() <= 1e-4, (i, spk)
    # "carl" has per-class transforms (regression classes): the clustered pass takes them as well
    sc.set_speaker("carl")
    osc.set_speaker("carl")
    assert len(osc.W) > 1 or not (np.asarray(osc.g2t) == 0).all()
    pcm = world["pcms"][1]
    fea = ch.generate(pcm, 0, ch.num_frames(len(pcm)))
    want = om.score_clustered_classes(fea, np.asarray(osc.g2t), np.asarray(osc.W))
    got = gm.score(fea.astype(np.float32))
    vis = want > -85
    assert np.abs(got - want)[vis].max() <= 1e-4


def test_module_classes_set_parameters_directly(capi, oracle, world, tmp_path):
    """NormalizationModule::set_normalization, LinTransformModule::set_transformation_matrix / _bias,
    VtlnModule::set_warp_factor (aku/FeatureModules.cc:1123-1133, 1273-1322, 1603-1612) reached
    through dynamic_cast as aku/feanorm.cc and aku/vtln.cc do: the generator's output afterwards is
    the oracle chain's with the same float parameters; an empty matrix / bias restores the identity."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aaltoasr_amd", "lib", "bin",
                       "aku_adapter_check")
    cfg = str(tmp_path / "f.cfg")
    open(cfg, "w").write(CFG)
    pcm = synth.make_audio(16000, seed=5)
    wav = str(tmp_path / "a.wav")
    import wave
    with wave.open(wav, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.astype("<i2").tobytes())
    out = str(tmp_path / "o.txt")
    r = subprocess.run([exe, "setters", cfg, wav, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().splitlines()
    head = lines[0].split()
    assert head[0] == "1" and abs(float(head[1]) - float(np.float32(1.0) + np.float32(0.03))) < 1e-7
    assert int(head[2]) == D * D and int(head[3]) == D
    i = np.arange(D, dtype=np.float32)
    mean = np.float32(0.1) * i - np.float32(0.7)
    scale = np.float32(1.0) / (np.float32(1.0) + np.float32(0.03) * i)
    bias = np.float32(0.01) * (np.arange(D) % 5).astype(np.float32) - np.float32(0.02)
    mat = np.eye(D, dtype=np.float32) + np.float32(0.001) * ((np.arange(D)[:, None] * 7 + np.arange(D)[None, :] * 3) % 11).astype(np.float32)
    vec = lambda v: " ".join("%.9g" % x for x in np.asarray(v, np.float32).ravel())
    chain = oracle.FeatureChain(CFG)
    chain.set_parameters("norm", {"mean": vec(mean), "scale": vec(scale)})
    chain.set_parameters("mllr", {"matrix": vec(mat), "bias": vec(bias)})
    chain.set_parameters("vtln", {"warp_factor": "%.9g" % (np.float32(1.0) + np.float32(0.03))})
    want = chain.generate(pcm, 0, 8)
    got = np.array([[float(x) for x in l.split()] for l in lines[1:9]])
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    chain.set_parameters("mllr", {})
    want2 = chain.generate(pcm, 0, 2)
    got2 = np.array([[float(x) for x in l.split()] for l in lines[9:11]])
    assert np.abs(got2 - want2).max() <= 1e-5 * max(1.0, np.abs(want2).max())
    assert np.abs(got2 - got[:2]).max() > 1e-3


SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spkc")


@pytest.mark.parametrize("name", ["default_vtln.spkc", "default_mllr.spkc", "default_vtln+mllr.spkc",
                                  "aku_scripts_vtln_default.spkc"])
def test_speaker_files_the_reference_ships(capi, oracle, world, name, tmp_path):
    """The speaker files that come with the reference (pyrectool/default_vtln.spkc, default_mllr.spkc,
    default_vtln+mllr.spkc, aku/scripts/vtln_default.spkc; copied as data to tests/golden/spkc): a `speaker default` entry
    with empty `feature vtln { }` / `feature mllr { }` / bare `vtln { }` blocks -- what pyrectool hands to phone_probs -S
    before any adaptation has been estimated.  Every recipe speaker is unknown and takes the default entry, an empty block
    resets its module (warp factor 1, identity transform), so the run has to give the oracle's SpeakerConfig loop on the same
    file AND the unadapted run, file for file."""
    text = open(os.path.join(SHIPPED, name)).read()
    # the files hold no utterance entries: a recipe that names an utterance is refused, as in the reference
    # ("Unknown utterance u7, and default utterance settings are missing", aku/SpeakerConfig.cc), so the recipe here
    # carries speaker keys only
    recipe = str(tmp_path / "r.recipe")
    open(recipe, "w").write("".join("audio=%s lna=%s speaker=%s\n" % (world["dir"] / ("a%d.wav" % i), world["dir"] / ("a%d.lna" % i), w[0])
                                    for i, w in enumerate(world["who"])))
    want = _oracle_lna(oracle, world, spkc_text=text, recipe=recipe)
    ft = capi.Feat.from_file(world["cfg"])
    gm = capi.Gmm.from_files(world["base"] + ".gk", world["base"] + ".mc", world["base"] + ".ph")
    sc = capi.SpeakerConfig(ft, gm)
    sc.read_file(os.path.join(SHIPPED, name))
    out = tmp_path / "spk"
    os.makedirs(out)
    st = capi.run_recipe(ft, gm, recipe, lnabytes=4, normalize=False, afname=True, out_dir=str(out), speakers=sc)
    assert st.utterances == 5
    with pytest.raises(capi.AasrError, match="Unknown utterance u7, and default utterance settings are missing"):
        capi.run_recipe(ft, gm, world["recipe"], lnabytes=4, normalize=False, afname=True, out_dir=str(out), speakers=sc)
    plain = tmp_path / "plain"
    os.makedirs(plain)
    ft2 = capi.Feat.from_file(world["cfg"])
    gm2 = capi.Gmm.from_files(world["base"] + ".gk", world["base"] + ".mc", world["base"] + ".ph")
    capi.run_recipe(ft2, gm2, recipe, lnabytes=4, normalize=False, afname=True, out_dir=str(plain))
    for i, (fea, ll) in enumerate(want):
        raw = open(out / ("a%d.lna" % i), "rb").read()
        got = oracle.lna_decode(raw)
        assert got.shape == ll.shape
        ok = ll > -85
        assert ok.mean() > 0.2 and np.abs(got - ll)[ok].max() <= 1e-4, (name, i)
        assert raw == open(plain / ("a%d.lna" % i), "rb").read(), (name, i)
    # the same file through the command-line tool, as pyrectool passes it (rectool.py:655-666)
    cli = tmp_path / "cli"
    os.makedirs(cli)
    r = subprocess.run([os.path.join(BIN, "phone_probs"), "-b", world["base"], "-c", world["cfg"], "-r", recipe,
                        "-a", "-o", str(cli), "--lnabytes=4", "-N", "-i", "1", "-S", os.path.join(SHIPPED, name)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for i in range(5):
        assert open(cli / ("a%d.lna" % i), "rb").read() == open(out / ("a%d.lna" % i), "rb").read(), (name, i)


def test_a_shipped_speaker_file_names_a_module_the_graph_lacks(capi, world):
    """aku/scripts/vtln_default.spkc against a feature graph without a `vtln` module: the reference throws
    "SpeakerConfig: unknown module requested: vtln" while reading the file (aku/SpeakerConfig.cc)."""
    cfg = CFG.replace('module\n{\n  name vtln\n  type vtln\n  sources fft\n}\n', '').replace("sources vtln", "sources fft")
    assert "vtln" not in cfg
    ft = capi.Feat(cfg)
    sc = capi.SpeakerConfig(ft)
    with pytest.raises(capi.AasrError, match="unknown module requested: vtln"):
        sc.read_file(os.path.join(SHIPPED, "aku_scripts_vtln_default.spkc"))
