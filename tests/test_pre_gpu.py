"""`pre` base module + the feacat tool (feature files in and
out).  The reference's own test scripts (aku/tests/*.script) are run with this
engine's feacat and compared with the reference's golden outputs (aku/tests/*.ref,
two decimals: 0.005), and the pre module is compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aaltoasr_amd", "lib", "bin")
FEACAT = os.path.join(BIN, "feacat")


def _text(out):
    return np.array([[float(v) for v in line.split()] for line in out.strip().split("\n")])


def test_reference_script_pre_test(golden_dir, tmp_path):
    """aku/tests/pre_test.script:
         feacat --start-frame 10 --end-frame 60 -c mfcc_p_dd.feaconf -H --raw-output short.wav > pre_test.tmp
         feacat -c pre.feaconf pre_test.tmp"""
    tmp = str(tmp_path / "pre_test.tmp")
    with open(tmp, "wb") as f:
        r = subprocess.run([FEACAT, "--start-frame", "10", "--end-frame", "60", "-c",
                            os.path.join(golden_dir, "mfcc_p_dd.feaconf"), "-H", "--raw-output",
                            os.path.join(golden_dir, "short.wav")], stdout=f, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr
    raw = open(tmp, "rb").read()
    assert np.frombuffer(raw[:4], "=i4")[0] == 39 and len(raw) == 4 + 51 * 39 * 4
    r = subprocess.run([FEACAT, "-c", os.path.join(golden_dir, "pre.feaconf"), tmp],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = _text(r.stdout)
    ref = np.loadtxt(os.path.join(golden_dir, "pre_test.ref"))
    assert got.shape == ref.shape == (51, 39)
    assert np.abs(got - ref).max() <= 0.005 + 1e-4
    # the file carries float32: the pre chain prints the float-rounded source features
    r2 = subprocess.run([FEACAT, "-s", "10", "-e", "60", "-c", os.path.join(golden_dir, "mfcc_p_dd.feaconf"),
                         os.path.join(golden_dir, "short.wav")], capture_output=True, text=True, timeout=300)
    src = _text(r2.stdout)
    assert np.abs(src - got).max() <= 1.01e-4 and np.array_equal(
        got, np.round(np.frombuffer(raw[4:], "<f4").reshape(51, 39).astype(np.float64), 4))


@pytest.mark.parametrize("name,args", [("mfcc_p_dd", ["--start-frame", "-10", "--end-frame", "80"]),
                                       ("mfcc_cms_norm", ["-s", "-15", "-e", "90"])])
def test_reference_scripts_from_stdin(golden_dir, name, args):
    """aku/tests/mfcc_p_dd.script (second command) and mfcc_cms_norm.script:
         cat short.wav | feacat <range> -c <name>.feaconf -"""
    wav = open(os.path.join(golden_dir, "short.wav"), "rb").read()
    r = subprocess.run([FEACAT] + args + ["-c", os.path.join(golden_dir, name + ".feaconf"), "-"],
                       input=wav, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = _text(r.stdout.decode())
    ref = np.loadtxt(os.path.join(golden_dir, name + ".ref"))
    # mfcc_p_dd.script runs feacat twice (the second time on the configuration the first
    # run wrote with --write-config), so its .ref holds the same 91 frames twice
    assert len(ref) % len(got) == 0
    for part in np.split(ref, len(ref) // len(got)):
        assert np.abs(got - part).max() <= 0.005 + 1e-4


def test_feacat_reverse_order_and_errors(golden_dir, tmp_path):
    cfg = os.path.join(golden_dir, "mfcc_p_dd.feaconf")
    wav = os.path.join(golden_dir, "short.wav")
    fwd = subprocess.run([FEACAT, "-s", "3", "-e", "9", "-c", cfg, wav], capture_output=True, text=True)
    rev = subprocess.run([FEACAT, "-s", "9", "-e", "3", "-c", cfg, wav], capture_output=True, text=True)
    assert fwd.returncode == 0 and rev.returncode == 0
    assert len(fwd.stdout.splitlines()) == 7 and rev.stdout.splitlines() == fwd.stdout.splitlines()[::-1]
    bad = str(tmp_path / "bad.fea")
    open(bad, "wb").write(np.int32(12).tobytes() + np.zeros(24, np.float32).tobytes())
    r = subprocess.run([FEACAT, "-c", os.path.join(golden_dir, "pre.feaconf"), bad], capture_output=True, text=True)
    assert r.returncode != 0 and "The file has invalid dimension" in r.stderr
    # -G adds N(0, std) noise to every printed value (aku/feacat.cc:38-42)
    clean = np.array([[float(x) for x in l.split()] for l in
                      subprocess.run([FEACAT, "-c", cfg, wav], capture_output=True, text=True).stdout.splitlines()])
    r = subprocess.run([FEACAT, "-c", cfg, "-G", "0.1", wav], capture_output=True, text=True)
    assert r.returncode == 0
    noisy = np.array([[float(x) for x in l.split()] for l in r.stdout.splitlines()])
    assert noisy.shape == clean.shape and 0.07 < (noisy - clean).std() < 0.13


PRE_CHAIN = """module
{
  name pre
  type pre
  dim 13
  %s
}
module
{
  name d
  type delta
  sources pre
}
module
{
  name m
  type merge
  sources pre d
}
module
{
  name cms
  type mean_subtractor
  left 20
  right 10
  sources m
}
"""


@pytest.mark.parametrize("legacy", [0, 1])
def test_pre_module_matches_oracle(capi, oracle, tmp_path, legacy):
    cfg = PRE_CHAIN % ("legacy_file 1" if legacy else "frame_rate 100")
    rng = np.random.default_rng(8)
    frames = rng.standard_normal((57, 13)).astype(np.float32) * 3
    ch = oracle.FeatureChain(cfg)
    ft = capi.Feat(cfg)
    assert ft.last_frame(2 * frames.size) == ch.last_frame(frames.size) == 56
    for mod in ("pre", "d", "cms"):
        want = ch.generate(frames, -25, 110, module=mod)
        got = ft.run_features(frames, -25, 110, module=mod, dtype=np.float64)
        assert np.abs(got - want).max() <= (0 if mod == "pre" else 1e-12), mod
    # borders: before 0 the first frame, past the end the last one
    got = ft.run_features(frames, -3, 70, module="pre", dtype=np.float64)
    assert np.array_equal(got[:3], np.repeat(frames[:1].astype(np.float64), 3, 0))
    assert np.array_equal(got[60:], np.repeat(frames[-1:].astype(np.float64), 10, 0))
    with pytest.raises(capi.AasrError, match="not a pre module"):
        capi.Feat("module\n{\n name a\n type audiofile\n sample_rate 16000\n}\n").run_features(frames, 0, 1)


def test_recipe_on_feature_files(capi, oracle, tmp_path):
    """phone_probs with a pre base module: the recipe's audio= entries are feature files."""
    D = 13
    cfg = PRE_CHAIN % ""
    cfg_path = str(tmp_path / "pre.cfg")
    open(cfg_path, "w").write(cfg)
    model = synth.make_model(D=2 * D, G=64, S=8, comps=8)
    base = str(tmp_path / "m")
    oracle.write_gk(base + ".gk", model[0], model[1])
    oracle.write_mc(base + ".mc", model[2], model[3], model[4])
    oracle.write_ph(base + ".ph", 8)
    rng = np.random.default_rng(3)
    lines, feats = [], []
    for i, n in enumerate((40, 75)):
        f = rng.standard_normal((n, D)).astype(np.float32)
        oracle.write_feature_file(str(tmp_path / ("u%d.fea" % i)), f)
        feats.append(f)
        lines.append("audio=%s lna=%s" % (tmp_path / ("u%d.fea" % i), tmp_path / ("u%d.lna" % i)))
    recipe = str(tmp_path / "r.recipe")
    open(recipe, "w").write("\n".join(lines) + "\n")
    r = subprocess.run([os.path.join(BIN, "phone_probs"), "-b", base, "-c", cfg_path, "-r", recipe,
                        "--lnabytes=4", "-N"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ch = oracle.FeatureChain(cfg)
    om = oracle.read_model(base)
    for i, f in enumerate(feats):
        want = om.score(ch.generate(f, 0, len(f)))
        got = oracle.lna_decode(open(tmp_path / ("u%d.lna" % i), "rb").read())
        ok = want > -80
        assert got.shape == want.shape and np.abs(got - want)[ok].max() <= 1e-4


EXPECTED_MFCC_P_DD_CFG = """module
{
  name audiofile
  type audiofile
  pre_emph_coef 0.97
  sample_rate 16000
  frame_rate 125
  window_width 256
  copy_borders 1
}

module
{
  name fft
  type fft
  magnitude 1
  sources audiofile
}

module
{
  name mel
  type mel
  sources fft
}

module
{
  name power
  type power
  sources fft
}

module
{
  name mfcc
  type dct
  dim 12
  zeroth 0
  sources mel
}

module
{
  name mfcc_power
  type merge
  sources mfcc power
}

module
{
  name delta1
  type delta
  width 2
  normalization 10
  sources mfcc_power
}

module
{
  name delta2
  type delta
  width 2
  normalization 10
  sources delta1
}

module
{
  name final
  type merge
  sources mfcc_power delta1 delta2
}

"""


def test_write_config_and_the_two_commands_of_mfcc_p_dd_script(capi, golden_dir, tmp_path):
    """aku/tests/mfcc_p_dd.script:
         cat short.wav | feacat --start-frame -10 --end-frame 80 --write-config X -c mfcc_p_dd.feaconf -
         cat short.wav | feacat --start-frame -10 --end-frame 80 -c X -
    The configuration written by the first command (FeatureGenerator::write_configuration:
    every option a module saves, defaults made explicit, "sources" last) is what the reference's
    get_module_config functions produce for this graph, and the concatenated output of the two
    commands is the reference's mfcc_p_dd.ref."""
    wav = open(os.path.join(golden_dir, "short.wav"), "rb").read()
    tmp = str(tmp_path / "mfcc_p_dd.feaconf.tmp")
    r1 = subprocess.run([FEACAT, "--start-frame", "-10", "--end-frame", "80", "--write-config", tmp,
                         "-c", os.path.join(golden_dir, "mfcc_p_dd.feaconf"), "-"],
                        input=wav, capture_output=True, timeout=300)
    assert r1.returncode == 0, r1.stderr
    assert open(tmp).read() == EXPECTED_MFCC_P_DD_CFG
    r2 = subprocess.run([FEACAT, "--start-frame", "-10", "--end-frame", "80", "-c", tmp, "-"],
                        input=wav, capture_output=True, timeout=300)
    assert r2.returncode == 0, r2.stderr
    assert r1.stdout == r2.stdout                      # the written graph is the same graph
    got = _text((r1.stdout + r2.stdout).decode())
    ref = np.loadtxt(os.path.join(golden_dir, "mfcc_p_dd.ref"))
    assert got.shape == ref.shape == (182, 39)
    assert np.abs(got - ref).max() <= 0.005 + 1e-4
    # the C ABI gives the same text; writing is idempotent
    ft = capi.Feat.from_file(tmp)
    assert ft.write_config() == EXPECTED_MFCC_P_DD_CFG
    # normalisation / transform values survive the "%g" round trip of a written configuration
    cms = capi.Feat.from_file(os.path.join(golden_dir, "mfcc_cms_norm.feaconf"))
    again = capi.Feat(cms.write_config())
    assert again.write_config() == cms.write_config()
