"""The reference's OWN caller sources against the engine's aku-named adapter headers
(aaltoasr_amd/csrc/aku): not re-typed copies -- the text of aku/phone_probs.cc, aku/feacat.cc and
the acoustics part of decoder/decode-stream.cc is read from /root/reference and handed to the
compiler on stdin, so that `#include "FeatureGenerator.hh"` and friends resolve to the adapters.

* CPU (needs the reference tree, i.e. runs in the build container): every one of them compiles.
* GPU (uses oracle/_ref/{phone_probs,feacat}_refmain, the same sources LINKED with libaasr.so by
  oracle/Makefile in the build container; the binaries travel with the snapshot): the reference's
  main() running on the engine writes the same LNA files / feature dumps as the engine's own
  tools."""
import os
import subprocess
import wave

import numpy as np
import pytest

from conftest import observed

from aaltoasr_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
AKU = os.path.join(ROOT, "aaltoasr_amd", "csrc", "aku")
BIN = os.path.join(ROOT, "aaltoasr_amd", "lib", "bin")
REFBIN = os.path.join(ROOT, "oracle", "_ref")

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "aku")), reason="reference tree not present")


def _syntax_check(text, tmp_path):
    r = subprocess.run(["g++", "-std=gnu++17", "-fsyntax-only", "-x", "c++", "-I", AKU, "-"], input=text,
                       capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]


@needs_ref
@pytest.mark.parametrize("src", ["aku/phone_probs.cc", "aku/feacat.cc"])
def test_reference_tool_sources_compile_against_the_adapters(src, tmp_path):
    _syntax_check(open(os.path.join(REF, src)).read(), tmp_path)


@needs_ref
def test_decode_stream_acoustics_compile_against_the_adapters(tmp_path):
    """decoder/decode-stream.cc:34-117 and :177-207: includes and constants, initialize_acoustics
    (open(stdin, true, true), dynamic_cast<AudioFileModule*>, get_config / set_config, read_mc /
    read_ph / read_gk / read_clustering / set_clustering_min_evals), get_features, get_likelihoods.
    The decoder's Toolbox (lines 120-174, 209-276) is outside the hot path; its include is
    dropped and FeatureModules.hh -- which the sample forgets -- is added."""
    lines = open(os.path.join(REF, "decoder", "decode-stream.cc")).read().split("\n")
    body = "\n".join(lines[33:117] + lines[176:207])
    body = body.replace("#include <Toolbox.hh>", "#include <aku/FeatureModules.hh>\n#include <cmath>")
    os.makedirs(tmp_path / "inc")
    os.symlink(AKU, tmp_path / "inc" / "aku")
    r = subprocess.run(["g++", "-std=gnu++17", "-fsyntax-only", "-x", "c++", "-I", str(tmp_path / "inc"), "-"],
                       input=body, capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]


@needs_ref
def test_reference_mains_are_linked_with_the_engine(capi, oracle):
    for name in ("phone_probs_refmain", "feacat_refmain"):
        assert os.access(os.path.join(REFBIN, name), os.X_OK), name


# ------------------------------------------------------------------------------- GPU --

def _write_wav(path, pcm, rate=16000):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(pcm.astype("<i2").tobytes())


@pytest.fixture(scope="module")
def world(capi, oracle, tmp_path_factory):
    if not os.access(os.path.join(REFBIN, "phone_probs_refmain"), os.X_OK):
        pytest.skip("oracle/_ref/phone_probs_refmain was not built (no reference tree in the build container)")
    d = tmp_path_factory.mktemp("refmain")
    cfg = str(d / "f.cfg")
    open(cfg, "w").write(synth.make_feature_config())
    model = synth.make_model(D=39, G=512, S=40, comps_range=(3, 20), tied=True)
    base = str(d / "model")
    oracle.write_gk(base + ".gk", model[0], model[1])
    oracle.write_mc(base + ".mc", model[2], model[3], model[4])
    oracle.write_ph(base + ".ph", 40)
    gcl = str(d / "model.gcl")
    oracle.write_gcl(gcl, 24, synth.make_clustering(model[0], 24))
    lines = []
    for i, n in enumerate([48000, 16000, 30011, 9000]):
        _write_wav(str(d / ("a%d.wav" % i)), synth.make_audio(n, seed=60 + i))
        lines.append("audio=%s lna=a%d.lna%s" % (d / ("a%d.wav" % i), i, " start-time=0.504 end-time=1.2" if i == 2 else ""))
    recipe = str(d / "test.recipe")
    open(recipe, "w").write("\n".join(lines) + "\n")
    return dict(dir=d, cfg=cfg, base=base, gcl=gcl, recipe=recipe)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--lnabytes=2"], ["--lnabytes=4"], ["--lnabytes=4", "-N"],
                                   ["--lnabytes=2", "-C", "GCL", "--eval-ming", "0.25"], ["-B", "2", "-I", "2"]])
def test_reference_phone_probs_main_on_the_engine(world, extra):
    """aku/phone_probs.cc's main() -- its option table, recipe loop, per-frame generate /
    precompute_likelihoods / state_likelihood calls, float normalisation and LNA packing -- linked
    with the engine, against the engine's batched phone_probs: the same files."""
    extra = [world["gcl"] if a == "GCL" else a for a in extra]
    outs = []
    for tag, exe in (("ref", os.path.join(REFBIN, "phone_probs_refmain")), ("eng", os.path.join(BIN, "phone_probs"))):
        out = world["dir"] / (tag + "_" + "_".join(a.strip("-").replace("/", "_")[-12:] for a in extra))
        os.makedirs(out, exist_ok=True)
        r = subprocess.run([exe, "-b", world["base"], "-c", world["cfg"], "-r", world["recipe"], "-o", str(out)] + extra,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        outs.append(out)
    names = sorted(os.listdir(outs[1]))
    assert names and sorted(os.listdir(outs[0])) == names
    four = "--lnabytes=4" in extra
    for n in names:
        a = open(outs[0] / n, "rb").read()
        b = open(outs[1] / n, "rb").read()
        assert len(a) == len(b) and a[:5] == b[:5], n
        if four:
            # the reference main normalises on the host from exp(log-likelihood) of the engine's
            # float rows; the engine normalises on the device: same values to float rounding
            x = np.frombuffer(a[5:], "<f4").astype(np.float64)
            y = np.frombuffer(b[5:], "<f4").astype(np.float64)
            ok = (y > -80.0)
            assert np.abs(x - y)[ok].max() <= 2e-5, n
        else:
            x = np.frombuffer(a[5:], ">u2").astype(np.int64)
            y = np.frombuffer(b[5:], ">u2").astype(np.int64)
            assert np.abs(x - y).max() <= 1, n
            observed('refmain codes equal ' + n, float((x == y).mean()), 0.9995)  # observed 1.0


@pytest.mark.gpu
def test_reference_feacat_main_on_the_engine(world):
    """aku/feacat.cc's main() linked with the engine prints the same text and the same raw floats
    as the engine's feacat; -G adds noise of the requested size."""
    wav = str(world["dir"] / "a1.wav")
    for flags in ([], ["--raw-output", "-H"], ["-s", "-3", "-e", "20"], ["-s", "30", "-e", "10"]):
        a = subprocess.run([os.path.join(REFBIN, "feacat_refmain"), "-c", world["cfg"]] + flags + [wav],
                           capture_output=True, timeout=300)
        b = subprocess.run([os.path.join(BIN, "feacat"), "-c", world["cfg"]] + flags + [wav],
                           capture_output=True, timeout=300)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr[-500:], b.stderr[-500:])
        assert a.stdout == b.stdout and len(a.stdout) > 100, flags
    clean = subprocess.run([os.path.join(REFBIN, "feacat_refmain"), "-c", world["cfg"], "--raw-output", wav],
                           capture_output=True, timeout=300).stdout
    noisy = subprocess.run([os.path.join(REFBIN, "feacat_refmain"), "-c", world["cfg"], "--raw-output", "-G", "0.5", wav],
                           capture_output=True, timeout=300).stdout
    d = np.frombuffer(noisy, "<f4") - np.frombuffer(clean, "<f4")
    assert 0.4 < d.std() < 0.6 and abs(d.mean()) < 0.05
