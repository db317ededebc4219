"""User-defined feature module types: aku::FeatureModule as a plugin base class
(aku/FeatureModule.hh:47-154 -- private virtual set_module_config / generate, m_sources, m_buffer)
registered with FeatureGenerator::register_module_type<T>() and evaluated on the host through
aasr_feat_register_module_type, inside a graph whose other modules run on the device."""
import os
import subprocess

import numpy as np
import pytest

from aaltoasr_amd import synth

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aaltoasr_amd", "lib", "bin")


def _mod(name, typ, src=None, **kw):
    t = "module\n{\n  name %s\n  type %s\n" % (name, typ)
    for k, v in kw.items():
        t += "  %s %s\n" % (k, v)
    return t + ("  sources %s\n" % src if src else "") + "}\n"


def _graph(delta_type, width=2, tail=True):
    cfg = _mod("a", "audiofile", sample_rate=16000) + _mod("f", "fft", "a", magnitude=0)
    cfg += _mod("m", "mel", "f") + _mod("p", "power", "f") + _mod("c", "dct", "m", dim=12) + _mod("cp", "merge", "c p")
    cfg += _mod("d1", delta_type, "cp", width=width) + _mod("d2", "delta", "d1", width=3)
    cfg += _mod("all", "merge", "cp d1 d2")
    if tail:
        cfg += _mod("cms", "mean_subtractor", "all", left=20, right=10)
    return cfg


def _run(tmp_path, cfg, wav, first, n):
    path = str(tmp_path / ("g%d.cfg" % abs(hash(cfg))))
    open(path, "w").write(cfg)
    r = subprocess.run([os.path.join(BIN, "plugin_check"), path, wav, str(first), str(n)], capture_output=True,
                       text=True, timeout=300)
    return r, (np.array([[float(x) for x in l.split()] for l in r.stdout.splitlines()]) if r.returncode == 0 else None)


@pytest.fixture(scope="module")
def wav(tmp_path_factory):
    import wave
    p = str(tmp_path_factory.mktemp("plugin") / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(synth.make_audio(24000, seed=91).astype("<i2").tobytes())
    return p


@pytest.mark.parametrize("width", [1, 2, 4])
def test_user_delta_module_equals_the_builtin_one(capi, tmp_path, wav, width):
    """my_delta restates DeltaModule in user code (generate(frame) over m_sources.back()->at(frame +- k)
    into m_buffer[frame]); the graph around it -- spectral front end before, second delta, merge and
    mean subtractor after -- runs on the device.  Same numbers as the built-in delta, bit for bit,
    including frames before 0 and past the end of the audio."""
    ra, a = _run(tmp_path, _graph("delta", width), wav, -8, 210)
    rb, b = _run(tmp_path, _graph("my_delta", width), wav, -8, 210)
    assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr, rb.stderr)
    assert a.shape == (210, 39) and np.array_equal(a, b)
    # write_configuration keeps the user module's own options
    assert "type my_delta" in rb.stderr and "width %d" % width in rb.stderr


def test_user_module_with_two_sources_and_options(capi, oracle, tmp_path, wav):
    cfg = _mod("a", "audiofile", sample_rate=16000) + _mod("f", "fft", "a", magnitude=0)
    cfg += _mod("m", "mel", "f") + _mod("p", "power", "f")
    cfg += _mod("g", "my_gain", "m p", gain=0.5)
    r, got = _run(tmp_path, cfg, wav, 0, 40)
    assert r.returncode == 0, r.stderr
    ft = capi.Feat(_mod("a", "audiofile", sample_rate=16000) + _mod("f", "fft", "a", magnitude=0) +
                   _mod("m", "mel", "f") + _mod("p", "power", "f"))
    pcm, _ = oracle.read_wav_pcm16(wav)
    mel = ft.run(pcm, 0, 40, module="m", dtype=np.float64)
    power = ft.run(pcm, 0, 40, module="p", dtype=np.float64)
    assert np.array_equal(got, np.float64(np.float32(0.5)) * mel + power[:, :1])


def test_user_module_errors_are_reported(capi, tmp_path, wav):
    # a configuration the module itself rejects
    r, _ = _run(tmp_path, _graph("my_delta", 0), wav, 0, 5)
    assert r.returncode != 0 and "Delta width must be greater than zero" in r.stderr
    # an unregistered type is still unknown (through the C ABI, no registration in this process)
    with pytest.raises(capi.AasrError, match="Unknown module type 'my_delta'"):
        capi.Feat(_graph("my_delta"))
    # a built-in name cannot be taken
    import ctypes as C
    L = capi.lib()
    L.aasr_feat_register_module_type.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
    buf = (C.c_void_p * 3)()
    assert L.aasr_feat_register_module_type(b"delta", C.cast(buf, C.c_void_p), None) != 0
    assert b"built in" in L.aasr_last_error()
