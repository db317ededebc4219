"""Host-side audio input (aasr_audio_read; replaces the libsndfile calls of aku/AudioReader.cc).
libsndfile is not in the image, so the containers are written with Python's own `wave`, `sunau`,
`aifc` and `audioop` modules -- independent implementations of the same published formats -- and
the expected samples follow sf_read_short()'s rule (the 16 most significant bits; G.711 tables)."""
import struct
import warnings

import numpy as np
import pytest

with warnings.catch_warnings():
    warnings.simplefilter("ignore", DeprecationWarning)
    import aifc
    import audioop
    import sunau
    import wave


def _pcm(n=4000, seed=5):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) * 6000).clip(-32768, 32767).astype(np.int16)
    x[:6] = [0, 1, -1, 32767, -32768, 255]
    return x


def _widen(x16, width, seed=9):
    """int16 -> `width`-byte little-endian samples with random low-order bits."""
    rng = np.random.default_rng(seed)
    if width == 1:
        return (x16 >> 8).astype(np.int8).tobytes()
    if width == 2:
        return x16.astype("<i2").tobytes()
    low = rng.integers(0, 1 << (8 * (width - 2)), len(x16), dtype=np.int64)
    v = (x16.astype(np.int64) << (8 * (width - 2))) | low
    raw = v.astype("<i8").tobytes()
    return b"".join(raw[8 * i:8 * i + width] for i in range(len(x16)))


@pytest.mark.parametrize("width", [1, 2, 3, 4])
def test_wav_pcm_widths(capi, tmp_path, width):
    x = _pcm()
    p = str(tmp_path / "a.wav")
    data = _widen(x, width)
    want = x if width > 1 else ((x >> 8) << 8).astype(np.int16)
    if width == 1:                                   # WAV 8-bit is unsigned
        data = bytes((b + 128) & 0xFF for b in data)
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(width); w.setframerate(16000)
        w.writeframes(data)
    got, rate = capi.audio_read(p)
    assert rate == 16000 and np.array_equal(got, want)


def test_wav_extra_chunks_extensible_and_odd_sizes(capi, tmp_path):
    x = _pcm(1001)
    fmt = struct.pack("<HHIIHH", 0xFFFE, 1, 8000, 16000, 2, 16) + struct.pack("<HHI", 22, 16, 4) + \
        struct.pack("<H", 1) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    body = b"WAVE" + b"LIST" + struct.pack("<I", 5) + b"abcde\x00" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + \
        b"fact" + struct.pack("<I", 4) + struct.pack("<I", len(x)) + b"data" + struct.pack("<I", 2 * len(x)) + \
        x.astype("<i2").tobytes() + b"cue " + struct.pack("<I", 4) + b"\0\0\0\0"
    p = str(tmp_path / "x.wav")
    open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    got, rate = capi.audio_read(p)
    assert rate == 8000 and np.array_equal(got, x)


@pytest.mark.parametrize("law", ["ulaw", "alaw"])
def test_g711_in_wav_au_aiffc(capi, tmp_path, law):
    x = _pcm()
    enc = audioop.lin2ulaw if law == "ulaw" else audioop.lin2alaw
    dec = audioop.ulaw2lin if law == "ulaw" else audioop.alaw2lin
    codes = enc(x.tobytes(), 2)
    want = np.frombuffer(dec(codes, 2), np.int16)
    every = bytes(range(256))                      # the complete decoding table
    want_all = np.frombuffer(dec(every, 2), np.int16)
    # WAV format 7 / 6
    fmt = struct.pack("<HHIIHH", 7 if law == "ulaw" else 6, 1, 8000, 8000, 1, 8)
    for payload, expect in ((codes, want), (every, want_all)):
        body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(payload)) + payload
        p = str(tmp_path / "g.wav")
        open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
        got, rate = capi.audio_read(p)
        assert rate == 8000 and np.array_equal(got, expect)
    # AU encoding 1 / 27
    p = str(tmp_path / "g.au")
    open(p, "wb").write(b".snd" + struct.pack(">IIIII", 28, len(codes), 1 if law == "ulaw" else 27, 8000, 1) +
                        b"\0\0\0\0" + codes)
    got, rate = capi.audio_read(p)
    assert rate == 8000 and np.array_equal(got, want)
    # AIFF-C written by the standard library (it compresses the 16-bit input itself)
    p = str(tmp_path / "g.aifc")
    with aifc.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
        w.setcomptype(law.upper().encode() if law == "ulaw" else b"ALAW", law.encode())
        w.writeframes(x.tobytes())
    got, rate = capi.audio_read(p)
    assert rate == 8000 and np.array_equal(got, want)


@pytest.mark.parametrize("width", [1, 2, 3, 4])
def test_au_and_aiff_linear(capi, tmp_path, width):
    x = _pcm(3001)
    le = _widen(x, width)
    be = b"".join(le[i:i + width][::-1] for i in range(0, len(le), width))
    want = x if width > 1 else ((x >> 8) << 8).astype(np.int16)
    p = str(tmp_path / "a.au")
    with sunau.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(width); w.setframerate(22050); w.setcomptype("NONE", "")
        w.writeframes(be)
    got, rate = capi.audio_read(p)
    assert rate == 22050 and np.array_equal(got, want)
    p = str(tmp_path / "a.aiff")
    with aifc.open(p, "wb") as w:
        w.aiff(); w.setnchannels(1); w.setsampwidth(width); w.setframerate(44100)
        w.writeframes(be)
    got, rate = capi.audio_read(p)
    assert rate == 44100 and np.array_equal(got, want)


@pytest.mark.parametrize("fmt", ["01", "10"])
def test_nist_sphere(capi, tmp_path, fmt):
    x = _pcm(2500)
    head = ("NIST_1A\n   1024\nsample_count -i %d\nsample_n_bytes -i 2\nchannel_count -i 1\n"
            "sample_byte_format -s2 %s\nsample_rate -i 16000\nsample_coding -s3 pcm\nend_head\n" % (len(x), fmt))
    p = str(tmp_path / "a.sph")
    open(p, "wb").write(head.encode().ljust(1024, b" ") + x.astype("<i2" if fmt == "01" else ">i2").tobytes() + b"zz")
    got, rate = capi.audio_read(p)
    assert rate == 16000 and np.array_equal(got, x)
    bad = head.replace("-s3 pcm", "-s26 pcm,embedded-shorten-v2.00")
    open(p, "wb").write(bad.encode().ljust(1024, b" ") + b"\0" * 64)
    with pytest.raises(capi.AasrError) as ei:
        capi.audio_read(p)
    assert ei.value.code == capi.AASR_ERR_UNSUPPORTED and "shorten" in str(ei.value)


def test_headerless_fallback_and_errors(capi, tmp_path):
    x = _pcm(777)
    p = str(tmp_path / "a.raw")
    open(p, "wb").write(x.astype("<i2").tobytes() + b"\x7f")     # trailing odd byte is dropped
    got, rate = capi.audio_read(p)
    assert rate == 0 and np.array_equal(got, x)
    open(p, "wb").write(b"")
    got, _ = capi.audio_read(p)
    assert len(got) == 0
    with pytest.raises(capi.AasrError) as ei:
        capi.audio_read(str(tmp_path / "missing.wav"))
    assert ei.value.code == capi.AASR_ERR_IO and "could not open file" in str(ei.value)
    # stereo is refused with the reference's message (aku/AudioReader.cc:148-150)
    p = str(tmp_path / "st.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(x[:100].tobytes())
    with pytest.raises(capi.AasrError) as ei:
        capi.audio_read(p)
    assert "multiple channels not supported" in str(ei.value)
    # float WAV: loud (the reference only warns and reads [-1, 1] as shorts)
    fmt = struct.pack("<HHIIHH", 3, 1, 16000, 64000, 4, 32)
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", 8) + b"\0" * 8
    open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    with pytest.raises(capi.AasrError) as ei:
        capi.audio_read(p)
    assert ei.value.code == capi.AASR_ERR_UNSUPPORTED
