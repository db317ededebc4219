"""Host-side logic of the product library that runs without a GPU."""
import pytest


def test_recipe_batch_range_matches_oracle(capi, oracle):
    """aasr_recipe_batch_range (C++ restatement of Recipe::read's batching,
    aku/Recipe.cc:63-112) against the oracle's Python restatement."""
    for L in range(0, 30):
        text = "\n".join("audio=a%d.wav lna=a%d.lna" % (i, i) for i in range(L)) + "\n"
        for n in range(0, 10):
            for b in range(1, max(n, 1) + 1):
                infos = oracle.recipe_read(text, n, b)
                first, cnt = capi.recipe_batch_range(L, n, b)
                assert cnt == len(infos), (L, n, b)
                if cnt:
                    assert infos[0].audio_path == "a%d.wav" % first


def test_recipe_batch_index_validation(capi):
    with pytest.raises(capi.AasrError, match="Invalid batch index"):
        capi.recipe_batch_range(10, 4, 5)
    with pytest.raises(capi.AasrError, match="Invalid batch index"):
        capi.recipe_batch_range(10, 4, 0)


def _random_text(rng, alphabet, n):
    return "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), n))


def test_str_split_and_clean_vs_reference(oracle):
    """The oracle's str::split / str::clean restatements against the reference's own str.cc
    (compiled in place into oracle/_ref/libaku_ref.so) on random strings, in the three ways the
    recipe and speaker-configuration readers call them (aku/Recipe.cc:57,79-83,
    aku/SpeakerConfig.cc:33-38,102)."""
    import ctypes as C
    import numpy as np
    A = oracle.ref_aku()
    if A is None:
        pytest.skip("oracle/_ref/libaku_ref.so not built (no reference tree)")
    rng = np.random.default_rng(23)
    buf = C.create_string_buffer(4096)
    cases = ["", "a=", "=b", "a=b=", "a==b", "=", "==", "a", " a  b ", "\ta \t b", "a b ", "k=v=w"]
    cases += [_random_text(rng, "ab= \t\rx", int(rng.integers(0, 12))) for _ in range(3000)]
    for s in cases:
        for delims, group, nf in ((" \t", 1, 0), ("=", 0, 0), (" \t", 1, 2)):
            n = A.ref_str_split(s.encode(), delims.encode(), group, nf, buf, 4096)
            want = buf.value.decode().split("\x1f") if n else []
            assert oracle.str_split(s, delims, bool(group), nf) == want, (s, delims, group, nf)
        for chars in (" \t", "\n\t "):
            A.ref_str_clean(s.encode(), chars.encode(), buf, 4096)
            assert oracle.str_clean(s, chars) == buf.value.decode(), (s, chars)


def test_engine_recipe_reader_matches_oracle(capi, oracle):
    """aasr_recipe_read (host) against the oracle's Recipe::read on recipes with the odd cases:
    key persistence, comments, tabs, runs of blanks, CR line ends (kept), trailing '=',
    empty keys, start/end times, every batch of several splits."""
    import numpy as np
    rng = np.random.default_rng(29)
    keys = ["audio", "lna", "speaker", "utterance", "start-time", "end-time", "transcript", ""]
    vals = ["a.wav", "x/y.lna", "spk1", "u7", "1.5", "2.25e1", "12abc", "", "v=", "=", "\r"]

    def rows(infos):
        return [(i.audio_path, i.lna_path, i.speaker_id, i.utterance_id, i.start_time, i.end_time) for i in infos]

    fixed = ("# comment\n\n  audio=a.wav\tlna=a.lna speaker=s1 start-time=0.5\n"
             "audio=b.wav   lna=b.lna=\n\t\naudio=c.wav lna=c.lna utterance=u\r end-time=3\r\n#x\nlna=d.lna")
    texts = [fixed, "", "\n\n", "audio=a lna=b"]
    for _ in range(300):
        lines = []
        for _ in range(int(rng.integers(0, 9))):
            fields = ["%s=%s" % (keys[int(rng.integers(0, len(keys)))], vals[int(rng.integers(0, len(vals)))])
                      for _ in range(int(rng.integers(0, 5)))]
            sep = [" ", "\t", "  ", " \t "][int(rng.integers(0, 4))]
            lines.append(["", " ", "#"][int(rng.integers(0, 3)) if rng.random() < 0.2 else 0] + sep.join(fields))
        texts.append("\n".join(lines) + ("\n" if rng.random() < 0.5 else ""))
    checked = 0
    for text in texts:
        for n, b in ((0, 0), (1, 1), (2, 1), (2, 2), (3, 2), (4, 4)):
            try:
                want = rows(oracle.recipe_read(text, n, b))
            except ValueError as e:
                with pytest.raises(capi.AasrError) as ei:
                    capi.recipe_read(text, n, b)
                assert str(e) in str(ei.value)
                continue
            assert capi.recipe_read(text, n, b) == want, (text, n, b)
            checked += 1
    assert checked > 500
    got = capi.recipe_read(fixed)
    assert got[1][:2] == ("b.wav", "b.lna") and got[1][2] == "s1" and got[2][4] == 0.5 and got[3][1] == "d.lna"
    assert got[2][3] == "u\r" and got[2][5] == 3.0          # the reference does not strip CR


def test_recipe_times_are_float_and_frame_limits_use_float_arithmetic(capi, oracle):
    """Recipe::Info::start_time / end_time are float fields (aku/Recipe.hh:48-49) and
    phone_probs multiplies them with the float frame_rate() (aku/phone_probs.cc:199-206): the
    product is rounded to float before the truncation.  2.008 s * 125 is frame 250 (251 in
    double), 8.008 s * 125 is 1001 (1000 in double)."""
    import numpy as np
    got = capi.recipe_read("audio=a lna=b start-time=2.008 end-time=8.008\n")
    assert got[0][4] == float(np.float32(2.008)) and got[0][5] == float(np.float32(8.008))
    assert capi.recipe_frame_limits(2.008, 8.008, 125.0) == (250, 1001)
    assert int(2.008 * 125.0) == 251 and int(8.008 * 125.0) == 1000     # what double would give
    assert capi.recipe_frame_limits(0.0, 0.0, 125.0) == (0, 2 ** 31 - 1)
    info = oracle.recipe_read("audio=a lna=b start-time=2.008 end-time=8.008\n")[0]
    assert oracle.recipe_frame_limits(info, 125.0) == (250, 1001)
    # a millisecond sweep: the engine's limits equal the float restatement everywhere, and the
    # double evaluation would differ on a few hundred of them
    differ = 0
    for ms in range(0, 200000, 7):
        t = "%.3f" % (ms / 1000.0)
        i = oracle.recipe_read("audio=a lna=b start-time=%s end-time=%s\n" % (t, t))[0]
        want = oracle.recipe_frame_limits(i, 125.0)
        assert capi.recipe_frame_limits(float(np.float32(float(t))), float(np.float32(float(t))), 125.0) == want
        differ += int(int(float(t) * 125.0) != want[0])
    assert differ > 20


def test_recipe_cluster_speakers_and_all_fields(capi, oracle):
    """aasr_recipe_read_all: every Recipe::Info field and the cluster_speakers rule (a batch only
    ends where the speaker changes, aku/Recipe.cc:86-101) against the oracle's Recipe::read."""
    import numpy as np
    rng = np.random.default_rng(31)
    for trial in range(120):
        lines = []
        spk = 0
        for i in range(int(rng.integers(1, 25))):
            if rng.random() < 0.35:
                spk += 1
            f = ["audio=a%d.wav" % i, "lna=l%d" % i]
            if rng.random() < 0.8:
                f.append("speaker=s%d" % spk)
            if rng.random() < 0.3:
                f += ["transcript=t%d.phn" % i, "alignment=al%d" % i, "hmmnet=h%d" % i, "den-hmmnet=d%d" % i,
                      "alt-audio=b%d.wav" % i, "start-line=%d" % i, "end-line=%d" % (i + 7), "utterance=u%d" % i]
            lines.append(" ".join(f))
        text = "\n".join(lines) + "\n"
        for n in (0, 2, 3, 5):
            for b in range(1, max(n, 1) + 1):
                for cl in (False, True):
                    want = [(i.audio_path, i.alt_audio_path, i.transcript_path, i.alignment_path, i.hmmnet_path,
                             i.den_hmmnet_path, i.lna_path, i.start_time, i.end_time, i.start_line, i.end_line,
                             i.speaker_id, i.utterance_id) for i in oracle.recipe_read(text, n, b, cl)]
                    assert capi.recipe_read_all(text, n, b, cl) == want, (text, n, b, cl)


@pytest.mark.parametrize("nbytes", [1, 2, 4])
def test_lna_file_reader_matches_the_reference_reader(capi, oracle, tmp_path, nbytes):
    """aasr_lna_read_file (1-, 2- and 4-byte LNA, decoder/src/LnaReaderCircular.cc:63-96,166-198)
    against the recogniser's own reader compiled in place (oracle/_ref/liblna_ref.so): every code
    value of the 1-byte form, random 2-byte codes incl. 0 and 0xFFFF, raw floats; a trailing
    partial frame is dropped by both."""
    import numpy as np
    rng = np.random.default_rng(5 + nbytes)
    S, F = 37, 41
    if nbytes == 1:
        body = (np.arange(F * S) % 256).astype(np.uint8)
    elif nbytes == 2:
        codes = rng.integers(0, 65536, F * S).astype(">u2")
        codes[:3] = (0, 65535, 1)
        body = codes.view(np.uint8)
    else:
        body = rng.uniform(-80, 0, F * S).astype("<f4").view(np.uint8)
    path = str(tmp_path / "x.lna")
    with open(path, "wb") as f:
        f.write(oracle.lna_header(S, nbytes))
        f.write(body.tobytes())
        f.write(b"\x07" * (S * nbytes - 1))     # an incomplete last frame
    lp, nb = capi.lna_read_file(path)
    assert nb == nbytes and lp.shape == (F, S) and lp.dtype == np.float32
    if nbytes == 1:
        assert np.array_equal(lp.ravel(), (body.astype(np.float64) / -24.0).astype(np.float32))
    if oracle.ref_lna() is not None:
        ref = oracle.ref_lna_read(path, F + 5, S, buf_size=8, order=0)
        assert ref.shape == (F, S) and np.array_equal(lp, ref)
    elif nbytes != 1:
        assert np.array_equal(lp, oracle.lna_decode(open(path, "rb").read()).astype(np.float32)[:F])
    with pytest.raises(capi.AasrError):
        capi.lna_read_file(str(tmp_path / "missing.lna"))
    bad = str(tmp_path / "bad.lna")
    open(bad, "wb").write(b"\x00\x00\x00\x05\x03" + b"\x00" * 30)
    with pytest.raises(capi.AasrError, match="invalid header"):
        capi.lna_read_file(bad)
