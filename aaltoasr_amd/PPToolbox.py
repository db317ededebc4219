"""PPToolbox -- the reference's Python facade for the path, on the engine.

The reference builds a SWIG module `PPToolbox` from aku/swig/PPToolbox.i:57-75 over
aku::PPToolbox (aku/PhoneProbsToolbox.{hh,cc}); scripts do

    import PPToolbox
    t = PPToolbox.PPToolbox()
    t.read_configuration("x.cfg"); t.read_models("/models/base")
    t.generate("a.wav", "a.lna", False)

This module keeps that class -- same method names, argument order and meaning, errors raised as
RuntimeError (what PPToolbox.i:14-29 maps every C++ exception to) -- and runs it through the C ABI
(include/aasr.h): 2-byte LNA, normalised, as aku/PhoneProbsToolbox.cc:84-131 hard-wires.
`sys.path.insert(0, ".../aaltoasr_amd"); import PPToolbox` is the drop-in spelling.
"""
import os

try:
    from . import capi as _A
except ImportError:  # imported as a top-level module named PPToolbox
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from aaltoasr_amd import capi as _A


def _guard(fn):
    def wrapped(*args, **kw):
        try:
            return fn(*args, **kw)
        except _A.AasrError as e:
            raise RuntimeError("Exception: %s" % e) from None
    wrapped.__name__ = fn.__name__
    wrapped.__doc__ = fn.__doc__
    return wrapped


class PPToolbox:
    def __init__(self):
        self._feat = None
        self._gmm = None

    @_guard
    def read_configuration(self, cfgname):
        """aku/PhoneProbsToolbox.cc:42-47: FeatureGenerator::load_configuration of the file."""
        try:
            with open(cfgname, "r", encoding="latin-1") as f:
                text = f.read()
        except OSError:
            raise RuntimeError("Exception: could not open %s" % cfgname) from None
        self._feat = _A.Feat(text)

    @_guard
    def read_models(self, base):
        """HmmSet::read_all(base): base.ph, base.mc, base.gk (aku/HmmSet.cc:662-667)."""
        self._gmm = _A.Gmm.from_files(base + ".gk", base + ".mc", base + ".ph")

    @_guard
    def set_clustering(self, clfile_name, eval_minc, eval_ming):
        """aku/PhoneProbsToolbox.cc:50-53."""
        self._need(models=True)
        self._gmm.read_clustering(clfile_name)
        self._gmm.set_clustering_min_evals(float(eval_minc), float(eval_ming))

    def _need(self, models=False, both=False):
        if (models or both) and self._gmm is None:
            raise RuntimeError("Exception: no models read")
        if both and self._feat is None:
            raise RuntimeError("Exception: no feature configuration read")

    def _lna(self, pcm):
        self._need(both=True)
        if self._gmm.dim != self._feat.dim:
            raise RuntimeError("Exception: Gaussian dimension is %d but feature dimension is %d."
                               % (self._gmm.dim, self._feat.dim))
        image, _ = _A.run_utterance(self._feat, self._gmm, pcm, 0, 0, True, 2)
        return image

    @_guard
    def generate(self, input_name, output_name, raw_flag):
        """aku/PhoneProbsToolbox.cc:210-222 -> generate_from_file_to_fd (:135-208).  Like the
        reference, raw_flag is not looked at: headerless input is the audiofile module's `raw`
        option or the reader's fallback for files without a known header."""
        self._need(both=True)
        try:
            with open(input_name, "rb") as f:
                data = f.read()
        except OSError:
            raise RuntimeError("Exception: could not open file %s" % input_name) from None
        pcm, _ = _A.audio_decode(data, self._feat)
        image = self._lna(pcm)
        with open(output_name, "wb") as f:
            f.write(image)

    @_guard
    def generate_to_fd(self, in_fd, out_fd, raw_flag):
        """aku/PhoneProbsToolbox.cc:55-133: audio from descriptor `in_fd` (read to its end), LNA to
        descriptor `out_fd`; neither is closed."""
        self._need(both=True)
        chunks = []
        while True:
            c = os.read(in_fd, 1 << 20)
            if not c:
                break
            chunks.append(c)
        pcm, _ = _A.audio_decode(b"".join(chunks), self._feat)
        image = self._lna(pcm)
        view = memoryview(image)
        while len(view):
            n = os.write(out_fd, view)
            if n <= 0:
                raise RuntimeError("Exception: Write error")
            view = view[n:]
