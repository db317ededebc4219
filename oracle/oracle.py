"""oracle.py -- Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module, and only as the checker.  The product path
(aaltoasr_amd/) never imports it.

The numeric kernels live in oracle/aasr_oracle.c (each function cites the
reference lines it restates); this file holds the text-format restatements
(.cfg feature graphs, .gk/.mc/.ph model files, recipes, LNA files) and the
graph walk that strings the per-module C functions together.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

TINY_FOR_LOG = 1e-50  # aku/util.hh:131


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "aasr_oracle.c"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


def _declare(L: C.CDLL) -> None:
    d, f, i32, i64 = C.c_double, C.c_float, C.c_int, C.c_int64
    pd, pf = C.POINTER(d), C.POINTER(f)
    pi32, pi16, pu8 = C.POINTER(C.c_int32), C.POINTER(C.c_int16), C.POINTER(C.c_uint8)
    L.orc_safe_log.restype = d
    L.orc_safe_log.argtypes = [d]
    L.orc_diag_setup.argtypes = [i32, i64, pd, pd, pd]
    L.orc_diag_loglik.restype = d
    L.orc_diag_loglik.argtypes = [i32, pd, pd, pd, d]
    L.orc_pool_likelihoods.argtypes = [i32, i64, pd, pd, pd, pd, pd, pd]
    L.orc_mixture_normalize.argtypes = [i64, pi32, pd]
    L.orc_state_likelihoods.argtypes = [i64, pi32, pi32, pd, pd, pd]
    L.orc_score_frames.argtypes = [i32, i64, pd, pd, pd, i64, pi32, pi32, pd, i64, pd, pd, pd, pd]
    L.orc_lna_frame.argtypes = [pd, i64, i32, i32, pf, pu8]
    L.orc_cluster_centres.argtypes = [i32, i32, pi32, pi32, pd, pd, pd, pd, pd]
    L.orc_score_frames_clustered.argtypes = [i32, i64, pd, pd, pd, i64, pi32, pi32, pd, i32, pi32, pi32,
                                             pd, pd, pd, i32, i32, i64, pd, pd, pd, pi32]
    L.orc_window_advance.restype = f
    L.orc_window_advance.argtypes = [i32, f]
    L.orc_default_window_width.restype = i32
    L.orc_default_window_width.argtypes = [i32, f]
    L.orc_last_frame.restype = i32
    L.orc_last_frame.argtypes = [i64, i32, f]
    L.orc_eof_frame.restype = i32
    L.orc_eof_frame.argtypes = [i64, i32, f]
    L.orc_audio_frames.restype = i32
    L.orc_audio_frames.argtypes = [pi16, i64, f, i32, f, i32, i32, i32, pd]
    L.orc_rfft.restype = i32
    L.orc_rfft.argtypes = [i32, pf, pf, pf]
    L.orc_hamming.argtypes = [i32, pf]
    L.orc_fft_module.restype = i32
    L.orc_fft_module.argtypes = [pd, i32, i32, i32, i32, pd]
    L.orc_mel_dim.restype = i32
    L.orc_mel_dim.argtypes = [i32]
    L.orc_mel_edges.argtypes = [i32, i32, i32, pf]
    L.orc_mel_module.argtypes = [pd, i32, i32, i32, i32, pd]
    L.orc_power_module.argtypes = [pd, i32, i32, pd]
    L.orc_dct_module.argtypes = [pd, i32, i32, i32, i32, pd]
    L.orc_delta_default_norm.restype = f
    L.orc_delta_default_norm.argtypes = [i32]
    L.orc_delta_module.argtypes = [pd, i32, i32, i32, f, pd]
    L.orc_var_to_scale.argtypes = [i32, pf]
    L.orc_normalization_module.argtypes = [pd, i32, i32, pf, pf, pd]
    L.orc_lin_transform_module.argtypes = [pd, i32, i32, i32, pf, pf, pd]
    L.orc_mean_subtract_module.argtypes = [pd, i32, i32, i32, i32, pd]
    L.orc_vtln_bins.argtypes = [i32, i32, f, i32, f, pf, i32, pf]
    L.orc_vtln_sinc_table.argtypes = [i32, pf, i32, i32, pi32, pi32, pf]
    L.orc_vtln_allpass_blin.argtypes = [i32, f, pf]
    L.orc_vtln_module.argtypes = [pd, i32, i32, i32, pf, pi32, pi32, pf, i32, pd]
    L.orc_srnorm_table.argtypes = [i32, i32, i32, f, pi32, pi32, pf]
    L.orc_srnorm_module.argtypes = [pd, i32, i32, i32, i32, pi32, pi32, pf, i32, pd]
    L.orc_mel_power_module.argtypes = [pd, i32, i32, pd]
    L.orc_quanteq_module.argtypes = [pd, i32, i32, pf, pf, pf, pd]
    L.orc_cpu_baseline_score.restype = d
    L.orc_cpu_baseline_score.argtypes = [i32, i64, pd, pd, pd, i64, pi32, pi32, pd, i64, pd]


# ---------------------------------------------------------------------------
# text helpers
# ---------------------------------------------------------------------------

def str2float(s: str) -> np.float32:
    """str::str2float (aku/str.cc:260-282): strtod, then narrowed to float."""
    return np.float32(float(s))


def _floats(s: str) -> np.ndarray:
    return np.array([str2float(x) for x in s.split()], dtype=np.float32)


# ---------------------------------------------------------------------------
# .cfg feature graph  (ModuleConfig::read aku/ModuleConfig.cc:166-202,
# FeatureGenerator::load_configuration aku/FeatureGenerator.cc:96-219)
# ---------------------------------------------------------------------------

def parse_feature_config(text: str) -> List[Dict[str, str]]:
    """Returns the list of module option dicts in file order."""
    lines = text.split("\n")
    mods: List[Dict[str, str]] = []
    i = 0
    while i < len(lines):
        line = lines[i].strip(" \t\r")
        i += 1
        if not line:
            continue
        if line != "module":
            raise ValueError("expected keyword 'module' on line %d: %s" % (i, line))
        opts: Dict[str, str] = {}
        first = True
        while True:
            if i >= len(lines):
                raise ValueError("unexpected end of module config file")
            l2 = lines[i].strip(" \t\r")
            i += 1
            if not l2:
                continue
            if first:
                if l2 != "{":
                    raise ValueError("'{' expected in module config file: " + l2)
                first = False
                continue
            if l2 == "}":
                break
            parts = l2.split(None, 1)
            if len(parts) == 1:
                raise ValueError("value missing for option: " + l2)
            if parts[0] in opts:
                raise ValueError("value redefined: " + l2)
            opts[parts[0]] = parts[1].strip(" \t")
        if "type" not in opts:
            raise ValueError("type not defined for module")
        if "name" not in opts:
            raise ValueError("name not defined for module")
        mods.append(opts)
    if not mods:
        raise ValueError("no feature modules defined")
    return mods


@dataclass
class _Mod:
    name: str
    type: str
    opts: Dict[str, str]
    sources: List["_Mod"] = field(default_factory=list)
    dim: int = 0
    prm: dict = field(default_factory=dict)


class FeatureChain:
    """Restatement of aku::FeatureGenerator for the MFCC-chain module types.

    Module semantics: aku/FeatureModules.cc (audiofile :327-440, fft :475-566,
    mel :775-849, power :874-885, mel_power :904-923, dct :937-979, delta
    :998-1037, normalization :1056-1142, lin_transform :1167-1269, merge
    :1335-1364, mean_subtractor :1384-1454, concat :1472-1501, vtln :1529-1937,
    sr_norm :1953-2058, quanteq :2078-2141).  set_parameters() is the
    per-speaker hook SpeakerConfig drives (aku/SpeakerConfig.cc:365-378).
    Output of a module at frame t is a pure function of t, so
    the ring buffers of FeatureModule::at (:102-158) are replaced by range
    evaluation with halos.
    """

    SUPPORTED = ("audiofile", "pre", "fft", "mel", "power", "dct", "delta",
                 "normalization", "lin_transform", "merge", "mean_subtractor",
                 "concat", "vtln", "sr_norm", "mel_power", "quanteq")

    def __init__(self, cfg_text: str):
        L = lib()
        self.mods: List[_Mod] = []
        self.by_name: Dict[str, _Mod] = {}
        for opts in parse_feature_config(cfg_text):
            m = _Mod(opts["name"], opts["type"], opts)
            if m.type not in self.SUPPORTED:
                raise ValueError("Unknown module type '%s'" % m.type)
            if not self.mods:
                if m.type not in ("audiofile", "pre"):
                    raise ValueError("first module should be a base module")
                if "sources" in opts:
                    raise ValueError("can not define sources for the first module")
            else:
                if "sources" not in opts:
                    raise ValueError("sources not defined for module: " + m.name)
                for s in opts["sources"].split():
                    if s not in self.by_name:
                        raise ValueError("unknown source module: " + s)
                    if m.sources and m.type != "merge":
                        raise ValueError("Multiple sources are not allowed for module " + m.type)
                    m.sources.append(self.by_name[s])
            if m.name in self.by_name:
                raise ValueError("multiple definitions of module name: " + m.name)
            self._configure(m, L)
            self.mods.append(m)
            self.by_name[m.name] = m
        self.base = self.mods[0]
        self.last = self.mods[-1]

    # -- per-module set_module_config ---------------------------------------
    def _configure(self, m: _Mod, L) -> None:
        o = m.opts
        t = m.type
        if t == "audiofile":
            if "sample_rate" not in o:
                raise ValueError("AudioFileModule: Must set sample rate")
            sr = int(o["sample_rate"])
            emph = str2float(o["pre_emph_coef"]) if "pre_emph_coef" in o else np.float32(0.97)
            fr = str2float(o["frame_rate"]) if "frame_rate" in o else np.float32(125)
            adv = np.float32(L.orc_window_advance(sr, float(fr)))
            ww = L.orc_default_window_width(sr, float(fr))
            if "window_width" in o:
                ww = int(o["window_width"])
            cb = int(o["copy_borders"]) if "copy_borders" in o else 1
            m.dim = ww
            m.prm = dict(sample_rate=sr, emph=emph, frame_rate=fr, advance=adv,
                         width=ww, copy_borders=cb)
            self.sample_rate = sr
        elif t == "pre":
            # PreModule::set_module_config (aku/FeatureModules.cc:672-690)
            if "dim" not in o:
                raise ValueError("PreModule: Must set dimension")
            m.dim = int(o["dim"])
            m.prm = dict(sample_rate=int(o.get("sample_rate", 16000)),
                         frame_rate=str2float(o["frame_rate"]) if "frame_rate" in o else np.float32(125),
                         legacy=int(o.get("legacy_file", 0)))
            self.sample_rate = m.prm["sample_rate"]
        elif t == "fft":
            m.prm = dict(magnitude=int(o.get("magnitude", 1)), log=int(o.get("log", 0)))
            m.dim = m.sources[-1].dim // 2 + 1
        elif t == "mel":
            m.prm = dict(root=int(o.get("root", 0)))
            m.dim = L.orc_mel_dim(self.sample_rate)
        elif t == "power":
            m.dim = 1
        elif t == "dct":
            m.dim = int(o.get("dim", 12))
            if m.dim < 1:
                raise ValueError("DCTModule: Dimension must be > 0")
            m.prm = dict(zeroth=int(o.get("zeroth", 0)))
        elif t == "delta":
            m.dim = m.sources[-1].dim
            w = int(o.get("width", 2))
            norm = np.float32(L.orc_delta_default_norm(w))
            if "normalization" in o:
                norm = str2float(o["normalization"])
            if w < 1:
                raise ValueError("DeltaModule: Delta width must be > 0")
            m.prm = dict(width=w, norm=norm)
        elif t == "normalization":
            m.dim = m.sources[-1].dim
            mean = np.zeros(m.dim, np.float32)
            scale = np.ones(m.dim, np.float32)
            if "mean" in o:
                mean = _floats(o["mean"])
            if len(mean) != m.dim:
                raise ValueError("NormalizationModule: Invalid mean dimension")
            if "var" in o and "scale" in o:
                raise ValueError("NormalizationModule: Both scale and var can not be defined simultaneously")
            if "var" in o:
                scale = _floats(o["var"])
                if len(scale) != m.dim:
                    raise ValueError("Normalization module: Invalid variance dimension")
                L.orc_var_to_scale(m.dim, _p(scale, C.c_float))
            elif "scale" in o:
                scale = _floats(o["scale"])
            m.prm = dict(mean=mean, scale=scale)
        elif t == "lin_transform":
            src_dim = m.sources[-1].dim
            m.dim = int(o["dim"]) if "dim" in o else src_dim
            if m.dim < 1:
                raise ValueError("LinTransformModule: Dimension must be > 0")
            mat = _floats(o["matrix"]) if "matrix" in o else None
            bias = _floats(o["bias"]) if "bias" in o else None
            if mat is not None and len(mat) == 0:
                mat = None
            if bias is not None and len(bias) == 0:
                bias = None
            if mat is not None and len(mat) != m.dim * src_dim:
                raise ValueError("LinTransformModule: Invalid matrix dimension")
            if mat is None and m.dim != src_dim:
                # identity fallback copies dim values (reference would read
                # past the source for dim > src_dim); keep it strict here
                raise ValueError("LinTransformModule: identity needs dim == src_dim")
            if bias is not None and len(bias) != m.dim:
                raise ValueError("LinTransformModule: Invalid bias dimension")
            m.prm = dict(matrix=mat, bias=bias, src_dim=src_dim)
        elif t == "merge":
            m.dim = sum(s.dim for s in m.sources)
        elif t == "mean_subtractor":
            m.dim = m.sources[-1].dim
            left = int(o.get("left", 75))
            right = int(o.get("right", 75))
            if left + 1 < 1 or right + 1 < 1:
                raise ValueError("MeanSubtractorModule: context widths must be >= 0")
            m.prm = dict(left=left, right=right)
        elif t == "concat":
            left, right = int(o.get("left", 0)), int(o.get("right", 0))
            if left < 0 or right < 0:
                raise ValueError("ConcatModule: context spans must be >= 0")
            m.dim = m.sources[-1].dim * (1 + left + right)
            m.prm = dict(left=left, right=right)
        elif t == "mel_power":
            m.dim = 1
        elif t == "vtln":
            m.dim = m.sources[0].dim
            p = dict(pwlin=int(o.get("pwlin_vtln", 0)),
                     turn=str2float(o["pwlin_turnpoint"]) if "pwlin_turnpoint" in o else np.float32(0.8),
                     slapt=int(o.get("slapt", 0)), rad=int(o.get("sinc_interpolation_rad", 8)),
                     all_pass=int(o.get("all-pass", 0)))
            if p["pwlin"] and p["slapt"]:
                raise ValueError("VtlnModule: Can not use both pwlin_vtln and slapt!")
            if p["pwlin"] and p["all_pass"]:
                raise ValueError("VtlnModule: Can not use both pwlin_vtln and all-pass!")
            p["lanczos"] = int(o.get("lanczos_window", 0 if p["all_pass"] else 1)) > 0
            if p["lanczos"] and p["all_pass"]:
                raise ValueError("VtlnModule: Can not use both lanczos_window and all-pass!")
            m.prm = p
            self._vtln_tables(m, warp=np.float32(1.0), slapt=np.zeros(1, np.float32))
        elif t == "sr_norm":
            p = dict(in_frames=int(o.get("in_frames", 0)), out_frames=int(o.get("out_frames", 0)),
                     order=int(o.get("lanczos_order", 4)))
            if p["in_frames"] == 0 or p["out_frames"] == 0:
                raise ValueError("SRNormModule: Must set both in_frames and out_frames.")
            if m.sources[0].dim % p["in_frames"]:
                raise ValueError("SRNormModule: in_frames does not match with the input dimension")
            if p["order"] < 1:
                raise ValueError("SRNormModule: lanczos_order must be positive.")
            p["frame_dim"] = m.sources[0].dim // p["in_frames"]
            m.dim = p["out_frames"] * p["frame_dim"]
            m.prm = p
            self._srnorm_table(m, str2float(o["speech_rate"]) if "speech_rate" in o else np.float32(1.0))
        elif t == "quanteq":
            m.dim = m.sources[-1].dim
            m.prm = dict(alpha=None, gamma=None, qmax=None)

    def _vtln_tables(self, m: _Mod, warp, slapt) -> None:
        L, p, dim = lib(), m.prm, m.dim
        pf, pi = C.c_float, C.c_int32
        slapt = np.ascontiguousarray(slapt, np.float32)
        p["warp"], p["slapt_params"] = np.float32(warp), slapt
        bins = np.zeros(dim, np.float32)
        L.orc_vtln_bins(dim, p["pwlin"], float(p["turn"]), p["slapt"], float(warp), _p(slapt, pf),
                        len(slapt), _p(bins, pf))
        p["bins"] = bins
        if p["all_pass"]:
            coef = np.zeros((dim, dim), np.float32)
            if p["slapt"]:
                L.orc_vtln_allpass_slapt.restype = None
                L.orc_vtln_allpass_slapt.argtypes = [C.c_int32, C.POINTER(pf), C.c_int32, C.POINTER(pf)]
                L.orc_vtln_allpass_slapt(dim, _p(slapt, pf), len(slapt), _p(coef, pf))
            else:
                L.orc_vtln_allpass_blin(dim, float(warp), _p(coef, pf))
            p["start"], p["len"], p["coef"] = np.zeros(dim, np.int32), np.full(dim, dim, np.int32), coef
        elif p["rad"] > 0:
            w = 2 * p["rad"] + 1
            p["start"], p["len"] = np.zeros(dim, np.int32), np.zeros(dim, np.int32)
            p["coef"] = np.zeros((dim, w), np.float32)
            L.orc_vtln_sinc_table(dim, _p(bins, pf), p["rad"], int(p["lanczos"]), _p(p["start"], pi),
                                  _p(p["len"], pi), _p(p["coef"], pf))
        else:
            p["start"], p["len"], p["coef"] = np.zeros(dim, np.int32), np.zeros(dim, np.int32), np.zeros((dim, 1), np.float32)

    def _srnorm_table(self, m: _Mod, sr) -> None:
        p = m.prm
        w = 2 * p["order"] + 1
        p["speech_rate"] = np.float32(sr)
        p["start"], p["len"] = np.zeros(p["out_frames"], np.int32), np.zeros(p["out_frames"], np.int32)
        p["coef"] = np.zeros((p["out_frames"], w), np.float32)
        lib().orc_srnorm_table(p["in_frames"], p["out_frames"], p["order"], float(sr), _p(p["start"], C.c_int32),
                               _p(p["len"], C.c_int32), _p(p["coef"], C.c_float))

    def set_parameters(self, module: str, opts: Dict[str, str]) -> None:
        """FeatureModule::set_parameters of `module` with one parsed { key value }
        block (normalization :1089-1112, lin_transform :1187-1196, vtln :1575-1592,
        sr_norm :1990-1996, quanteq :2085-2094; a no-op for every other type)."""
        if module not in self.by_name:
            raise ValueError("unknown module requested: " + module)
        m = self.by_name[module]
        o = opts
        if m.type == "normalization":
            mean = _floats(o["mean"]) if "mean" in o else m.prm["mean"]
            if len(mean) != m.dim:
                raise ValueError("NormalizationModule: Invalid mean dimension")
            scale = m.prm["scale"]
            if "var" in o and "scale" in o:
                raise ValueError("NormalizationModule: Both scale and var can not be defined simultaneously")
            if "var" in o:
                scale = _floats(o["var"])
                if len(scale) != m.dim:
                    raise ValueError("Normalization module: Invalid variance dimension")
                lib().orc_var_to_scale(m.dim, _p(scale, C.c_float))
            elif "scale" in o:
                scale = _floats(o["scale"])
                if len(scale) != m.dim:
                    raise ValueError("NormalizationModule: Invalid scale dimension")
            m.prm = dict(mean=mean, scale=scale)
        elif m.type == "lin_transform":
            mat = _floats(o["matrix"]) if "matrix" in o else None
            bias = _floats(o["bias"]) if "bias" in o else None
            if mat is not None and len(mat) == 0:
                mat = None
            if bias is not None and len(bias) == 0:
                bias = None
            sd = m.prm["src_dim"]
            if mat is not None and len(mat) != m.dim * sd:
                raise ValueError("LinTransformModule: Invalid matrix dimension")
            if bias is not None and len(bias) != m.dim:
                raise ValueError("LinTransformModule: Invalid bias dimension")
            m.prm = dict(matrix=mat, bias=bias, src_dim=sd)
        elif m.type == "vtln":
            if m.prm["slapt"]:
                self._vtln_tables(m, m.prm["warp"], _floats(o["slapt_coef"]) if "slapt_coef" in o
                                  else np.zeros(1, np.float32))
            else:
                self._vtln_tables(m, str2float(o["warp_factor"]) if "warp_factor" in o else np.float32(1.0),
                                  m.prm["slapt_params"])
        elif m.type == "sr_norm":
            self._srnorm_table(m, str2float(o["speech_rate"]) if "speech_rate" in o else np.float32(1.0))
        elif m.type == "quanteq":
            m.prm = dict(alpha=_floats(o["alpha"]) if "alpha" in o else None,
                         gamma=_floats(o["gamma"]) if "gamma" in o else None,
                         qmax=_floats(o["quant_max"]) if "quant_max" in o else None)

    def get_parameters(self, module: str, opts: Dict[str, str]) -> None:
        """FeatureModule::get_parameters: writes the module's current parameters
        into opts, every float printed with "%g" (ModuleConfig::set,
        aku/ModuleConfig.cc:21-26, 49-60)."""
        m = self.by_name[module]
        g = lambda v: " ".join("%g" % float(x) for x in v)
        if m.type == "normalization":
            opts["mean"], opts["scale"] = g(m.prm["mean"]), g(m.prm["scale"])
        elif m.type == "lin_transform":
            mat, bias, sd = m.prm["matrix"], m.prm["bias"], m.prm["src_dim"]
            opts["matrix"] = g(mat if mat is not None else np.eye(m.dim, sd, dtype=np.float32).ravel())
            opts["bias"] = g(bias if bias is not None else np.zeros(m.dim, np.float32))
        elif m.type == "vtln":
            if m.prm["slapt"]:
                opts["slapt_coef"] = g(m.prm["slapt_params"])
            else:
                opts["warp_factor"] = "%g" % float(m.prm["warp"])
        elif m.type == "sr_norm":
            opts["speech_rate"] = "%g" % float(m.prm["speech_rate"])
        elif m.type == "quanteq":
            for key, name in (("alpha", "alpha"), ("gamma", "gamma"), ("qmax", "quant_max")):
                opts[name] = g(m.prm[key]) if m.prm[key] is not None else ""

    # -- FeatureGenerator surface ---------------------------------------------
    @property
    def dim(self) -> int:
        return self.last.dim

    @property
    def frame_rate(self) -> float:
        return float(self.base.prm["frame_rate"])

    def last_frame(self, n_samples: int) -> int:
        if self.base.type == "pre":
            # PreModule::last_frame (:649-660); n_samples = number of float values
            return int(n_samples) // self.base.dim - 1
        p = self.base.prm
        return lib().orc_last_frame(int(n_samples), p["width"], float(p["advance"]))

    def num_frames(self, n_samples: int) -> int:
        """frames phone_probs emits for a whole file: it stops at the first frame whose window
        crosses the end (aku/phone_probs.cc:217-221 with AudioFileModule::eof; orc_eof_frame) --
        last_frame() + 1 except where that float formula is off (files beyond 2^24 samples,
        fractional window advances)."""
        if self.base.type == "pre":
            return self.last_frame(n_samples) + 1
        p = self.base.prm
        return lib().orc_eof_frame(int(n_samples), p["width"], float(p["advance"]))

    def halo(self) -> Tuple[int, int]:
        """(left, right) base-module frames needed around one output frame."""
        def rec(m: _Mod) -> Tuple[int, int]:
            l = r = 0
            if m.type == "delta":
                l = r = m.prm["width"]
            elif m.type in ("mean_subtractor", "concat"):
                l, r = m.prm["left"], m.prm["right"]
            bl = br = 0
            for s in m.sources:
                sl, sr = rec(s)
                bl, br = max(bl, sl), max(br, sr)
            return l + bl, r + br
        return rec(self.last)

    def generate(self, pcm: np.ndarray, first_frame: int, n_frames: int,
                 module: Optional[str] = None) -> np.ndarray:
        """Frames first_frame .. first_frame+n_frames-1 of `module` (default:
        the last module) as float64 [n_frames x dim]."""
        if self.base.type == "pre":
            pcm = np.ascontiguousarray(pcm, dtype=np.float32).reshape(-1, self.base.dim)
        else:
            pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        target = self.by_name[module] if module else self.last
        memo: Dict[Tuple[str, int, int], np.ndarray] = {}
        return self._eval(target, first_frame, first_frame + n_frames - 1, pcm, memo)

    def _eval(self, m: _Mod, lo: int, hi: int, pcm: np.ndarray, memo) -> np.ndarray:
        key = (m.name, lo, hi)
        if key in memo:
            return memo[key]
        L = lib()
        n = hi - lo + 1
        out = np.empty((n, m.dim), dtype=np.float64)
        pd = C.c_double
        t = m.type
        if t == "pre":
            # PreModule::generate (aku/FeatureModules.cc:705-755): frames before 0
            # give frame 0, frames from the end of the file on give the last frame
            if len(pcm) < 1:
                raise ValueError("PreModule: Could not read the file")
            idx = np.clip(np.arange(lo, hi + 1), 0, len(pcm) - 1)
            out = pcm[idx].astype(np.float64)
        elif t == "audiofile":
            p = m.prm
            rc = L.orc_audio_frames(_p(pcm, C.c_int16), len(pcm), float(p["advance"]),
                                    p["width"], float(p["emph"]), p["copy_borders"],
                                    lo, n, _p(out, pd))
            if rc != 0:
                raise ValueError("audio shorter than frame")
        elif t == "merge":
            out = np.ascontiguousarray(
                np.hstack([self._eval(s, lo, hi, pcm, memo) for s in m.sources]))
        elif t == "delta":
            w = m.prm["width"]
            src = self._eval(m.sources[-1], lo - w, hi + w, pcm, memo)
            L.orc_delta_module(_p(src, pd), n, m.dim, w, float(m.prm["norm"]), _p(out, pd))
        elif t == "mean_subtractor":
            l, r = m.prm["left"], m.prm["right"]
            src = self._eval(m.sources[-1], lo - l - 1, hi + r, pcm, memo)
            L.orc_mean_subtract_module(_p(src, pd), n, m.dim, l, r, _p(out, pd))
        elif t == "concat":
            # ConcatModule::generate (aku/FeatureModules.cc:1488-1501)
            l, r = m.prm["left"], m.prm["right"]
            src = self._eval(m.sources[-1], lo - l, hi + r, pcm, memo)
            out = np.ascontiguousarray(np.hstack([src[i:i + n] for i in range(l + r + 1)]))
        else:
            src = self._eval(m.sources[-1], lo, hi, pcm, memo)
            sd = m.sources[-1].dim
            if t == "fft":
                rc = L.orc_fft_module(_p(src, pd), n, sd, m.prm["magnitude"], m.prm["log"], _p(out, pd))
                if rc != 0:
                    raise ValueError("FFT window width %d needs a radix the oracle does not restate" % sd)
            elif t == "mel":
                L.orc_mel_module(_p(src, pd), n, sd, self.sample_rate, m.prm["root"], _p(out, pd))
            elif t == "power":
                L.orc_power_module(_p(src, pd), n, sd, _p(out, pd))
            elif t == "dct":
                L.orc_dct_module(_p(src, pd), n, sd, m.dim, m.prm["zeroth"], _p(out, pd))
            elif t == "normalization":
                L.orc_normalization_module(_p(src, pd), n, m.dim, _p(m.prm["mean"], C.c_float),
                                           _p(m.prm["scale"], C.c_float), _p(out, pd))
            elif t == "lin_transform":
                mat, bias = m.prm["matrix"], m.prm["bias"]
                L.orc_lin_transform_module(
                    _p(src, pd), n, sd, m.dim,
                    _p(mat, C.c_float) if mat is not None else None,
                    _p(bias, C.c_float) if bias is not None else None, _p(out, pd))
            elif t == "mel_power":
                L.orc_mel_power_module(_p(src, pd), n, sd, _p(out, pd))
            elif t == "vtln":
                p = m.prm
                rad = m.dim if p["all_pass"] else p["rad"]   # all-pass: full-row weights
                L.orc_vtln_module(_p(src, pd), n, m.dim, rad, _p(p["bins"], C.c_float),
                                  _p(p["start"], C.c_int32), _p(p["len"], C.c_int32),
                                  _p(p["coef"], C.c_float), p["coef"].shape[1], _p(out, pd))
            elif t == "sr_norm":
                p = m.prm
                L.orc_srnorm_module(_p(src, pd), n, p["in_frames"], p["out_frames"], p["frame_dim"],
                                    _p(p["start"], C.c_int32), _p(p["len"], C.c_int32),
                                    _p(p["coef"], C.c_float), p["coef"].shape[1], _p(out, pd))
            elif t == "quanteq":
                p = m.prm
                full = p["alpha"] is not None and p["gamma"] is not None and p["qmax"] is not None \
                    and len(p["alpha"]) and len(p["gamma"]) and len(p["qmax"])
                L.orc_quanteq_module(_p(src, pd), n, m.dim,
                                     _p(p["alpha"], C.c_float) if full else None,
                                     _p(p["gamma"], C.c_float) if full else None,
                                     _p(p["qmax"], C.c_float) if full else None, _p(out, pd))
            else:
                raise AssertionError(t)
        memo[key] = out
        return out


# ---------------------------------------------------------------------------
# acoustic model  (HmmSet / PDFPool / Mixture)
# ---------------------------------------------------------------------------

@dataclass
class DiagModel:
    """Diagonal-Gaussian HmmSet: pool means/variances + per-state mixtures.

    mix_off[S+1], mix_idx, mix_w form the CSR of Mixture pointers/weights
    (aku/Distributions.hh Mixture::m_pointers/m_weights); legacy .ph models map
    state i to emission pdf i (aku/HmmSet.cc:245,319-322).
    """
    mean: np.ndarray      # [G, D] float64
    var: np.ndarray       # [G, D] float64
    mix_off: np.ndarray   # [S+1] int32
    mix_idx: np.ndarray   # [K] int32
    mix_w: np.ndarray     # [K] float64 (normalised like Mixture::read)
    prec: np.ndarray = None
    cst: np.ndarray = None

    def __post_init__(self):
        self.mean = np.ascontiguousarray(self.mean, np.float64)
        self.var = np.ascontiguousarray(self.var, np.float64)
        self.mix_off = np.ascontiguousarray(self.mix_off, np.int32)
        self.mix_idx = np.ascontiguousarray(self.mix_idx, np.int32)
        self.mix_w = np.ascontiguousarray(self.mix_w, np.float64).copy()
        L = lib()
        L.orc_mixture_normalize(self.S, _p(self.mix_off, C.c_int32), _p(self.mix_w, C.c_double))
        self.prec = np.empty_like(self.mean)
        self.cst = np.empty(self.G, np.float64)
        L.orc_diag_setup(self.D, self.G, _p(self.var, C.c_double), _p(self.prec, C.c_double),
                         _p(self.cst, C.c_double))

    @property
    def G(self) -> int:
        return self.mean.shape[0]

    @property
    def D(self) -> int:
        return self.mean.shape[1]

    @property
    def S(self) -> int:
        return len(self.mix_off) - 1

    def gauss_loglik(self, frames: np.ndarray) -> np.ndarray:
        """[F x G] log-likelihood of every pool Gaussian."""
        frames = np.ascontiguousarray(frames, np.float64)
        L = lib()
        F = frames.shape[0]
        out = np.empty((F, self.G))
        lik = np.empty(self.G)
        pd = C.c_double
        for f in range(F):
            L.orc_pool_likelihoods(self.D, self.G, _p(self.mean, pd), _p(self.prec, pd),
                                   _p(self.cst, pd), _p(frames[f], pd), _p(lik, pd), _p(out[f], pd))
        return out

    def score(self, frames: np.ndarray, want_lik: bool = False):
        """[F x S] log state likelihood, log(max(sum_k w_k exp(ll_k), 1e-50))."""
        frames = np.ascontiguousarray(frames, np.float64)
        L = lib()
        F = frames.shape[0]
        out = np.empty((F, self.S))
        lik = np.empty((F, self.S)) if want_lik else None
        scratch = np.empty(self.G)
        pd = C.c_double
        L.orc_score_frames(self.D, self.G, _p(self.mean, pd), _p(self.prec, pd), _p(self.cst, pd),
                           self.S, _p(self.mix_off, C.c_int32), _p(self.mix_idx, C.c_int32),
                           _p(self.mix_w, pd), F, _p(frames, pd), _p(scratch, pd), _p(out, pd),
                           _p(lik, pd) if want_lik else None)
        return (out, lik) if want_lik else out

    # -- Gaussian clustering (PDFPool::read_clustering / HmmSet::set_clustering_min_evals)
    def set_clustering(self, n_clusters: int, pairs, eval_minc: float = 0.0, eval_ming: float = 0.1):
        """pairs = [(gauss_index, cluster_index), ...] exactly as
        PDFPool::read_clustering pushes them (aku/Distributions.cc:3136-3148), i.e.
        read_gcl()'s output with the final pair repeated.  Thresholds follow
        HmmSet::set_clustering_min_evals (aku/HmmSet.cc:1359-1366):
        int(ratio * clusters), int(ratio * pool size)."""
        members = [[] for _ in range(n_clusters)]
        for g, c in pairs:
            members[c].append(g)
        self.cl_off = np.zeros(n_clusters + 1, np.int32)
        self.cl_off[1:] = np.cumsum([len(m) for m in members])
        self.cl_members = np.array([g for m in members for g in m], np.int32)
        self.n_clusters = n_clusters
        self.c_mean = np.zeros((n_clusters, self.D))
        self.c_prec = np.zeros((n_clusters, self.D))
        self.c_cst = np.zeros(n_clusters)
        pd = C.c_double
        lib().orc_cluster_centres(self.D, n_clusters, _p(self.cl_off, C.c_int32),
                                  _p(self.cl_members, C.c_int32), _p(self.mean, pd), _p(self.var, pd),
                                  _p(self.c_mean, pd), _p(self.c_prec, pd), _p(self.c_cst, pd))
        self.min_clusters = int(eval_minc * n_clusters)
        self.min_gaussians = int(eval_ming * self.G)

    def score_clustered(self, frames: np.ndarray, want_counts: bool = False):
        """[F x S] log state likelihood with the clustered pool evaluation of
        PDFPool::precompute_likelihoods (aku/Distributions.cc:2684-2722)."""
        frames = np.ascontiguousarray(frames, np.float64)
        F = frames.shape[0]
        out = np.empty((F, self.S))
        scratch = np.empty(self.G)
        counts = np.zeros(F, np.int32)
        pd, pi = C.c_double, C.c_int32
        lib().orc_score_frames_clustered(
            self.D, self.G, _p(self.mean, pd), _p(self.prec, pd), _p(self.cst, pd), self.S,
            _p(self.mix_off, pi), _p(self.mix_idx, pi), _p(self.mix_w, pd), self.n_clusters,
            _p(self.cl_off, pi), _p(self.cl_members, pi), _p(self.c_mean, pd), _p(self.c_prec, pd),
            _p(self.c_cst, pd), self.min_clusters, self.min_gaussians, F, _p(frames, pd),
            _p(scratch, pd), _p(out, pd), _p(counts, pi))
        return (out, counts) if want_counts else out

    def score_clustered_adapted(self, frames: np.ndarray, W: np.ndarray, want_counts: bool = False):
        """score_clustered with one global constrained-MLLR transform W = [b | A] on the pool
        (AdaptedGaussian members, plain cluster centres)."""
        frames = np.ascontiguousarray(frames, np.float64)
        W = np.ascontiguousarray(W, np.float64)
        F = frames.shape[0]
        out = np.empty((F, self.S))
        scratch = np.empty(self.G)
        counts = np.zeros(F, np.int32)
        pd, pi = C.c_double, C.c_int32
        L = lib()
        L.orc_score_frames_clustered_adapted.restype = None
        L.orc_score_frames_clustered_adapted.argtypes = [
            C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
            C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_score_frames_clustered_adapted(
            self.D, self.G, self.mean.ctypes.data, self.prec.ctypes.data, self.cst.ctypes.data, self.S,
            self.mix_off.ctypes.data, self.mix_idx.ctypes.data, self.mix_w.ctypes.data, self.n_clusters,
            self.cl_off.ctypes.data, self.cl_members.ctypes.data, self.c_mean.ctypes.data, self.c_prec.ctypes.data,
            self.c_cst.ctypes.data, self.min_clusters, self.min_gaussians, W.ctypes.data, F, frames.ctypes.data,
            scratch.ctypes.data, out.ctypes.data, counts.ctypes.data)
        return (out, counts) if want_counts else out

    def score_clustered_classes(self, frames: np.ndarray, g2t, W, want_counts: bool = False):
        """score_clustered with per-class constrained-MLLR transforms: g2t[g] = transform of Gaussian
        g (-1: none), W = [T x D x (D+1)] matrices [b | A] (AdaptedGaussian members of each class,
        plain cluster centres)."""
        frames = np.ascontiguousarray(frames, np.float64)
        W = np.ascontiguousarray(W, np.float64)
        g2t = np.ascontiguousarray(g2t, np.int32)
        F = frames.shape[0]
        out = np.empty((F, self.S))
        scratch = np.empty(self.G)
        counts = np.zeros(F, np.int32)
        L = lib()
        L.orc_score_frames_clustered_classes.restype = None
        L.orc_score_frames_clustered_classes.argtypes = [
            C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_score_frames_clustered_classes(
            self.D, self.G, self.mean.ctypes.data, self.prec.ctypes.data, self.cst.ctypes.data, self.S,
            self.mix_off.ctypes.data, self.mix_idx.ctypes.data, self.mix_w.ctypes.data, self.n_clusters,
            self.cl_off.ctypes.data, self.cl_members.ctypes.data, self.c_mean.ctypes.data, self.c_prec.ctypes.data,
            self.c_cst.ctypes.data, self.min_clusters, self.min_gaussians, W.shape[0], g2t.ctypes.data,
            W.ctypes.data, F, frames.ctypes.data, scratch.ctypes.data, out.ctypes.data, counts.ctypes.data)
        return (out, counts) if want_counts else out

    def cpu_baseline(self, frames: np.ndarray) -> float:
        frames = np.ascontiguousarray(frames, np.float64)
        pd = C.c_double
        return lib().orc_cpu_baseline_score(
            self.D, self.G, _p(self.mean, pd), _p(self.prec, pd), _p(self.cst, pd), self.S,
            _p(self.mix_off, C.c_int32), _p(self.mix_idx, C.c_int32), _p(self.mix_w, pd),
            frames.shape[0], _p(frames, pd))


def lna_encode(state_lik: np.ndarray, normalize: bool = True, lnabytes: int = 2):
    """phone_probs frame loop tail (aku/phone_probs.cc:224-262) on linear state
    likelihoods [F x S] (float64).  Returns (float32 log-probs, uint8 bytes)."""
    state_lik = np.ascontiguousarray(state_lik, np.float64)
    F, S = state_lik.shape
    lp = np.empty((F, S), np.float32)
    by = np.empty((F, S * lnabytes), np.uint8)
    L = lib()
    for f in range(F):
        L.orc_lna_frame(_p(state_lik[f], C.c_double), S, int(normalize), lnabytes,
                        _p(lp[f], C.c_float), _p(by[f], C.c_uint8))
    return lp, by


def lna_header(num_states: int, lnabytes: int) -> bytes:
    """write_int + fputc (aku/phone_probs.cc:32-43, 213-214)."""
    return struct.pack(">I", num_states) + bytes([lnabytes])


def lna_decode(data: bytes) -> np.ndarray:
    """LnaReaderCircular::go_to conventions (decoder/src/LnaReaderCircular.cc
    :129-209): 2-byte -> -(hi*256+lo)/1820, 4-byte -> LE float."""
    S, B = struct.unpack(">IB", data[:5])
    body = np.frombuffer(data[5:], np.uint8)
    if B == 2:
        v = body.reshape(-1, S, 2).astype(np.float64)
        return -(v[..., 0] * 256 + v[..., 1]) / 1820.0
    if B == 4:
        return body.view("<f4").reshape(-1, S).astype(np.float64)
    raise ValueError("bytes per value %d" % B)


# ---------------------------------------------------------------------------
# model files
# ---------------------------------------------------------------------------

def write_gk(path: str, mean: np.ndarray, var: np.ndarray, legacy: bool = False) -> None:
    """PDFPool::write_gk format (read side: aku/Distributions.cc:2811-2910):
    header 'G dim variable' then per Gaussian 'diag mean.. var..'; legacy
    header 'G dim diagonal_cov' has no per-line tag."""
    G, D = mean.shape
    with open(path, "w") as f:
        f.write("%d %d %s\n" % (G, D, "diagonal_cov" if legacy else "variable"))
        for g in range(G):
            vals = " ".join(repr(float(x)) for x in mean[g]) + " " + \
                   " ".join(repr(float(x)) for x in var[g])
            f.write(vals + "\n" if legacy else "diag " + vals + "\n")


def read_gk(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """PDFPool::read_gk (aku/Distributions.cc:2811-2910), diagonal types only.
    Token-based like the reference's istream >>."""
    toks = open(path).read().split()
    G, D, kind = int(toks[0]), int(toks[1]), toks[2]
    pos = 3
    mean = np.empty((G, D))
    var = np.empty((G, D))
    for g in range(G):
        if kind == "variable":
            tag = toks[pos]
            pos += 1
            if tag != "diag":
                raise ValueError("oracle reads diagonal Gaussians only, got " + tag)
        elif kind != "diagonal_cov":
            raise ValueError("oracle reads diagonal Gaussians only, got " + kind)
        mean[g] = [float(x) for x in toks[pos:pos + D]]
        pos += D
        var[g] = [float(x) for x in toks[pos:pos + D]]
        pos += D
    return mean, var


def write_mc(path: str, mix_off, mix_idx, mix_w) -> None:
    """HmmSet::write_mc / Mixture::write (read side aku/HmmSet.cc:156-180,
    aku/Distributions.cc:2418-2434): 'S' then per mixture 'n idx w idx w ...'."""
    S = len(mix_off) - 1
    with open(path, "w") as f:
        f.write("%d\n" % S)
        for s in range(S):
            a, b = mix_off[s], mix_off[s + 1]
            f.write("%d" % (b - a))
            for k in range(a, b):
                f.write(" %d %s" % (mix_idx[k], repr(float(mix_w[k]))))
            f.write("\n")


def read_mc(path: str):
    toks = open(path).read().split()
    S = int(toks[0])
    pos = 1
    off = [0]
    idx: List[int] = []
    w: List[float] = []
    for _ in range(S):
        n = int(toks[pos])
        pos += 1
        for _k in range(n):
            idx.append(int(toks[pos]))
            w.append(float(toks[pos + 1]))
            pos += 2
        off.append(len(idx))
    return np.array(off, np.int32), np.array(idx, np.int32), np.array(w, np.float64)


def write_ph(path: str, num_states: int, states_per_hmm: int = 1) -> None:
    """Legacy Noway PHONE file (read side HmmSet::read_legacy_ph,
    aku/HmmSet.cc:194-329): header 'PHONE', hmm count, then per hmm
    'index nstates+2 label', state line ('-1 -2 s0 s1..': dummy initial and
    final state ids are negative), and one transition line per state
    'from ntrans target prob ...'.  State i's emission pdf is i."""
    assert num_states % states_per_hmm == 0
    nh = num_states // states_per_hmm
    with open(path, "w") as f:
        f.write("PHONE\n%d\n" % nh)
        for h in range(nh):
            ns = states_per_hmm
            f.write("%d %d h%d\n" % (h + 1, ns + 2, h))
            ids = [-1, -2] + [h * ns + j for j in range(ns)]
            f.write(" ".join(str(x) for x in ids) + "\n")
            # source dummy state: one transition to the first real state
            f.write("0 1 2 1.0\n")
            f.write("1 0\n")
            for j in range(ns):
                nxt = 2 + j + 1 if j + 1 < ns else 1
                f.write("%d 2 %d 0.5 %d 0.5\n" % (2 + j, 2 + j, nxt))


def write_feature_file(path: str, frames: np.ndarray, legacy: bool = False) -> None:
    """feacat --raw-output -H (aku/feacat.cc:19-24, 95-100): int32 dimension (one
    byte for PreModule's legacy_file), then float32 frames."""
    frames = np.ascontiguousarray(frames, np.float32)
    with open(path, "wb") as f:
        f.write(struct.pack("b", frames.shape[1]) if legacy else struct.pack("=i", frames.shape[1]))
        f.write(frames.tobytes())


def write_gcl(path: str, n_clusters: int, gauss_to_cluster) -> None:
    """.gcl as gcluster writes it: cluster count, then 'gauss cluster' pairs
    (aku/Distributions.cc:3121-3147)."""
    with open(path, "w") as f:
        f.write("%d\n" % n_clusters)
        for g, c in enumerate(gauss_to_cluster):
            if c >= 0:
                f.write("%d %d\n" % (g, c))


def read_gcl(path: str, pool_size: int):
    """PDFPool::read_clustering's reader (aku/Distributions.cc:3121-3148).
    Returns (n_clusters, pairs).  The reference loop is
        while (in) { int g, c; in >> g >> c; ...checks...; push }
    so the iteration that hits end-of-file still runs its body with the values
    left from the previous pair (libstdc++ leaves the operands untouched when
    the stream sentry fails): the LAST pair of every file is pushed twice.
    That changes the centre of its cluster (the Gaussian is merged with weight
    2) and the member count used by the min-Gaussians test; kept."""
    toks = open(path).read().split()
    if not toks:
        raise ValueError("empty clustering file")
    n = int(toks[0])
    if n > 0.3 * pool_size:
        raise ValueError("PDFPool::read_clustering(): Number of clusters (%d) seems insensible "
                         "compared to the number of Gaussians (%d)." % (n, pool_size))
    pairs = []
    i = 1
    while i + 1 < len(toks):
        try:
            g, c = int(toks[i]), int(toks[i + 1])
        except ValueError:
            break
        if g >= pool_size:
            raise ValueError("PDFPool::read_clustering(): Gauss index out of bounds")
        if c >= n:
            raise ValueError("PDFPool::read_clustering(): Cluster index out of bounds")
        pairs.append((g, c))
        i += 2
    if pairs:
        pairs.append(pairs[-1])
    return n, pairs


def read_model(base: str) -> DiagModel:
    mean, var = read_gk(base + ".gk")
    off, idx, w = read_mc(base + ".mc")
    return DiagModel(mean, var, off, idx, w)


# ---------------------------------------------------------------------------
# speaker configuration  (SpeakerConfig, aku/SpeakerConfig.cc)
# ---------------------------------------------------------------------------

class SpeakerConfig:
    """aku::SpeakerConfig driving a FeatureChain and, for "model cmllr" blocks,
    a DiagModel: read_speaker_file (aku/SpeakerConfig.cc:20-153), set_speaker
    (:239-286), set_utterance (:288-318), retrieve_* (:322-362), set_modules
    (:365-378); ConstrainedMllr::set_parameters / get_parameters / load_transform
    (aku/ModelModules.cc:62-97, 129-160, 164-232) with the Gaussian sets of
    RegClassTree::Unit*::get_gaussians (aku/RegClassTree.cc:301-479).

    After set_speaker, `g2t` / `W` describe the model transform in force:
    g2t[g] = index into W (or -1), W[t] = [b | A] (float-rounded entries)."""

    UNITS = ("UNIT_PHONE", "UNIT_MIX", "UNIT_GAUSSIAN", "UNIT_NO")

    def __init__(self, chain: "FeatureChain", model: Optional[DiagModel] = None,
                 hmms: Optional[List[Tuple[str, List[int]]]] = None):
        self.chain, self.model, self.hmms = chain, model, hmms
        self.speakers: Dict[str, Dict[str, Dict[str, str]]] = {}
        self.utterances: Dict[str, Dict[str, Dict[str, str]]] = {}
        self.default_speaker: Optional[Dict[str, Dict[str, str]]] = None
        self.default_utterance: Optional[Dict[str, Dict[str, str]]] = None
        self.cur_speaker = self.cur_utterance = ""
        self.has_cmllr = False
        self.trans_is_reset, self.cmllr_loaded = True, False
        self.unit_mode = "UNIT_NO"
        self.trans: Dict[Tuple[str, ...], np.ndarray] = {}
        self.g2t = None if model is None else np.full(model.G, -1, np.int32)
        self.W = np.zeros((0, 0, 0))

    # -- reading ---------------------------------------------------------------
    def read_text(self, text: str) -> None:
        lines = text.split("\n")
        i = 0

        def nonempty():
            nonlocal i
            while i < len(lines):
                l = str_clean(lines[i], " \t")
                i += 1
                if l:
                    return l
            return None
        while True:
            l = nonempty()
            if l is None:
                return
            f = str_split(l, " \t", True)
            if len(f) != 2 or f[0] not in ("speaker", "utterance"):
                raise ValueError("SpeakerConfig: Syntax error on line %d: %s" % (i, l))
            is_spk, is_def = f[0] == "speaker", f[1] == "default"
            if is_def:
                if (self.default_speaker if is_spk else self.default_utterance) is not None:
                    raise ValueError("SpeakerConfig: Default %s configuration already defined" % f[0])
                target: Dict[str, Dict[str, str]] = {}
                if is_spk:
                    self.default_speaker = target
                else:
                    self.default_utterance = target
            else:
                target = (self.speakers if is_spk else self.utterances).setdefault(f[1], {})
            l = nonempty()
            if l != "{":
                raise ValueError("'{' expected in speaker config file: %s" % l)
            while True:
                l = nonempty()
                if l is None or l == "}":
                    break
                parts = str_split(l, " \t", True, 2)
                if len(parts) < 2:
                    l = "feature " + l
                    parts = str_split(l, " \t", True, 2)
                elif parts[0] not in ("model", "feature"):
                    raise ValueError("SpeakerConfig: Unknown module namespace at line %d" % i)
                if parts[0] == "feature" and parts[1] not in self.chain.by_name:
                    raise ValueError("SpeakerConfig: error on line %d: unknown module requested: %s" % (i, parts[1]))
                if parts[0] == "model":
                    if parts[1] != "cmllr":
                        raise ValueError("SpeakerConfig: error on line %d: unknown model module requested: %s"
                                         % (i, parts[1]))
                    self.has_cmllr = True
                # ModuleConfig::read
                opts: Dict[str, str] = {}
                if nonempty() != "{":
                    raise ValueError("SpeakerConfig: Failed reading module parameters: '{' expected")
                while True:
                    v = nonempty()
                    if v is None:
                        raise ValueError("SpeakerConfig: Failed reading module parameters: unexpected end")
                    if v == "}":
                        break
                    kv = v.split(None, 1)
                    if len(kv) < 2:
                        raise ValueError("value missing for option: " + v)
                    if kv[0] in opts:
                        raise ValueError("value redefined: " + v)
                    opts[kv[0]] = kv[1].strip(" \t")
                target.setdefault(l, opts)          # std::map::insert: first one stays

    # -- ConstrainedMllr -----------------------------------------------------------
    def _cmllr_set(self, opts: Dict[str, str]) -> None:
        self.trans = {}
        if opts.get("unitmode") in self.UNITS:
            self.unit_mode = opts["unitmode"]
        d = self.model.D
        n = d * (d + 1)
        k = 1
        while ("w%d" % k) in opts:
            parts = opts["w%d" % k].split()
            if len(parts) < (n if self.unit_mode == "UNIT_NO" else n + 1):
                raise ValueError("ERROR: not enough elements for matrix w%d" % k)
            unit = tuple(parts[:len(parts) - n])
            self.trans[unit] = np.array([str2float(x) for x in parts[len(parts) - n:]],
                                        np.float64).reshape(d, d + 1)
            k += 1
        if self.unit_mode == "UNIT_NO" and len(self.trans) > 1:
            raise ValueError("ERROR: speaker can only contain one transform when UNIT_NO (global transform) is set")

    def _cmllr_get(self, opts: Dict[str, str]) -> None:
        for k, unit in enumerate(sorted(self.trans), 1):
            opts["w%d" % k] = " ".join(list(unit) + ["%g" % v for v in self.trans[unit].ravel()])
        opts["unitmode"] = self.unit_mode

    def _unit_gaussians(self, unit: Tuple[str, ...]) -> List[int]:
        m = self.model
        comps = lambda s: [int(g) for g in m.mix_idx[m.mix_off[s]:m.mix_off[s + 1]]]
        if self.unit_mode == "UNIT_NO":
            return list(range(m.G))
        if self.unit_mode == "UNIT_GAUSSIAN":
            return [int(e) for e in unit]
        if self.unit_mode == "UNIT_MIX":
            return [g for e in unit if e.lstrip("-").isdigit() for g in comps(int(e))]
        out: List[int] = []
        for label, states in self.hmms or []:
            p1, p2 = label.rfind("-"), label.find("+")
            if p1 >= 0 and p2 >= 0:
                c = label[p1 + 1:p2] if p2 > p1 + 1 else ""
            elif p1 >= 0:
                c = label[p1 + 1:]
            elif p2 >= 0:
                c = label[:p2]
            else:
                c = label
            if c in unit:
                for s_ in states:
                    out += comps(s_)
        return out

    def _load_transforms(self) -> None:
        if self.has_cmllr and not self.cmllr_loaded:
            self.g2t = np.full(self.model.G, -1, np.int32)
            units = sorted(self.trans)               # std::map order over vector<string>
            for t, unit in enumerate(units):
                for g in self._unit_gaussians(unit):
                    self.g2t[g] = t                  # later transforms override
            d = self.model.D
            self.W = np.array([self.trans[u] for u in units]).reshape(len(units), d, d + 1)
            self.cmllr_loaded = True
        self.trans_is_reset = False

    # -- SpeakerConfig ---------------------------------------------------------------
    def _set_modules(self, modules: Dict[str, Dict[str, str]]) -> None:
        for key in sorted(modules):                  # std::map order over the key line
            ns, name = key.split(None, 1)
            if ns == "feature":
                self.chain.set_parameters(name, modules[key])
            else:
                self._cmllr_set(modules[key])

    def set_utterance(self, utterance_id: str = "") -> None:
        if self.cur_utterance:
            self._set_modules(self.utterances[self.cur_utterance])   # "retrieve" sets (reference quirk)
        if not utterance_id:
            if self.default_utterance is None:
                raise ValueError("SpeakerConfig: Default utterance is required.")
            self._set_modules(self.default_utterance)
        else:
            if utterance_id not in self.utterances:
                if self.default_utterance is None:
                    raise ValueError("SpeakerConfig: Unknown utterance %s, and default utterance settings are missing."
                                     % utterance_id)
                self.utterances[utterance_id] = {k: dict(v) for k, v in self.default_utterance.items()}
            self._set_modules(self.utterances[utterance_id])
        self.cur_utterance = utterance_id

    def set_speaker(self, speaker_id: str = "") -> None:
        if self.cur_speaker:
            for key, opts in self.speakers[self.cur_speaker].items():
                ns, name = key.split(None, 1)
                if ns == "feature":
                    self.chain.get_parameters(name, opts)
                else:
                    self._cmllr_get(opts)
        if self.cur_utterance:
            self.set_utterance("")
        if speaker_id != self.cur_speaker and not self.trans_is_reset and self.has_cmllr:
            self.cmllr_loaded, self.trans_is_reset = False, True
        load_new = self.trans_is_reset
        if not speaker_id:
            if self.default_speaker is None:
                raise ValueError("SpeakerConfig: No speaker defined, needs a default speaker.")
            self._set_modules(self.default_speaker)
        else:
            if speaker_id not in self.speakers:
                if self.default_speaker is None:
                    raise ValueError("SpeakerConfig: Unknown speaker %s, and default speaker settings are missing."
                                     % speaker_id)
                self.speakers[speaker_id] = {k: dict(v) for k, v in self.default_speaker.items()}
            self._set_modules(self.speakers[speaker_id])
        self.cur_speaker = speaker_id
        if load_new:
            self._load_transforms()


def score_adapted(model: DiagModel, frames: np.ndarray, g2t, W) -> np.ndarray:
    """AdaptedGaussian scoring (aku/ModelModules.hh:172-173, 208-212): Gaussian g
    evaluates A f + b of its transform and is scaled by |prod diag A|
    (full_matrix_determinant, aku/LinearAlgebra.cc:73-86); then the usual mixture
    sum and floor."""
    x = np.asarray(frames, np.float64)
    ll = model.gauss_loglik(x)
    for t in range(len(W)):
        A, b = W[t][:, 1:], W[t][:, 0]
        with np.errstate(divide="ignore"):
            adapted = model.gauss_loglik(x @ A.T + b) + np.log(abs(np.prod(np.diag(A))))
        sel = np.flatnonzero(np.asarray(g2t) == t)
        ll[:, sel] = adapted[:, sel]
    lik = np.exp(ll)
    out = np.empty((x.shape[0], model.S))
    for s_ in range(model.S):
        a, b_ = model.mix_off[s_], model.mix_off[s_ + 1]
        out[:, s_] = np.log(np.maximum(lik[:, model.mix_idx[a:b_]] @ model.mix_w[a:b_], TINY_FOR_LOG))
    return out


# ---------------------------------------------------------------------------
# recipes  (Recipe::read, aku/Recipe.cc:23-149)
# ---------------------------------------------------------------------------

@dataclass
class RecipeInfo:
    audio_path: str = ""
    alt_audio_path: str = ""
    transcript_path: str = ""
    alignment_path: str = ""
    hmmnet_path: str = ""
    den_hmmnet_path: str = ""
    lna_path: str = ""
    start_time: float = 0.0
    end_time: float = 0.0
    start_line: int = 0
    end_line: int = 0
    speaker_id: str = ""
    utterance_id: str = ""


_RECIPE_KEYS = {
    "audio": "audio_path", "alt-audio": "alt_audio_path", "transcript": "transcript_path",
    "alignment": "alignment_path", "hmmnet": "hmmnet_path", "den-hmmnet": "den_hmmnet_path",
    "lna": "lna_path", "speaker": "speaker_id", "utterance": "utterance_id",
}


def str_clean(s: str, chars: str) -> str:
    """str::clean (aku/str.cc:124-140)."""
    return s.strip(chars)


def str_split(s: str, delims: str, group: bool, num_fields: int = 0) -> List[str]:
    """str::split (aku/str.cc:142-172): one delimiter (a run of them with `group`) ends a
    field; the loop ends with the text, so a trailing delimiter opens no empty last field."""
    fields: List[str] = []
    begin = 0
    n = len(s)
    while begin < n:
        if num_fields > 0 and len(fields) == num_fields - 1:
            fields.append(s[begin:])
            break
        end = begin
        while end < n and s[end] not in delims:
            end += 1
        fields.append(s[begin:end])
        end += 1
        if group:
            while end < n and s[end] in delims:
                end += 1
        begin = end
    return fields


def recipe_read(text: str, num_batches: int = 0, batch_index: int = 0,
                cluster_speakers: bool = False) -> List[RecipeInfo]:
    """Restates Recipe::read including its quirks: the key=value map is NOT
    cleared between lines (keys persist, aku/Recipe.cc:31,82-90), batches are
    contiguous with the first (L mod n) batches one line longer (:63-112)."""
    if num_batches > 1 and (batch_index < 1 or batch_index > num_batches):
        raise ValueError("Invalid batch index")
    line_buffer = []
    for raw in text.split("\n"):
        line = str_clean(raw, "\n\t ")          # '\r' stays, as in the reference
        if not line or line[0] == "#":
            continue
        line_buffer.append(line)
    batch_remainder = 0
    if num_batches <= 1:
        target_lines = len(line_buffer)
    else:
        target_lines = len(line_buffer) // num_batches
        batch_remainder = len(line_buffer) % num_batches
    extra_line = 1
    if target_lines < 1:
        target_lines = 1
        extra_line = 0
    if batch_remainder == 0:
        extra_line = 0
    infos: List[RecipeInfo] = []
    kv: Dict[str, str] = {}
    cur_index = 1
    cur_line = 0
    cur_speaker = ""
    for line in line_buffer:
        for fld in str_split(line, " \t", True):
            parts = str_split(fld, "=", False)
            if len(parts) != 2:
                raise ValueError("Invalid recipe line: " + line)
            kv[parts[0]] = parts[1]
        if num_batches > 1 and cur_index < num_batches:
            new_speaker = kv.get("speaker", "")
            if cur_line >= target_lines + extra_line and (
                    not cluster_speakers or len(cur_speaker) == 0 or cur_speaker != new_speaker):
                cur_index += 1
                if cur_index > batch_index:
                    break
                cur_line -= target_lines + extra_line
                if cur_index > batch_remainder:
                    extra_line = 0
            cur_speaker = new_speaker
        if num_batches <= 1 or cur_index == batch_index:
            info = RecipeInfo()
            for k, attr in _RECIPE_KEYS.items():
                if k in kv:
                    setattr(info, attr, kv[k])
            if "start-time" in kv:
                info.start_time = float(np.float32(_atof(kv["start-time"])))  # float field, aku/Recipe.hh:48
            if "end-time" in kv:
                info.end_time = float(np.float32(_atof(kv["end-time"])))
            if "start-line" in kv:
                info.start_line = _atoi(kv["start-line"])
            if "end-line" in kv:
                info.end_line = _atoi(kv["end-line"])
            infos.append(info)
        cur_line += 1
    return infos


def recipe_frame_limits(info: "RecipeInfo", frame_rate) -> Tuple[int, int]:
    """aku/phone_probs.cc:199-206: `(int)(start_time * gen.frame_rate())` with both operands
    float (aku/Recipe.hh:48-49, aku/FeatureGenerator.hh:84), i.e. a float product truncated
    toward zero; an end frame of 0 means INT_MAX."""
    fr = np.float32(frame_rate)
    start = int(np.float32(info.start_time) * fr)
    end = int(np.float32(info.end_time) * fr)
    return start, (end if end != 0 else 2 ** 31 - 1)


def _atof(s: str) -> float:
    import re
    m = re.match(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)", s)
    return float(m.group(0)) if m else 0.0


def _atoi(s: str) -> int:
    import re
    m = re.match(r"\s*[-+]?\d+", s)
    return int(m.group(0)) if m else 0


# ---------------------------------------------------------------------------
# WAV helper (PCM16 mono) -- stands in for libsndfile's sf_read_short on the
# files the tests use (aku/AudioReader.cc:92-110,196-197)
# ---------------------------------------------------------------------------

def read_wav_pcm16(path: str) -> Tuple[np.ndarray, int]:
    import wave
    with wave.open(path, "rb") as w:
        if w.getnchannels() != 1:
            raise ValueError("AudioReader: sorry, audio files with multiple channels not supported")
        if w.getsampwidth() != 2:
            raise ValueError("oracle WAV reader handles PCM16 only")
        data = w.readframes(w.getnframes())
        return np.frombuffer(data, "<i2").copy(), w.getframerate()


# ---------------------------------------------------------------------------
# real-reference hooks (oracle/_ref, present only where it was built)
# ---------------------------------------------------------------------------

def ref_kissfft():
    p = os.path.join(_HERE, "_ref", "libkissfft_ref.so")
    if not os.path.exists(p):
        return None
    K = C.CDLL(p)
    K.kiss_fftr_alloc.restype = C.c_void_p
    K.kiss_fftr_alloc.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    K.kiss_fftr.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return K


def ref_lna():
    """The reference recogniser's LNA reader (decoder/src/LnaReaderCircular.cc)
    compiled in place (oracle/Makefile) -- None when oracle/_ref is absent."""
    p = os.path.join(_HERE, "_ref", "liblna_ref.so")
    if not os.path.exists(p):
        return None
    R = C.CDLL(p)
    R.ref_lna_read.restype = C.c_int
    R.ref_lna_read.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int,
                               C.POINTER(C.c_int)]
    return R


def ref_lna_read(path: str, max_frames: int, num_states: int, buf_size: int = 8, order: int = 0):
    """[frames x S] float32 as LnaReaderCircular::go_to / log_prob serve them."""
    R = ref_lna()
    out = np.zeros((max_frames, num_states), np.float32)
    S = C.c_int()
    n = R.ref_lna_read(path.encode(), buf_size, order, _p(out, C.c_float), max_frames, C.byref(S))
    if n < 0:
        raise RuntimeError("LnaReaderCircular look-back returned different values")
    if S.value != num_states:
        raise ValueError("header says %d states" % S.value)
    return out[:n]


def ref_aku():
    p = os.path.join(_HERE, "_ref", "libaku_ref.so")
    if not os.path.exists(p):
        return None
    A = C.CDLL(p)
    A.ref_safe_log.restype = C.c_double
    A.ref_safe_log.argtypes = [C.c_double]
    A.ref_str2float.restype = C.c_double
    A.ref_str2float.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    A.ref_str_split.restype = C.c_int
    A.ref_str_split.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
    A.ref_str_clean.restype = C.c_int
    A.ref_str_clean.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    A.ref_module_config_read.restype = C.c_int
    A.ref_module_config_read.argtypes = [C.c_char_p, C.c_long, C.c_char_p, C.c_int, C.POINTER(C.c_long)]
    A.ref_module_config_get_floats.restype = C.c_int
    A.ref_module_config_get_floats.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_float), C.c_int]
    return A


# ---------------------------------------------------------------------------
# full-covariance Gaussians (G2)
# ---------------------------------------------------------------------------

def map_m2v(mat: np.ndarray) -> np.ndarray:
    """LinearAlgebra::map_m2v (aku/LinearAlgebra.cc:219-240): lower triangle in
    row order, off-diagonal elements times sqrt(2)."""
    d = mat.shape[0]
    out = []
    for i in range(d):
        for j in range(i + 1):
            out.append(mat[i, j] if i == j else np.sqrt(2.0) * mat[i, j])
    return np.array(out)


class FullModel:
    """FullCovarianceGaussian pool + mixtures, scored the way
    PDFPool::precompute_likelihoods does without clustering
    (aku/Distributions.cc:2664-2680): exponential feature phi(f) = [f ;
    map_m2v(f f^T)], ll = theta . phi + normalizer + constant with
    theta = [P mu ; -1/2 map_m2v(P)], normalizer = -1/2 mu^T P mu
    (recompute_exponential_parameters, :1529-1547), constant = log sqrt det P
    (set_covariance, :1559-1586).  A covariance that is not SPD (is_spd: all
    eigenvalues > 0, aku/LinearAlgebra.cc:420-434) leaves precision and
    constant at 0: an 'invalid' Gaussian with log-likelihood 0.

    PARITY UNPINNED: no reference goldens; LapackPP (LU inverse, dsyev) is
    replaced by numpy.linalg here."""

    def __init__(self, mean, cov, mix_off, mix_idx, mix_w):
        self.mean = np.ascontiguousarray(mean, np.float64)
        self.cov = np.ascontiguousarray(cov, np.float64)
        self.mix_off = np.ascontiguousarray(mix_off, np.int32)
        self.mix_idx = np.ascontiguousarray(mix_idx, np.int32)
        self.mix_w = np.ascontiguousarray(mix_w, np.float64).copy()
        lib().orc_mixture_normalize(len(self.mix_off) - 1, _p(self.mix_off, C.c_int32),
                                    _p(self.mix_w, C.c_double))
        G, D = self.mean.shape
        self.G, self.D, self.S = G, D, len(self.mix_off) - 1
        self.theta = np.zeros((G, D * (D + 3) // 2))
        self.norm = np.zeros(G)
        self.cst = np.zeros(G)
        self.valid = np.zeros(G, bool)
        for g in range(G):
            c = self.cov[g]
            if np.all(np.linalg.eigvalsh(0.5 * (c + c.T)) > 0):
                P = np.linalg.inv(c)
                self.cst[g] = np.log(np.sqrt(np.linalg.det(P)))
                tm = P @ self.mean[g]
                self.norm[g] = -0.5 * tm @ self.mean[g]
                self.theta[g, :D] = tm
                self.theta[g, D:] = -0.5 * map_m2v(P)
                self.valid[g] = True

    def gauss_loglik(self, frames):
        frames = np.asarray(frames, np.float64)
        out = np.empty((frames.shape[0], self.G))
        for f, x in enumerate(frames):
            phi = np.concatenate([x, map_m2v(np.outer(x, x))])
            out[f] = self.theta @ phi + self.norm + self.cst
        return out

    def score(self, frames):
        ll = self.gauss_loglik(frames)
        lik = np.exp(ll)
        out = np.empty((ll.shape[0], self.S))
        for s in range(self.S):
            a, b = self.mix_off[s], self.mix_off[s + 1]
            l = lik[:, self.mix_idx[a:b]] @ self.mix_w[a:b] if b > a else np.zeros(ll.shape[0])
            out[:, s] = np.log(np.maximum(l, TINY_FOR_LOG))
        return out


def _fullmodel_set_clustering(self, n_clusters: int, pairs, eval_minc: float = 0.0, eval_ming: float = 0.1):
    """PDFPool::read_clustering over a full-covariance pool (aku/Distributions.cc:3114-3170): the centres are
    DiagonalGaussians merged from the members' means and the DIAGONALS of their covariances (Gaussian::merge,
    :853-898, reads get_covariance() and the diagonal target keeps the diagonal of the result)."""
    members = [[] for _ in range(n_clusters)]
    for g, c in pairs:
        members[c].append(g)
    self.cl_off = np.zeros(n_clusters + 1, np.int32)
    self.cl_off[1:] = np.cumsum([len(m) for m in members])
    self.cl_members = np.array([g for m in members for g in m], np.int32)
    self.n_clusters = n_clusters
    self.c_mean = np.zeros((n_clusters, self.D))
    self.c_prec = np.zeros((n_clusters, self.D))
    self.c_cst = np.zeros(n_clusters)
    covdiag = np.ascontiguousarray(np.einsum("gii->gi", self.cov))
    pd = C.c_double
    lib().orc_cluster_centres(self.D, n_clusters, _p(self.cl_off, C.c_int32), _p(self.cl_members, C.c_int32),
                              _p(self.mean, pd), _p(covdiag, pd), _p(self.c_mean, pd), _p(self.c_prec, pd),
                              _p(self.c_cst, pd))
    self.min_clusters = int(eval_minc * n_clusters)
    self.min_gaussians = int(eval_ming * self.G)


def _fullmodel_score_clustered(self, frames: np.ndarray, want_counts: bool = False):
    """[F x S] log state likelihood with the cluster branch of PDFPool::precompute_likelihoods
    (aku/Distributions.cc:2684-2722), which reaches the members through the PDF interface: diagonal centres
    ranked per frame, full-covariance members evaluated exactly for the best clusters, the centre's likelihood
    for the rest (and the > 0 access rule of compute_likelihood, :2636-2644)."""
    frames = np.ascontiguousarray(frames, np.float64)
    F = frames.shape[0]
    exact = np.ascontiguousarray(np.exp(self.gauss_loglik(frames)))
    out = np.empty((F, self.S))
    counts = np.zeros(F, np.int32)
    glik = np.empty(self.G)
    L = lib()
    pd, pi = C.c_double, C.c_int32
    L.orc_pool_likelihoods_clustered_pre.restype = None
    L.orc_pool_likelihoods_clustered_pre.argtypes = [C.c_int, C.c_int64, C.c_int, C.POINTER(pi), C.POINTER(pi),
                                                     C.POINTER(pd), C.POINTER(pd), C.POINTER(pd), C.c_int, C.c_int,
                                                     C.POINTER(pd), C.POINTER(pd), C.POINTER(pd), C.POINTER(pi)]
    cnt = np.zeros(1, np.int32)
    for f in range(F):
        L.orc_pool_likelihoods_clustered_pre(self.D, self.G, self.n_clusters, _p(self.cl_off, pi),
                                             _p(self.cl_members, pi), _p(self.c_mean, pd), _p(self.c_prec, pd),
                                             _p(self.c_cst, pd), self.min_clusters, self.min_gaussians,
                                             _p(np.ascontiguousarray(frames[f]), pd), _p(np.ascontiguousarray(exact[f]), pd),
                                             _p(glik, pd), _p(cnt, pi))
        counts[f] = cnt[0]
        for s in range(self.S):
            a, b = self.mix_off[s], self.mix_off[s + 1]
            l = float(glik[self.mix_idx[a:b]] @ self.mix_w[a:b]) if b > a else 0.0
            out[f, s] = np.log(max(l, TINY_FOR_LOG))
    return (out, counts) if want_counts else out


FullModel.set_clustering = _fullmodel_set_clustering
FullModel.score_clustered = _fullmodel_score_clustered


def write_gk_full(path: str, mean: np.ndarray, cov: np.ndarray, is_full=None, var=None,
                  legacy: bool = False) -> None:
    """'variable' .gk with 'full' (mean + d*d covariance) and 'diag' entries, or
    the legacy 'full_cov' header (aku/Distributions.cc:2823-2906)."""
    G, D = mean.shape
    with open(path, "w") as f:
        f.write("%d %d %s\n" % (G, D, "full_cov" if legacy else "variable"))
        for g in range(G):
            full = True if is_full is None else bool(is_full[g])
            vals = " ".join(repr(float(x)) for x in mean[g]) + " "
            if full:
                vals += " ".join(repr(float(x)) for x in cov[g].ravel())
                f.write(vals + "\n" if legacy else "full " + vals + "\n")
            else:
                vals += " ".join(repr(float(x)) for x in var[g])
                f.write("diag " + vals + "\n")


# ---------------------------------------------------------------------------
# subspace Gaussians (G6): PCGMM / SCGMM.  PARITY UNPINNED -- the reference does not compile
# these (USE_SUBSPACE_COV is never defined, aku/Subspaces.hh needs the un-vendored HCL library)
# and holds no goldens; this restates the text of aku/Subspaces.cc and aku/Distributions.cc.
# ---------------------------------------------------------------------------

def map_v2m(vec: np.ndarray) -> np.ndarray:
    """LinearAlgebra::map_v2m (aku/LinearAlgebra.cc:242-266): inverse of map_m2v, but the factor
    for the off-diagonal elements is `float a = 1/sqrt(2.0)` -- a FLOAT."""
    n = len(vec)
    d = int(0.5 * np.sqrt(1.0 + 8.0 * n) - 0.5)
    a = float(np.float32(1.0 / np.sqrt(2.0)))
    m = np.zeros((d, d))
    pos = 0
    for i in range(d):
        for j in range(i + 1):
            if i == j:
                m[j, j] = vec[pos]
            else:
                m[i, j] = m[j, i] = a * vec[pos]
            pos += 1
    return m


def _spd_determinant(a: np.ndarray) -> float:
    """LinearAlgebra::spd_determinant (aku/LinearAlgebra.cc:25-39): (prod diag chol)^2."""
    c = np.linalg.cholesky(0.5 * (a + a.T))
    det = 1.0
    for i in range(c.shape[0]):
        det *= c[i, i]
    return det * det


class SubspaceModel:
    """A 'variable' .gk pool with precision_subspace / exponential_subspace / pcgmm / scgmm / diag /
    full entries (aku/Distributions.cc:2831-2868) + mixtures, scored as
    PDFPool::precompute_likelihoods' no-clustering branch would (:2663-2682).

    pcgmm  PrecisionConstrainedGaussian::read (:1683-1704): ss_dim, transformed mean m~[dim],
           lambda[ss_dim]; recompute_constant (:1785-1802): P = sum_b lambda_b S_b,
           const = log sqrt spd_det(P) - 1/2 m~^T P^-1 m~.  compute_log_likelihood (:1638-1648):
           q_b = -1/2 f^T S_b f (PrecisionSubspace::precompute, aku/Subspaces.cc:458-469) and
               double result = m_constant + Blas_Dot_Prod(m_transformed_mean, f);
                               + m_ps->dotproduct(m_coeffs);
           -- the ';' after the first line ends the statement, so AS WRITTEN the lambda.q term is
           dropped and the value is const + m~.f (`pcgmm_as_written=True`).  The default here is
           the intended density const + m~.f + lambda.q, which is what the engine scores.
    scgmm  SubspaceConstrainedGaussian::read (:1886-1916): ss_dim, lambda[ss_dim]; psi = sum_b
           lambda_b psi_b, P = sum_b lambda_b P_b with P_b = map_v2m(Pvec_b) (float 1/sqrt 2),
           const = log det(P) - psi^T P^-1 psi - d*log(2*3.1416)  (as written: no halves, the
           literal 3.1416); compute_log_likelihood (:1851-1859) = const + lambda.q with
           q_b = theta_b . [f ; map_m2v(-1/2 f f^T)] (ExponentialSubspace::precompute,
           aku/Subspaces.cc:745-768)."""

    def __init__(self, entries, dim, mix_off, mix_idx, mix_w, pcgmm_as_written=False):
        self.dim = dim
        self.as_written = pcgmm_as_written
        self.mix_off = np.ascontiguousarray(mix_off, np.int32)
        self.mix_idx = np.ascontiguousarray(mix_idx, np.int32)
        self.mix_w = np.ascontiguousarray(mix_w, np.float64).copy()
        lib().orc_mixture_normalize(len(self.mix_off) - 1, _p(self.mix_off, C.c_int32), _p(self.mix_w, C.c_double))
        self.S = len(self.mix_off) - 1
        self.pspace: Dict[int, np.ndarray] = {}      # ssid -> [K][d][d]
        self.espace: Dict[int, np.ndarray] = {}      # ssid -> [K][exp_dim] theta
        self.gauss = []                              # per pool Gaussian: a tuple by kind
        for e in entries:
            kind = e[0]
            if kind == "precision_subspace":
                self.pspace[e[1]] = np.asarray(e[2], np.float64)
            elif kind == "exponential_subspace":
                self.espace[e[1]] = np.asarray(e[2], np.float64)
            elif kind == "pcgmm":
                _, ssid, mt, lam = e
                Sb = self.pspace[ssid]
                lam = np.asarray(lam, np.float64)
                mt = np.asarray(mt, np.float64)
                P = np.zeros((dim, dim))
                for b in range(len(lam)):
                    P += lam[b] * Sb[b]
                const = np.log(np.sqrt(_spd_determinant(P))) - 0.5 * mt @ (np.linalg.inv(P) @ mt)
                self.gauss.append(("pcgmm", ssid, mt, lam, const))
            elif kind == "scgmm":
                _, ssid, lam = e
                th = self.espace[ssid]
                lam = np.asarray(lam, np.float64)
                psi = np.zeros(dim)
                P = np.zeros((dim, dim))
                for b in range(len(lam)):
                    psi += lam[b] * th[b][:dim]
                    P += lam[b] * map_v2m(th[b][dim:])
                cov = np.linalg.inv(P)
                const = np.log(np.prod(np.linalg.eigvalsh(0.5 * (P + P.T))))     # LinearAlgebra::determinant
                const -= psi @ (cov @ psi)
                const -= dim * np.log(2 * 3.1416)
                self.gauss.append(("scgmm", ssid, lam, const))
            elif kind == "diag":
                _, mean, var = e
                mean, var = np.asarray(mean, np.float64), np.asarray(var, np.float64)
                prec = np.where(var > 0, 1.0 / np.where(var > 0, var, 1.0), 0.0)
                c = float(np.prod(prec))
                self.gauss.append(("diag", mean, prec, np.log(np.sqrt(c)) if c > 0 else c))
            else:
                raise ValueError("unknown entry " + kind)
        self.G = len(self.gauss)

    def gauss_loglik(self, frames):
        frames = np.asarray(frames, np.float64)
        out = np.empty((frames.shape[0], self.G))
        d = self.dim
        for fi, f in enumerate(frames):
            qp = {k: np.array([-0.5 * (f @ (Sb @ f)) for Sb in v]) for k, v in self.pspace.items()}
            phi = np.concatenate([f, map_m2v(-0.5 * np.outer(f, f))])
            qe = {k: v @ phi for k, v in self.espace.items()}
            for g, e in enumerate(self.gauss):
                if e[0] == "pcgmm":
                    ll = e[4] + e[2] @ f
                    if not self.as_written:
                        ll += e[3] @ qp[e[1]][:len(e[3])]
                elif e[0] == "scgmm":
                    ll = e[3] + e[2] @ qe[e[1]][:len(e[2])]
                else:
                    diff = f - e[1]
                    ll = -0.5 * float((diff * diff * e[2]).sum()) + e[3]
                out[fi, g] = ll
        return out

    def score(self, frames):
        lik = np.exp(self.gauss_loglik(frames))
        out = np.empty((lik.shape[0], self.S))
        for s in range(self.S):
            a, b = self.mix_off[s], self.mix_off[s + 1]
            l = lik[:, self.mix_idx[a:b]] @ self.mix_w[a:b] if b > a else np.zeros(lik.shape[0])
            out[:, s] = np.log(np.maximum(l, TINY_FOR_LOG))
        return out


def write_gk_subspace(path: str, dim: int, entries) -> None:
    """'variable' .gk file with subspace definitions and pcgmm / scgmm / diag Gaussians in the
    given order (aku/Distributions.cc:2831-2868; PrecisionSubspace::write_subspace,
    aku/Subspaces.cc:169-182; ExponentialSubspace::write_subspace :1201-1215; Gaussian write
    methods :1670-1680, 1876-1883)."""
    n_gauss = sum(1 for e in entries if e[0] in ("pcgmm", "scgmm", "diag"))

    def nums(v):
        return " ".join(repr(float(x)) for x in np.asarray(v).ravel())

    with open(path, "w") as f:
        f.write("%d %d variable\n" % (n_gauss, dim))
        for e in entries:
            if e[0] == "precision_subspace":
                b = np.asarray(e[2])
                f.write("precision_subspace %d %d %d\n" % (e[1], dim, b.shape[0]))
                for k in range(b.shape[0]):
                    f.write(nums(b[k]) + "\n")
            elif e[0] == "exponential_subspace":
                b = np.asarray(e[2])
                f.write("exponential_subspace %d %d %d\n" % (e[1], dim, b.shape[0]))
                for k in range(b.shape[0]):
                    f.write(nums(b[k]) + "\n")
            elif e[0] == "pcgmm":
                f.write("pcgmm %d %d %s %s\n" % (e[1], len(e[3]), nums(e[2]), nums(e[3])))
            elif e[0] == "scgmm":
                f.write("scgmm %d %d %s\n" % (e[1], len(e[2]), nums(e[2])))
            else:
                f.write("diag %s %s\n" % (nums(e[1]), nums(e[2])))
