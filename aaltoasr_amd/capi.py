"""capi.py -- ctypes binding of libaasr.so (the C ABI in include/aasr.h).

The Python binding of the engine: every function of include/aasr.h with its argument types, plus
thin handle classes (Feat, Gmm, SpeakerConfig) that mirror aku::FeatureGenerator / HmmSet /
SpeakerConfig call for call.  Tests, bench.py and the pipeline / shard helpers go through it; it
holds no arithmetic of its own.  It never falls back to a CPU path: if the shared library is
missing it raises, and compute calls without a HIP device return AASR_ERR_NO_DEVICE.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# AASR_LIBDIR: another build of the same library (the ablation build of tools/, aaltoasr_amd/lib_ablation)
LIB_PATH = os.path.join(os.environ.get("AASR_LIBDIR") or os.path.join(HERE, "lib"), "libaasr.so")
HEADER_PATH = os.path.join(HERE, "..", "include", "aasr.h")

AASR_OK = 0
AASR_ERR_INVALID = -1
AASR_ERR_UNSUPPORTED = -2
AASR_ERR_NO_DEVICE = -3
AASR_ERR_IO = -4
AASR_ERR_SHORT_AUDIO = -5


class AasrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("aasr status %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


class RunOptions(C.Structure):
    _fields_ = [("lnabytes", C.c_int32), ("normalize", C.c_int32), ("num_batches", C.c_int32),
                ("batch_index", C.c_int32), ("no_overwrite", C.c_int32), ("raw_audio", C.c_int32),
                ("info", C.c_int32), ("afname", C.c_int32), ("out_dir", C.c_char_p),
                ("speakers", C.c_void_p), ("sort_recipe", C.c_int32)]


class RunStats(C.Structure):
    _fields_ = [("utterances", C.c_int64), ("frames", C.c_int64),
                ("seconds_total", C.c_double), ("seconds_device", C.c_double),
                ("seconds_copy_out", C.c_double)]


class RecipeTiming(C.Structure):
    _fields_ = [("seconds_total", C.c_double), ("wait_reader", C.c_double), ("wait_result_slot", C.c_double),
                ("enqueue", C.c_double), ("wait_copies", C.c_double), ("device", C.c_double),
                ("copy_out", C.c_double), ("writer_threads", C.c_int32), ("usable_cores", C.c_int32),
                ("host_share", C.c_int32)]

    def as_dict(self) -> dict:
        return {k: (round(getattr(self, k), 4) if t is C.c_double else int(getattr(self, k))) for k, t in self._fields_}


_lib: Optional[C.CDLL] = None


def declared_symbols() -> list:
    """Function names declared in include/aasr.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aasr_[a-z0-9_]+)\s*\(", text)))


def lib() -> C.CDLL:
    """Loads libaasr.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libaasr.so is missing (%s): build it with `python -m aaltoasr_amd.build`; "
            "this engine has no CPU fallback" % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; importing it first makes both
    # share one HIP runtime so torch device pointers are valid in libaasr.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    _declare(L)
    _lib = L
    return L


def _declare(L: C.CDLL) -> None:
    i32, i64, f, d, vp, cp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_char_p
    pvp = C.POINTER(vp)
    L.aasr_last_error.restype = cp
    L.aasr_version.restype = cp
    L.aasr_device_count.restype = C.c_int
    L.aasr_set_device.argtypes = [C.c_int]
    L.aasr_feat_create.argtypes = [cp, pvp]
    L.aasr_feat_destroy.argtypes = [vp]
    L.aasr_feat_destroy.restype = None
    L.aasr_feat_dim.argtypes = [vp]
    L.aasr_feat_frame_rate.argtypes = [vp]
    L.aasr_feat_frame_rate.restype = f
    L.aasr_feat_sample_rate.argtypes = [vp]
    L.aasr_feat_module_dim.argtypes = [vp, cp]
    L.aasr_feat_halo.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.aasr_feat_halo.restype = None
    L.aasr_feat_last_frame.argtypes = [vp, i64]
    L.aasr_feat_eof_frame.argtypes = [vp, i64]
    L.aasr_feat_run.argtypes = [vp, vp, i64, i32, i32, cp, vp]
    L.aasr_feat_run_dev.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.aasr_feat_run_f64.argtypes = [vp, vp, i64, i32, i32, cp, vp]
    L.aasr_feat_run_batch_dev.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    L.aasr_feat_set_parameters.argtypes = [vp, cp, cp]
    L.aasr_feat_run_features.argtypes = [vp, vp, i64, i32, i32, cp, vp]
    L.aasr_feat_run_features_f64.argtypes = [vp, vp, i64, i32, i32, cp, vp]
    L.aasr_feat_input_is_features.argtypes = [vp]
    L.aasr_feat_pre_legacy.argtypes = [vp]
    L.aasr_feat_input_dim.argtypes = [vp]
    L.aasr_gmm_create_diag.argtypes = [i32, i32, vp, vp, i32, vp, vp, vp, pvp]
    L.aasr_gmm_create_full.argtypes = [i32, i32, vp, vp, i32, vp, vp, vp, pvp]
    L.aasr_gmm_create_from_files.argtypes = [cp, cp, cp, pvp]
    L.aasr_gmm_destroy.argtypes = [vp]
    L.aasr_gmm_destroy.restype = None
    L.aasr_gmm_dim.argtypes = [vp]
    L.aasr_gmm_num_states.argtypes = [vp]
    L.aasr_gmm_num_gaussians.argtypes = [vp]
    L.aasr_gmm_expanded_rows.argtypes = [vp]
    L.aasr_gmm_expanded_rows.restype = i64
    L.aasr_gmm_write_cache.argtypes = [vp, cp]
    L.aasr_gmm_create_from_cache.argtypes = [cp, C.POINTER(vp)]
    L.aasr_gmm_score_scratch_floats.argtypes = [vp, i64]
    L.aasr_gmm_score_scratch_floats.restype = i64
    L.aasr_gmm_score_lna_dev.argtypes = [vp, vp, i64, C.c_int, C.c_int, vp, vp, vp]
    L.aasr_gmm_create_from_cache_checked.argtypes = [cp, cp, cp, cp, C.POINTER(vp)]
    L.aasr_recipe_frame_limits.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(i32), C.POINTER(i32)]
    L.aasr_recipe_frame_limits.restype = None
    L.aasr_gmm_set_precision.argtypes = [vp, C.c_int]
    L.aasr_gmm_set_cmllr.argtypes = [vp, i32, vp, vp]
    L.aasr_gmm_read_clustering.argtypes = [vp, cp]
    L.aasr_gmm_set_clustering.argtypes = [vp, i32, i64, vp, vp]
    L.aasr_gmm_set_clustering_min_evals.argtypes = [vp, C.c_double, C.c_double]
    L.aasr_gmm_num_clusters.argtypes = [vp]
    L.aasr_gmm_num_clusters.restype = i32
    L.aasr_gmm_score.argtypes = [vp, vp, i64, vp]
    L.aasr_gmm_score_dev.argtypes = [vp, vp, i64, vp, vp]
    L.aasr_gmm_gauss_loglik.argtypes = [vp, vp, i64, vp]
    L.aasr_gmm_gauss_loglik_dev.argtypes = [vp, vp, i64, vp, vp]
    L.aasr_lna_encode.argtypes = [vp, i64, i32, C.c_int, C.c_int, vp, vp]
    L.aasr_lna_encode_dev.argtypes = [vp, i64, i32, C.c_int, C.c_int, vp, vp, vp]
    L.aasr_lna_header.argtypes = [i32, C.c_int, vp]
    L.aasr_lna_header.restype = None
    L.aasr_spkc_create.argtypes = [vp, vp, C.POINTER(vp)]
    L.aasr_spkc_destroy.argtypes = [vp]
    L.aasr_spkc_destroy.restype = None
    L.aasr_spkc_set_model.argtypes = [vp, vp]
    L.aasr_spkc_read_file.argtypes = [vp, cp]
    L.aasr_spkc_read_text.argtypes = [vp, cp]
    L.aasr_spkc_set_speaker.argtypes = [vp, cp]
    L.aasr_spkc_set_utterance.argtypes = [vp, cp]
    L.aasr_spkc_num_changes.argtypes = [vp]
    L.aasr_spkc_num_changes.restype = i64
    L.aasr_recipe_batch_range.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.aasr_run_recipe.argtypes = [vp, vp, cp, C.POINTER(RunOptions), C.POINTER(RunStats)]
    L.aasr_set_host_share.argtypes = [i32]
    L.aasr_host_usable_cores.argtypes = []
    L.aasr_host_usable_cores.restype = i32
    L.aasr_recipe_last_timing.argtypes = [vp, C.POINTER(RecipeTiming)]
    L.aasr_run_utterance.argtypes = [vp, vp, vp, i64, i32, i32, C.c_int, C.c_int,
                                     C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(i64), C.POINTER(i64)]
    L.aasr_lna_read_file.argtypes = [cp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64), C.POINTER(C.POINTER(C.c_float))]
    L.aasr_audio_decode.argtypes = [vp, vp, i64, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(i64), C.POINTER(i32)]
    L.aasr_gmm_score_f64.argtypes = [vp, vp, i64, vp]
    L.aasr_gmm_get_precision.argtypes = [vp]
    L.aasr_gmm_effective_precision.argtypes = [vp]
    L.aasr_gmm_precision_states.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.aasr_free.argtypes = [vp]
    L.aasr_free.restype = None
    L.aasr_feat_get_parameters.argtypes = [vp, cp, C.POINTER(C.c_void_p), C.POINTER(i64)]
    L.aasr_feat_num_modules.argtypes = [vp]
    L.aasr_feat_module_name.argtypes = [vp, C.c_int]
    L.aasr_feat_module_name.restype = cp
    L.aasr_feat_module_type.argtypes = [vp, C.c_int]
    L.aasr_feat_module_type.restype = cp
    L.aasr_gmm_score_pitch_ok.argtypes = [vp]
    L.aasr_gmm_score_dev_pitched.argtypes = [vp, vp, i64, vp, i64, vp]
    L.aasr_lna_encode_dev_pitched.argtypes = [vp, i64, i64, i32, C.c_int, C.c_int, vp, vp, vp]
    L.aasr_feat_write_config.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(i64)]
    L.aasr_recipe_read.argtypes = [cp, i32, i32, C.POINTER(C.c_void_p), C.POINTER(i64)]
    L.aasr_recipe_read_all.argtypes = [cp, i32, i32, i32, C.POINTER(C.c_void_p), C.POINTER(i64)]
    L.aasr_audio_read.argtypes = [vp, cp, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(i64),
                                  C.POINTER(i32)]


def check(status: int) -> None:
    if status != AASR_OK:
        raise AasrError(status, lib().aasr_last_error().decode("utf-8", "replace"))


def _ptr(a) -> int:
    """Address of a numpy array or torch tensor (host or device)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return a.data_ptr()


def _stream_handle(stream) -> Optional[int]:
    if stream is None:
        try:
            import torch
            if torch.cuda.is_available():
                return torch.cuda.current_stream().cuda_stream
        except Exception:
            pass
        return None
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream


class Gmm:
    """Owner of an aasr_gmm handle (HmmSet scoring surface)."""

    def __init__(self, handle: int):
        self._h = handle

    @classmethod
    def from_arrays(cls, mean, var, mix_off, mix_idx, mix_w) -> "Gmm":
        mean = np.ascontiguousarray(mean, np.float64)
        var = np.ascontiguousarray(var, np.float64)
        mix_off = np.ascontiguousarray(mix_off, np.int32)
        mix_idx = np.ascontiguousarray(mix_idx, np.int32)
        mix_w = np.ascontiguousarray(mix_w, np.float64)
        h = C.c_void_p()
        check(lib().aasr_gmm_create_diag(mean.shape[1], mean.shape[0], _ptr(mean), _ptr(var),
                                         len(mix_off) - 1, _ptr(mix_off), _ptr(mix_idx),
                                         _ptr(mix_w), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_full(cls, mean, cov, mix_off, mix_idx, mix_w) -> "Gmm":
        """Full-covariance pool: cov [G, D, D]."""
        mean = np.ascontiguousarray(mean, np.float64)
        cov = np.ascontiguousarray(cov, np.float64)
        mix_off = np.ascontiguousarray(mix_off, np.int32)
        mix_idx = np.ascontiguousarray(mix_idx, np.int32)
        mix_w = np.ascontiguousarray(mix_w, np.float64)
        h = C.c_void_p()
        check(lib().aasr_gmm_create_full(mean.shape[1], mean.shape[0], _ptr(mean), _ptr(cov),
                                         len(mix_off) - 1, _ptr(mix_off), _ptr(mix_idx),
                                         _ptr(mix_w), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_files(cls, gk: str, mc: str, ph: Optional[str] = None) -> "Gmm":
        h = C.c_void_p()
        check(lib().aasr_gmm_create_from_files(gk.encode(), mc.encode(),
                                               ph.encode() if ph else None, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_cache(cls, path: str) -> "Gmm":
        h = C.c_void_p()
        check(lib().aasr_gmm_create_from_cache(path.encode(), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_cache_checked(cls, path: str, gk: str, mc: str, ph: Optional[str] = None) -> "Gmm":
        """The cache, if it was written from exactly these text files (else AasrError 'stale')."""
        h = C.c_void_p()
        check(lib().aasr_gmm_create_from_cache_checked(path.encode(), gk.encode(), mc.encode(),
                                                       ph.encode() if ph else None, C.byref(h)))
        return cls(h.value)

    def write_cache(self, path: str) -> None:
        check(lib().aasr_gmm_write_cache(self._h, path.encode()))

    def close(self) -> None:
        if self._h:
            lib().aasr_gmm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dim(self) -> int:
        return lib().aasr_gmm_dim(self._h)

    @property
    def num_states(self) -> int:
        return lib().aasr_gmm_num_states(self._h)

    @property
    def num_gaussians(self) -> int:
        return lib().aasr_gmm_num_gaussians(self._h)

    @property
    def expanded_rows(self) -> int:
        return lib().aasr_gmm_expanded_rows(self._h)

    def set_cmllr(self, gauss_to_transform=None, W=None) -> None:
        """W [n, D, D+1] with column 0 = bias; gauss_to_transform [G] (-1 = none).
        Call without arguments to remove the adaptation."""
        if W is None:
            check(lib().aasr_gmm_set_cmllr(self._h, 0, None, None))
            return
        W = np.ascontiguousarray(W, np.float64)
        g2t = np.ascontiguousarray(gauss_to_transform, np.int32)
        check(lib().aasr_gmm_set_cmllr(self._h, W.shape[0], _ptr(g2t), _ptr(W)))

    def score_f64(self, frames: np.ndarray) -> np.ndarray:
        """AASR_PREC_F64: double frames [F x D] -> double log state likelihoods [F x S]."""
        frames = np.ascontiguousarray(frames, np.float64)
        self._check_frames(frames)
        out = np.empty((frames.shape[0], self.num_states), np.float64)
        check(lib().aasr_gmm_score_f64(self._h, _ptr(frames), frames.shape[0], _ptr(out)))
        return out

    def read_clustering(self, path: str) -> None:
        """HmmSet::read_clustering (.gcl file)."""
        check(lib().aasr_gmm_read_clustering(self._h, path.encode()))

    def set_clustering(self, n_clusters: int, pairs=()) -> None:
        """In-memory clustering: pairs = [(gauss_index, cluster_index), ...] taken
        literally; n_clusters = 0 removes it."""
        gi = np.ascontiguousarray([p[0] for p in pairs], np.int32)
        ci = np.ascontiguousarray([p[1] for p in pairs], np.int32)
        check(lib().aasr_gmm_set_clustering(self._h, n_clusters, len(gi), _ptr(gi), _ptr(ci)))

    def set_clustering_min_evals(self, min_clusters: float = 1.0, min_gaussians: float = 1.0) -> None:
        """HmmSet::set_clustering_min_evals: ratios of clusters / pool Gaussians."""
        check(lib().aasr_gmm_set_clustering_min_evals(self._h, min_clusters, min_gaussians))

    @property
    def num_clusters(self) -> int:
        return lib().aasr_gmm_num_clusters(self._h)

    def cluster_exact_counts(self, n: int) -> np.ndarray:
        """Diagnostic: clusters evaluated exactly for the first n frames of the last
        clustered scoring pass."""
        L = lib()
        L.aasr_debug_cluster_exact_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        out = np.zeros(n, np.int32)
        if L.aasr_debug_cluster_exact_counts(self._h, _ptr(out), n) != 0:
            raise RuntimeError("no clustered scoring pass to report")
        return out

    def cluster_tie_frames(self) -> int:
        """Diagnostic: frames of the last clustered sub-pass settled by the priority-queue replay."""
        L = lib()
        L.aasr_debug_cluster_tie_frames.argtypes = [C.c_void_p]
        return L.aasr_debug_cluster_tie_frames(self._h)

    def set_precision(self, prec: int) -> None:
        """0 = f32, 1 = f64, 2 = f32 centred form, 3 = bf16x3 split, 4 = f16x2 split where the model is
        eligible, else bf16x3 (default)."""
        check(lib().aasr_gmm_set_precision(self._h, prec))

    def get_precision(self) -> int:
        return int(lib().aasr_gmm_get_precision(self._h))

    def effective_precision(self) -> int:
        """The arithmetic the diagonal scoring path actually runs under the current setting."""
        return int(lib().aasr_gmm_effective_precision(self._h))

    def frame_operand_ms(self, d_frames, reps: int = 10, stream=None) -> float:
        """Diagnostic: ms of one k_frame_operand launch for these frames under the current setting (< 0: not applicable)."""
        L = lib()
        L.aasr_debug_frame_operand_ms.restype = C.c_double
        L.aasr_debug_frame_operand_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        return float(L.aasr_debug_frame_operand_ms(self._h, C.c_void_p(d_frames.data_ptr()), d_frames.shape[0], reps,
                                                   C.c_void_p(stream.cuda_stream if stream is not None else 0)))

    def engine_parts(self):
        """The model's engine parts (multi-pivot internal models, aasr_debug_engine_parts): a dict with `cols` (columns of
        an engine score row) and `parts`, a list of {arith (2: two fp16 terms, 3: three bf16 terms, 0: ordinary model),
        states, pivot_groups, rows}; None when the model is scored by its own layouts."""
        L = lib()
        L.aasr_debug_engine_parts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        out = (C.c_int64 * 14)()
        n = L.aasr_debug_engine_parts(self._h, out, 14)
        if n <= 0:
            return None
        return {"cols": int(out[1]),
                "parts": [{"arith": int(out[2 + 4 * i]), "states": int(out[3 + 4 * i]), "pivot_groups": int(out[4 + 4 * i]),
                           "rows": int(out[5 + 4 * i])} for i in range(n)]}

    def engine_layout(self, part: int = 0):
        """(colmap [S], col0, begin [P], real_end [P], pivots [P, dim]) of engine part `part` (aasr_debug_engine_layout);
        None when there is no such part."""
        L = lib()
        L.aasr_debug_engine_layout.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        colmap = np.zeros(self.num_states, np.int32)
        begin = np.zeros(64, np.int32)
        real_end = np.zeros(64, np.int32)
        piv = np.zeros((64, self.dim), np.float32)
        col0 = C.c_int64(0)
        n = L.aasr_debug_engine_layout(self._h, part, _ptr(colmap), _ptr(begin), _ptr(real_end), _ptr(piv), C.byref(col0))
        if n < 0:
            return None
        return colmap, int(col0.value), begin[:n].copy(), real_end[:n].copy(), piv[:n].copy()

    def engine_plan_note(self) -> str:
        L = lib()
        L.aasr_debug_engine_plan_note.argtypes = [C.c_void_p]
        L.aasr_debug_engine_plan_note.restype = C.c_char_p
        return L.aasr_debug_engine_plan_note(self._h).decode("utf-8", "replace")

    def precision_states(self):
        """(states the two-term fp16 rows cover under the current setting, states the load-time probe took out of
        that form): per-state precision routing, aasr_gmm_precision_states."""
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().aasr_gmm_precision_states(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_layouts(self, mask: int) -> None:
        """Diagnostic: restrict the scoring kernels the launcher may pick (bit 0
        grouped tracks, bit 1 independent tracks, 0 = general LDS-staged)."""
        L = lib()
        L.aasr_debug_set_layouts.argtypes = [C.c_void_p, C.c_int]
        L.aasr_debug_set_layouts.restype = None
        L.aasr_debug_set_layouts(self._h, mask)

    def active_layout(self) -> int:
        L = lib()
        L.aasr_debug_active_layout.argtypes = [C.c_void_p]
        return L.aasr_debug_active_layout(self._h)

    def _check_frames(self, frames) -> None:
        # the C ABI takes a pointer and a frame count: the row width is the caller's promise
        if frames.ndim != 2 or frames.shape[1] != self.dim:
            raise ValueError("frames must be [F x %d], got %s" % (self.dim, tuple(frames.shape)))

    def score(self, frames: np.ndarray) -> np.ndarray:
        frames = np.ascontiguousarray(frames, np.float32)
        self._check_frames(frames)
        out = np.empty((frames.shape[0], self.num_states), np.float32)
        check(lib().aasr_gmm_score(self._h, _ptr(frames), frames.shape[0], _ptr(out)))
        return out

    def score_dev(self, d_frames, d_out, stream=None) -> None:
        check(lib().aasr_gmm_score_dev(self._h, _ptr(d_frames), d_frames.shape[0], _ptr(d_out),
                                       _stream_handle(stream)))

    def score_scratch_floats(self, F: int) -> int:
        return lib().aasr_gmm_score_scratch_floats(self._h, F)

    def score_lna_dev(self, d_frames, d_scratch, d_bytes, normalize: bool = True, lnabytes: int = 2, stream=None) -> None:
        """Frames (device) -> packed LNA rows (device) through the engine's own intermediate layout."""
        check(lib().aasr_gmm_score_lna_dev(self._h, d_frames.data_ptr(), d_frames.shape[0], int(normalize), lnabytes,
                                           d_scratch.data_ptr(), d_bytes.data_ptr(), _stream_handle(stream)))

    def score_pitch_ok(self) -> bool:
        """Whether score_dev_pitched accepts a row pitch other than the state count."""
        return bool(lib().aasr_gmm_score_pitch_ok(self._h))

    def score_dev_pitched(self, d_frames, d_out, pitch: int, stream=None) -> None:
        """d_out: device buffer of frames x pitch floats (pitch >= states)."""
        check(lib().aasr_gmm_score_dev_pitched(self._h, _ptr(d_frames), d_frames.shape[0], _ptr(d_out),
                                               pitch, _stream_handle(stream)))

    def gauss_loglik(self, frames: np.ndarray) -> np.ndarray:
        frames = np.ascontiguousarray(frames, np.float32)
        self._check_frames(frames)
        out = np.empty((frames.shape[0], self.num_gaussians), np.float32)
        check(lib().aasr_gmm_gauss_loglik(self._h, _ptr(frames), frames.shape[0], _ptr(out)))
        return out


def lna_encode(state_loglik: np.ndarray, normalize: bool = True, lnabytes: int = 2):
    x = np.ascontiguousarray(state_loglik, np.float32)
    F, S = x.shape
    lp = np.empty((F, S), np.float32)
    by = np.empty((F, S * lnabytes), np.uint8)
    check(lib().aasr_lna_encode(_ptr(x), F, S, int(normalize), lnabytes, _ptr(lp), _ptr(by)))
    return lp, by


def lna_encode_dev(d_loglik, normalize: bool, lnabytes: int, d_lp=None, d_bytes=None, stream=None,
                   num_states: Optional[int] = None):
    """d_loglik: [F x S], or [F x pitch] with num_states = S < pitch (padded rows)."""
    F, P = d_loglik.shape
    if num_states is None or num_states == P:
        check(lib().aasr_lna_encode_dev(_ptr(d_loglik), F, P, int(normalize), lnabytes, _ptr(d_lp),
                                        _ptr(d_bytes), _stream_handle(stream)))
    else:
        check(lib().aasr_lna_encode_dev_pitched(_ptr(d_loglik), P, F, num_states, int(normalize), lnabytes,
                                                _ptr(d_lp), _ptr(d_bytes), _stream_handle(stream)))


def lna_header(num_states: int, lnabytes: int) -> bytes:
    buf = (C.c_uint8 * 5)()
    lib().aasr_lna_header(num_states, lnabytes, buf)
    return bytes(buf)


class Feat:
    """Owner of an aasr_feat handle (FeatureGenerator surface)."""

    def __init__(self, cfg_text: str):
        h = C.c_void_p()
        check(lib().aasr_feat_create(cfg_text.encode(), C.byref(h)))
        self._h = h.value

    @classmethod
    def from_file(cls, path: str) -> "Feat":
        return cls(open(path).read())

    def write_config(self) -> str:
        """FeatureGenerator::write_configuration: the graph as .cfg text."""
        out = C.c_void_p()
        n = C.c_int64()
        check(lib().aasr_feat_write_config(self._h, C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value).decode()
        finally:
            lib().aasr_free(out)

    def close(self) -> None:
        if self._h:
            lib().aasr_feat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dim(self) -> int:
        return lib().aasr_feat_dim(self._h)

    @property
    def frame_rate(self) -> float:
        return lib().aasr_feat_frame_rate(self._h)

    @property
    def sample_rate(self) -> int:
        return lib().aasr_feat_sample_rate(self._h)

    def module_dim(self, name: str) -> int:
        return lib().aasr_feat_module_dim(self._h, name.encode())

    def halo(self):
        l, r = C.c_int(), C.c_int()
        lib().aasr_feat_halo(self._h, C.byref(l), C.byref(r))
        return l.value, r.value

    def last_frame(self, n_samples: int) -> int:
        return lib().aasr_feat_last_frame(self._h, n_samples)

    def eof_frame(self, n_samples: int) -> int:
        """first frame whose window crosses the end of the input = frames a whole-file run emits"""
        return lib().aasr_feat_eof_frame(self._h, n_samples)

    def run(self, pcm: np.ndarray, first_frame: int, n_frames: int, module: Optional[str] = None,
            dtype=np.float32) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, np.int16)
        dim = self.module_dim(module) if module else self.dim
        if dim < 0:
            raise AasrError(AASR_ERR_INVALID, "unknown module requested: %s" % module)
        out = np.empty((n_frames, dim), dtype)
        fn = lib().aasr_feat_run if dtype == np.float32 else lib().aasr_feat_run_f64
        check(fn(self._h, _ptr(pcm), len(pcm), first_frame, n_frames,
                 module.encode() if module else None, _ptr(out)))
        return out

    def run_features(self, frames: np.ndarray, first_frame: int, n_frames: int,
                     module: Optional[str] = None, dtype=np.float32) -> np.ndarray:
        """Graphs with a `pre` base module: float feature frames [N x dim] in."""
        frames = np.ascontiguousarray(frames, np.float32)
        dim = self.module_dim(module) if module else self.dim
        out = np.empty((n_frames, dim), dtype)
        fn = lib().aasr_feat_run_features if dtype == np.float32 else lib().aasr_feat_run_features_f64
        check(fn(self._h, _ptr(frames), frames.size, first_frame, n_frames,
                 module.encode() if module else None, _ptr(out)))
        return out

    def run_batch_dev(self, d_pcm, pcm_off: np.ndarray, frame_off: np.ndarray, d_out, stream=None):
        pcm_off = np.ascontiguousarray(pcm_off, np.int64)
        frame_off = np.ascontiguousarray(frame_off, np.int64)
        check(lib().aasr_feat_run_batch_dev(self._h, _ptr(d_pcm), _ptr(pcm_off), _ptr(frame_off),
                                            len(pcm_off) - 1, _ptr(d_out), _stream_handle(stream)))

    def set_parameters(self, module: str, block_text: str) -> None:
        check(lib().aasr_feat_set_parameters(self._h, module.encode(), block_text.encode()))

    def get_parameters(self, module: str) -> str:
        """FeatureModule::get_parameters as a "{ name value ... }" block."""
        out = C.c_void_p()
        n = C.c_int64()
        check(lib().aasr_feat_get_parameters(self._h, module.encode(), C.byref(out), C.byref(n)))
        try:
            return C.string_at(out, n.value).decode()
        finally:
            lib().aasr_free(out)

    def modules(self):
        """[(name, type)] in configuration order."""
        L = lib()
        return [(L.aasr_feat_module_name(self._h, i).decode(), L.aasr_feat_module_type(self._h, i).decode())
                for i in range(L.aasr_feat_num_modules(self._h))]


def recipe_batch_range(total: int, num_batches: int, batch_index: int):
    f, n = C.c_int32(), C.c_int32()
    check(lib().aasr_recipe_batch_range(total, num_batches, batch_index, C.byref(f), C.byref(n)))
    return f.value, n.value


def debug_feat_fusion(on: bool) -> None:
    """Diagnostic: False = every feature module runs its own kernel (the fused kernels must match it
    bit for bit)."""
    L = lib()
    L.aasr_debug_feat_fusion.argtypes = [C.c_int]
    L.aasr_debug_feat_fusion.restype = None
    L.aasr_debug_feat_fusion(int(on))


def debug_cluster_heap(on: bool) -> None:
    """Diagnostic: send every frame's cluster selection through the priority-queue replay."""
    L = lib()
    L.aasr_debug_cluster_heap.argtypes = [C.c_int]
    L.aasr_debug_cluster_heap.restype = None
    L.aasr_debug_cluster_heap(int(on))


def recipe_frame_limits(start_time: float, end_time: float, frame_rate: float):
    """(start_frame, end_frame) of aku/phone_probs.cc:199-206, float arithmetic."""
    a, b = C.c_int32(), C.c_int32()
    lib().aasr_recipe_frame_limits(start_time, end_time, frame_rate, C.byref(a), C.byref(b))
    return a.value, b.value


def recipe_read(text, num_batches: int = 0, batch_index: int = 0):
    """Recipe::read on the host: list of (audio, lna, speaker, utterance, start_time, end_time)."""
    out = C.c_void_p()
    n = C.c_int64()
    raw = text if isinstance(text, bytes) else text.encode()
    check(lib().aasr_recipe_read(raw, num_batches, batch_index, C.byref(out), C.byref(n)))
    try:
        table = C.string_at(out, n.value)
    finally:
        lib().aasr_free(out)
    rows = []
    for line in table.split(b"\n")[:-1]:
        f = line.split(b"\x1f")
        # the times are float fields printed with 9 significant digits: exact through float32
        rows.append(tuple(x.decode("latin-1") for x in f[:4]) +
                    (float(np.float32(float(f[4]))), float(np.float32(float(f[5])))))
    return rows


def recipe_read_all(text, num_batches: int = 0, batch_index: int = 0, cluster_speakers: bool = False):
    """Recipe::read with every Info field: list of 13-tuples (audio, alt-audio, transcript,
    alignment, hmmnet, den-hmmnet, lna, start_time, end_time, start_line, end_line, speaker,
    utterance)."""
    out = C.c_void_p()
    n = C.c_int64()
    raw = text if isinstance(text, bytes) else text.encode()
    check(lib().aasr_recipe_read_all(raw, num_batches, batch_index, int(cluster_speakers), C.byref(out), C.byref(n)))
    try:
        table = C.string_at(out, n.value)
    finally:
        lib().aasr_free(out)
    rows = []
    for line in table.split(b"\n")[:-1]:
        f = [x.decode("latin-1") for x in line.split(b"\x1f")]
        rows.append(tuple(f[:7]) + (float(np.float32(float(f[7]))), float(np.float32(float(f[8]))),
                                    int(f[9]), int(f[10]), f[11], f[12]))
    return rows


def lna_read_file(path: str):
    """LnaReaderCircular's view of an LNA file (1-, 2- or 4-byte): (float32 [frames x states], lnabytes)."""
    out = C.POINTER(C.c_float)()
    S = C.c_int32()
    nb = C.c_int32()
    F = C.c_int64()
    check(lib().aasr_lna_read_file(path.encode(), C.byref(S), C.byref(nb), C.byref(F), C.byref(out)))
    try:
        n = F.value * S.value
        lp = np.ctypeslib.as_array(out, shape=(max(n, 1),))[:n].copy().reshape(F.value, S.value)
    finally:
        lib().aasr_free(out)
    return lp, nb.value


def audio_read(path: str, feat: Optional["Feat"] = None):
    """AudioReader::open + read (host only).  Returns (int16 samples, sample rate)."""
    out = C.POINTER(C.c_int16)()
    n = C.c_int64()
    rate = C.c_int32()
    check(lib().aasr_audio_read(feat._h if feat is not None else None, path.encode(), C.byref(out),
                                C.byref(n), C.byref(rate)))
    try:
        pcm = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[:n.value].copy()
    finally:
        lib().aasr_free(out)
    return pcm, rate.value


def audio_decode(data: bytes, feat: Optional["Feat"] = None):
    """aasr_audio_decode: the file decoding of audio_read for bytes already in memory."""
    out = C.POINTER(C.c_int16)()
    n = C.c_int64()
    rate = C.c_int32()
    buf = C.create_string_buffer(data, len(data))
    check(lib().aasr_audio_decode(feat._h if feat is not None else None, C.cast(buf, C.c_void_p), len(data),
                                  C.byref(out), C.byref(n), C.byref(rate)))
    try:
        pcm = np.ctypeslib.as_array(out, shape=(max(n.value, 1),))[:n.value].copy()
    finally:
        lib().aasr_free(out)
    return pcm, rate.value


def run_utterance(feat: Feat, gmm: Gmm, pcm: np.ndarray, start_frame: int = 0, end_frame: int = 0,
                  normalize: bool = True, lnabytes: int = 2):
    """Returns (lna file image as bytes, number of frames)."""
    pcm = np.ascontiguousarray(pcm, np.int16)
    out = C.POINTER(C.c_uint8)()
    n = C.c_int64()
    frames = C.c_int64()
    check(lib().aasr_run_utterance(feat._h, gmm._h, _ptr(pcm), len(pcm), start_frame, end_frame,
                                   int(normalize), lnabytes, C.byref(out), C.byref(n), C.byref(frames)))
    try:
        data = C.string_at(out, n.value)
    finally:
        lib().aasr_free(out)
    return data, frames.value


class SpeakerConfig:
    """aku::SpeakerConfig on the engine's handles (aasr_spkc_*)."""

    def __init__(self, feat: "Feat", gmm: Optional["Gmm"] = None):
        self._h = C.c_void_p()
        self._keep = (feat, gmm)
        check(lib().aasr_spkc_create(feat._h, gmm._h if gmm is not None else None, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.aasr_spkc_destroy(self._h)
            self._h = None

    def set_model(self, gmm: "Gmm") -> None:
        self._keep = (self._keep[0], gmm)
        check(lib().aasr_spkc_set_model(self._h, gmm._h))

    def read_file(self, path: str) -> None:
        check(lib().aasr_spkc_read_file(self._h, path.encode()))

    def read_text(self, text: str) -> None:
        check(lib().aasr_spkc_read_text(self._h, text.encode()))

    def set_speaker(self, speaker_id: str = "") -> None:
        check(lib().aasr_spkc_set_speaker(self._h, speaker_id.encode()))

    def set_utterance(self, utterance_id: str = "") -> None:
        check(lib().aasr_spkc_set_utterance(self._h, utterance_id.encode()))

    @property
    def num_changes(self) -> int:
        return lib().aasr_spkc_num_changes(self._h)


def run_recipe(feat: Feat, gmm: Gmm, recipe_path: str, lnabytes: int = 2, normalize: bool = True,
               num_batches: int = 0, batch_index: int = 0, no_overwrite: bool = False,
               raw_audio: bool = False, info: int = 0, afname: bool = False,
               out_dir: Optional[str] = None, speakers: Optional[SpeakerConfig] = None,
               sort_recipe: bool = False) -> RunStats:
    opt = RunOptions(lnabytes, int(normalize), num_batches, batch_index, int(no_overwrite),
                     int(raw_audio), info, int(afname), out_dir.encode() if out_dir else None,
                     speakers._h if speakers is not None else None, int(sort_recipe))
    st = RunStats()
    check(lib().aasr_run_recipe(feat._h, gmm._h, recipe_path.encode(), C.byref(opt), C.byref(st)))
    return st


def recipe_last_timing(gmm: Gmm) -> dict:
    t = RecipeTiming()
    check(lib().aasr_recipe_last_timing(gmm._h, C.byref(t)))
    return t.as_dict()


def set_host_share(processes: int) -> None:
    check(lib().aasr_set_host_share(int(processes)))
