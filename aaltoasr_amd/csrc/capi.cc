// capi.cc -- extern "C" boundary of libaasr (declared in include/aasr.h).
// Nothing throws across this file; every entry point is wrapped in guarded().
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "feat.h"
#include "gmm.h"
#include "pipeline.h"

namespace aasr {

std::string &last_error() {
  static thread_local std::string msg;
  return msg;
}

aasr_status fail(aasr_status code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

void raise(aasr_status code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  throw Error{code, buf};
}

void require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    raise(AASR_ERR_NO_DEVICE,
          "no HIP device available (%s): this engine has no CPU fallback",
          e != hipSuccess ? hipGetErrorString(e) : "0 devices");
}


}  // namespace aasr

using namespace aasr;

extern "C" {

const char *aasr_last_error(void) { return last_error().c_str(); }
const char *aasr_version(void) { return "aaltoasr_amd 0.1 (gfx950)"; }

int aasr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

aasr_status aasr_set_device(int ordinal) {
  return guarded([&] { AASR_HIP(hipSetDevice(ordinal)); });
}

// ---------------------------------------------------------------- GMM -----

aasr_status aasr_gmm_create_diag(int32_t dim, int32_t num_gaussians, const double *mean,
                                 const double *var, int32_t num_states,
                                 const int32_t *mix_off, const int32_t *mix_idx,
                                 const double *mix_w, aasr_gmm **out) {
  return guarded([&] {
    if (!out || !mean || !var || !mix_off || !mix_idx || !mix_w)
      raise(AASR_ERR_INVALID, "aasr_gmm_create_diag: null argument");
    if (dim <= 0 || num_gaussians <= 0 || num_states <= 0)
      raise(AASR_ERR_INVALID, "aasr_gmm_create_diag: non-positive size");
    *out = nullptr;
    HostModel m;
    m.dim = dim;
    m.G = num_gaussians;
    m.S = num_states;
    m.mean.assign(mean, mean + (size_t)num_gaussians * dim);
    m.var.assign(var, var + (size_t)num_gaussians * dim);
    m.mix_off.assign(mix_off, mix_off + num_states + 1);
    if (m.mix_off[0] != 0) raise(AASR_ERR_INVALID, "mix_off[0] must be 0");
    for (int s = 0; s < num_states; s++)
      if (m.mix_off[s + 1] < m.mix_off[s]) raise(AASR_ERR_INVALID, "mix_off must be non-decreasing");
    size_t K = (size_t)m.mix_off[num_states];
    m.mix_idx.assign(mix_idx, mix_idx + K);
    m.mix_w.assign(mix_w, mix_w + K);
    aasr_gmm *g = new aasr_gmm();
    try {
      AASR_HIP(hipGetDevice(&g->device));
      gmm_build(g, m);
    } catch (...) {
      delete g;
      throw;
    }
    *out = g;
  });
}

aasr_status aasr_gmm_create_full(int32_t dim, int32_t num_gaussians, const double *mean,
                                 const double *cov, int32_t num_states, const int32_t *mix_off,
                                 const int32_t *mix_idx, const double *mix_w, aasr_gmm **out) {
  return guarded([&] {
    if (!out || !mean || !cov || !mix_off || !mix_idx || !mix_w)
      raise(AASR_ERR_INVALID, "aasr_gmm_create_full: null argument");
    if (dim <= 0 || num_gaussians <= 0 || num_states <= 0)
      raise(AASR_ERR_INVALID, "aasr_gmm_create_full: non-positive size");
    *out = nullptr;
    HostModel m;
    m.dim = dim;
    m.G = num_gaussians;
    m.S = num_states;
    const size_t D = (size_t)dim;
    m.mean.assign(mean, mean + (size_t)num_gaussians * D);
    m.cov.assign(cov, cov + (size_t)num_gaussians * D * D);
    m.is_full.assign((size_t)num_gaussians, 1);
    m.var.resize((size_t)num_gaussians * D);
    for (size_t g = 0; g < (size_t)num_gaussians; g++)
      for (size_t i = 0; i < D; i++) m.var[g * D + i] = cov[g * D * D + i * D + i];
    m.mix_off.assign(mix_off, mix_off + num_states + 1);
    if (m.mix_off[0] != 0) raise(AASR_ERR_INVALID, "mix_off[0] must be 0");
    for (int s = 0; s < num_states; s++)
      if (m.mix_off[s + 1] < m.mix_off[s]) raise(AASR_ERR_INVALID, "mix_off must be non-decreasing");
    size_t K = (size_t)m.mix_off[num_states];
    m.mix_idx.assign(mix_idx, mix_idx + K);
    m.mix_w.assign(mix_w, mix_w + K);
    aasr_gmm *g = new aasr_gmm();
    try {
      AASR_HIP(hipGetDevice(&g->device));
      gmm_build(g, m);
    } catch (...) {
      delete g;
      throw;
    }
    *out = g;
  });
}

aasr_status aasr_gmm_create_from_files(const char *gk_path, const char *mc_path,
                                       const char *ph_path, aasr_gmm **out) {
  return guarded([&] {
    if (!out || !gk_path || !mc_path)
      raise(AASR_ERR_INVALID, "aasr_gmm_create_from_files: null argument");
    *out = nullptr;
    HostModel m = read_model_files(gk_path, mc_path, ph_path);
    model_files_fingerprint(gk_path, mc_path, ph_path, m.src_fp);
    m.has_src_fp = true;
    aasr_gmm *g = new aasr_gmm();
    try {
      AASR_HIP(hipGetDevice(&g->device));
      gmm_build(g, m);
    } catch (...) {
      delete g;
      throw;
    }
    *out = g;
  });
}

aasr_status aasr_gmm_write_cache(const aasr_gmm *h, const char *cache_path) {
  return guarded([&] {
    if (!h || !cache_path) raise(AASR_ERR_INVALID, "aasr_gmm_write_cache: null argument");
    write_model_cache(h->host, cache_path);
  });
}

aasr_status aasr_gmm_create_from_cache(const char *cache_path, aasr_gmm **out) {
  return guarded([&] {
    if (!out || !cache_path) raise(AASR_ERR_INVALID, "aasr_gmm_create_from_cache: null argument");
    *out = nullptr;
    HostModel m = read_model_cache(cache_path);
    aasr_gmm *g = new aasr_gmm();
    try {
      AASR_HIP(hipGetDevice(&g->device));
      gmm_build(g, m);
    } catch (...) {
      delete g;
      throw;
    }
    *out = g;
  });
}

aasr_status aasr_gmm_create_from_cache_checked(const char *cache_path, const char *gk_path,
                                               const char *mc_path, const char *ph_path, aasr_gmm **out) {
  return guarded([&] {
    if (!out || !cache_path || !gk_path || !mc_path)
      raise(AASR_ERR_INVALID, "aasr_gmm_create_from_cache_checked: null argument");
    *out = nullptr;
    HostModel m = read_model_cache_checked(cache_path, gk_path, mc_path, ph_path);
    aasr_gmm *g = new aasr_gmm();
    try {
      AASR_HIP(hipGetDevice(&g->device));
      gmm_build(g, m);
    } catch (...) {
      delete g;
      throw;
    }
    *out = g;
  });
}

int32_t aasr_gmm_mixture_size(const aasr_gmm *h, int32_t state) {
  if (!h || state < 0 || state >= h->S) return -1;
  return h->host.mix_off[(size_t)state + 1] - h->host.mix_off[(size_t)state];
}

aasr_status aasr_gmm_mixture_get(const aasr_gmm *h, int32_t state, int32_t *index, double *weight) {
  return guarded([&] {
    if (!h || state < 0 || state >= h->S) raise(AASR_ERR_INVALID, "aasr_gmm_mixture_get: bad state index");
    const HostModel &m = h->host;
    for (int32_t k = m.mix_off[(size_t)state], j = 0; k < m.mix_off[(size_t)state + 1]; k++, j++) {
      if (index) index[j] = m.mix_idx[(size_t)k];
      if (weight) weight[j] = m.mix_w[(size_t)k];
    }
  });
}

aasr_status aasr_gmm_gaussian_get(const aasr_gmm *h, int32_t gaussian, double *mean, double *var) {
  return guarded([&] {
    if (!h || gaussian < 0 || gaussian >= h->G) raise(AASR_ERR_INVALID, "aasr_gmm_gaussian_get: bad Gaussian index");
    const HostModel &m = h->host;
    const size_t D = (size_t)m.dim;
    for (size_t d = 0; d < D; d++) {
      if (mean) mean[d] = m.mean[(size_t)gaussian * D + d];
      if (var) {
        const bool full = m.any_full() && m.is_full[(size_t)gaussian];
        var[d] = full ? m.cov[((size_t)gaussian * D + d) * D + d] : m.var[(size_t)gaussian * D + d];
      }
    }
  });
}

void aasr_gmm_destroy(aasr_gmm *h) { delete h; }
int aasr_gmm_dim(const aasr_gmm *h) { return h ? h->dim : -1; }
int aasr_gmm_num_states(const aasr_gmm *h) { return h ? (int)h->S : -1; }
int aasr_gmm_num_gaussians(const aasr_gmm *h) { return h ? (int)h->G : -1; }
int64_t aasr_gmm_expanded_rows(const aasr_gmm *h) { return h ? h->mix.rows : -1; }

int aasr_gmm_get_precision(const aasr_gmm *h) { return h ? h->precision : -1; }

int aasr_gmm_effective_precision(const aasr_gmm *h) {
  if (!h) return -1;
  if (h->precision != AASR_PREC_F16X2 && h->precision != AASR_PREC_BF16X3) return h->precision;
  if (!h->dim_parts.empty()) return AASR_PREC_F32;   // parts are scored per Gaussian by the f32 pool kernels
  // what the matrix path runs: the centred / general forms have no f16x2 form
  if (h->host.factor_path())   // the factor-row kernels: three bf16 terms, or two fp16 terms where the pool qualifies
    return (h->precision == AASR_PREC_F16X2 && h->full.a16h.p) ? AASR_PREC_F16X2 : AASR_PREC_BF16X3;
  if (aasr::gmm_engine_parts_active(h)) return AASR_PREC_F16X2;   // multi-pivot engine parts: part 0 is on two fp16 terms
  if (h->ill_conditioned) return AASR_PREC_F32_CENTRED;
  const aasr::TrackLayout &L = h->paired.ok ? h->paired : h->tracks;
  if (!L.ok || !L.a16.p) return AASR_PREC_F32;
  return (h->precision == AASR_PREC_F16X2 && L.a16h.p) ? AASR_PREC_F16X2 : AASR_PREC_BF16X3;
}

aasr_status aasr_gmm_precision_states(const aasr_gmm *h, int64_t *states_f16x2, int64_t *states_probe_moved) {
  return guarded([&] {
    if (!h) raise(AASR_ERR_INVALID, "null handle");
    int64_t n = 0;
    if (aasr::gmm_engine_parts_active(h)) {
      for (const auto &part : h->engine_parts) {
        if (part.arith == 2) n += part.states;   // (the plain two-term layout: slab-constant rows count as routed)
        else if (part.arith == 0) {   // the remainder is a model of its own: it may hold two-term states around its pivot
          int64_t k = 0;   // (the remainder's precision follows the handle's: aasr_gmm_set_precision, gmm_plan_engine_parts)
          aasr_gmm_precision_states(part.model.get(), &k, nullptr);
          n += k;
        }
      }
    } else if (aasr_gmm_effective_precision(h) == AASR_PREC_F16X2 && !h->host.factor_path()) {
      const aasr::TrackLayout &L = h->paired.ok ? h->paired : h->tracks;
      n = L.states_f16;
    }
    if (states_f16x2) *states_f16x2 = n;
    if (states_probe_moved) *states_probe_moved = h->f16_probe_moved;
  });
}

aasr_status aasr_gmm_set_precision(aasr_gmm *h, int prec) {
  return guarded([&] {
    if (!h) raise(AASR_ERR_INVALID, "null handle");
    if (prec == AASR_PREC_F32 || prec == AASR_PREC_F32_CENTRED || prec == AASR_PREC_BF16X3 || prec == AASR_PREC_F16X2) {
      if (prec == AASR_PREC_F32_CENTRED && !h->centred_ok)
        raise(AASR_ERR_UNSUPPORTED, "the centred kernel is not available for dimension %d", h->dim);
      if ((prec == AASR_PREC_BF16X3 || prec == AASR_PREC_F16X2) && !((h->paired.ok && h->paired.a16.p) || (h->tracks.ok && h->tracks.a16.p) ||
                                        (h->full.ok && h->full.a16.p) || h->class_routing || !h->dim_parts.empty()))
        raise(AASR_ERR_UNSUPPORTED, "the split-operand (bf16x3 / f16x2) kernels are not available for this model");
      h->precision = prec;
      h->use_bf16x3 = (prec == AASR_PREC_BF16X3 || prec == AASR_PREC_F16X2);
      // the remainder part of a model with engine parts is an ordinary model scored under the handle's precision
      // (queries such as aasr_gmm_precision_states read it and must not write: ADVICE round 5)
      for (auto &part : h->engine_parts)
        if (part.arith == 0 && part.model) {
          part.model->precision = h->precision;
          part.model->use_bf16x3 = h->use_bf16x3;
        }
      return;
    }
    if (prec == AASR_PREC_F64) {
      if (!h->dim_parts.empty() && h->dim > 192)
        raise(AASR_ERR_UNSUPPORTED, "AASR_PREC_F64 is built for feature dimensions <= 192");
      if (h->host.any_full())
        raise(AASR_ERR_UNSUPPORTED, "AASR_PREC_F64 is built for diagonal pools");
      h->precision = prec;
      h->use_bf16x3 = false;
      return;
    }
    raise(AASR_ERR_INVALID, "unknown precision %d", prec);
  });
}

aasr_status aasr_gmm_score_f64_dev(aasr_gmm *h, const double *d_frames, int64_t F, double *d_state_loglik, void *stream) {
  return guarded([&] {
    if (!h || (F > 0 && (!d_frames || !d_state_loglik))) raise(AASR_ERR_INVALID, "aasr_gmm_score_f64_dev: null argument");
    gmm_score_f64_launch(h, d_frames, F, d_state_loglik, 0, (hipStream_t)stream);
  });
}

aasr_status aasr_gmm_score_f64(aasr_gmm *h, const double *frames, int64_t F, double *state_loglik) {
  return guarded([&] {
    if (!h || (F > 0 && (!frames || !state_loglik))) raise(AASR_ERR_INVALID, "aasr_gmm_score_f64: null argument");
    if (F <= 0) return;
    DevBuf<double> d_x, d_o;
    d_x.upload(frames, (size_t)F * h->dim);
    d_o.alloc((size_t)F * h->S);
    gmm_score_f64_launch(h, d_x.p, F, d_o.p, 0, nullptr);
    AASR_HIP(hipDeviceSynchronize());
    AASR_HIP(hipMemcpy(state_loglik, d_o.p, (size_t)F * h->S * sizeof(double), hipMemcpyDeviceToHost));
  });
}

aasr_status aasr_gmm_set_cmllr(aasr_gmm *h, int32_t n_transforms,
                               const int32_t *gauss_to_transform, const double *W) {
  return guarded([&] {
    if (!h) raise(AASR_ERR_INVALID, "null handle");
    if (n_transforms < 0 || (n_transforms > 0 && (!gauss_to_transform || !W)))
      raise(AASR_ERR_INVALID, "aasr_gmm_set_cmllr: bad argument");
    if (n_transforms > 0)
      for (int64_t i = 0; i < h->G; i++)
        if (gauss_to_transform[i] < -1 || gauss_to_transform[i] >= n_transforms)
          raise(AASR_ERR_INVALID, "transform index %d out of range", gauss_to_transform[i]);
    gmm_set_transforms(h, n_transforms, gauss_to_transform, W);
  });
}

aasr_status aasr_gmm_read_clustering(aasr_gmm *h, const char *gcl_path) {
  return guarded([&] {
    if (!h || !gcl_path) raise(AASR_ERR_INVALID, "aasr_gmm_read_clustering: null argument");
    gmm_read_clustering(h, gcl_path);
  });
}

aasr_status aasr_gmm_set_clustering(aasr_gmm *h, int32_t n_clusters, int64_t n_pairs,
                                    const int32_t *gauss_index, const int32_t *cluster_index) {
  return guarded([&] {
    if (!h) raise(AASR_ERR_INVALID, "null handle");
    gmm_set_clustering(h, n_clusters, n_pairs, gauss_index, cluster_index);
  });
}

aasr_status aasr_gmm_set_clustering_min_evals(aasr_gmm *h, double min_clusters,
                                              double min_gaussians) {
  return guarded([&] {
    if (!h) raise(AASR_ERR_INVALID, "null handle");
    gmm_set_clustering_min_evals(h, min_clusters, min_gaussians);
  });
}

int32_t aasr_gmm_num_clusters(const aasr_gmm *h) { return h && h->cl.loaded ? h->cl.C : 0; }

aasr_status aasr_gmm_score_dev(aasr_gmm *h, const float *d_frames, int64_t F,
                               float *d_state_loglik, void *stream) {
  return guarded([&] {
    if (!h || (F > 0 && (!d_frames || !d_state_loglik)))
      raise(AASR_ERR_INVALID, "aasr_gmm_score_dev: null argument");
    gmm_score_launch(h, d_frames, F, d_state_loglik, (hipStream_t)stream);
  });
}

aasr_status aasr_gmm_score(aasr_gmm *h, const float *frames, int64_t F, float *state_loglik) {
  return guarded([&] {
    if (!h || (F > 0 && (!frames || !state_loglik)))
      raise(AASR_ERR_INVALID, "aasr_gmm_score: null argument");
    if (F <= 0) return;
    h->d_frames.ensure((size_t)F * h->dim);
    AASR_HIP(hipMemcpy(h->d_frames.p, frames, (size_t)F * h->dim * sizeof(float), hipMemcpyHostToDevice));
    // Host callers get dense [F x S] rows, but on the device the rows are padded to whole 128-byte
    // lines where the kernel supports a pitch (every output group then is one full cache line:
    // 1.00x write traffic instead of 1.32x for S = 3125); the 2-D copy back drops the padding.
    const int64_t pitch = gmm_score_pitch_ok(h) ? (h->S + 31) / 32 * 32 : h->S;
    h->d_out.ensure((size_t)F * pitch);
    if (pitch != h->S) {
      gmm_score_launch_pitched(h, h->d_frames.p, F, h->d_out.p, pitch, nullptr);
      AASR_HIP(hipMemcpy2D(state_loglik, (size_t)h->S * sizeof(float), h->d_out.p, (size_t)pitch * sizeof(float),
                           (size_t)h->S * sizeof(float), (size_t)F, hipMemcpyDeviceToHost));
    } else {
      gmm_score_launch(h, h->d_frames.p, F, h->d_out.p, nullptr);
      AASR_HIP(hipMemcpy(state_loglik, h->d_out.p, (size_t)F * h->S * sizeof(float), hipMemcpyDeviceToHost));
    }
  });
}

aasr_status aasr_gmm_gauss_loglik_dev(aasr_gmm *h, const float *d_frames, int64_t F,
                                      float *d_gauss_loglik, void *stream) {
  return guarded([&] {
    if (!h || (F > 0 && (!d_frames || !d_gauss_loglik)))
      raise(AASR_ERR_INVALID, "aasr_gmm_gauss_loglik_dev: null argument");
    gmm_gauss_launch(h, d_frames, F, d_gauss_loglik, (hipStream_t)stream);
  });
}

aasr_status aasr_gmm_gauss_loglik(aasr_gmm *h, const float *frames, int64_t F,
                                  float *gauss_loglik) {
  return guarded([&] {
    if (!h || (F > 0 && (!frames || !gauss_loglik)))
      raise(AASR_ERR_INVALID, "aasr_gmm_gauss_loglik: null argument");
    if (F <= 0) return;
    h->d_frames.ensure((size_t)F * h->dim);
    h->d_out.ensure((size_t)F * h->G);
    AASR_HIP(hipMemcpy(h->d_frames.p, frames, (size_t)F * h->dim * sizeof(float), hipMemcpyHostToDevice));
    gmm_gauss_launch(h, h->d_frames.p, F, h->d_out.p, nullptr);
    AASR_HIP(hipMemcpy(gauss_loglik, h->d_out.p, (size_t)F * h->G * sizeof(float), hipMemcpyDeviceToHost));
  });
}

// ---------------------------------------------------------------- LNA -----

int64_t aasr_gmm_score_scratch_floats(const aasr_gmm *h, int64_t F) {
  if (!h || F < 0) return -1;
  return F * gmm_engine_pitch_max(h);   // independent of the precision / clustering / transform state (ADVICE, round 4)
}

aasr_status aasr_gmm_score_lna_dev(aasr_gmm *h, const float *d_frames, int64_t F, int normalize, int lnabytes,
                                   float *d_scratch, uint8_t *d_bytes_out, void *stream) {
  return guarded([&] {
    if (!h || (F > 0 && (!d_frames || !d_scratch || !d_bytes_out)))
      raise(AASR_ERR_INVALID, "aasr_gmm_score_lna_dev: null argument");
    if (lnabytes != 2 && lnabytes != 4) raise(AASR_ERR_INVALID, "lnabytes must be 2 or 4, got %d", lnabytes);
    if (F <= 0) return;
    hipStream_t st = (hipStream_t)stream;
    const int64_t pitch = gmm_engine_pitch(h);
    gmm_score_launch_engine(h, d_frames, F, d_scratch, pitch, st);
    lna_encode_launch(d_scratch, F, (int)h->S, normalize, lnabytes, nullptr, d_bytes_out, st, pitch, gmm_engine_colmap(h));
  });
}

aasr_status aasr_lna_encode_dev(const float *d_state_loglik, int64_t F, int32_t S,
                                int normalize, int lnabytes, float *d_lp_out,
                                uint8_t *d_bytes_out, void *stream) {
  return guarded([&] {
    if (lnabytes != 2 && lnabytes != 4)
      raise(AASR_ERR_INVALID, "lnabytes must be 2 or 4, got %d", lnabytes);
    if (F > 0 && !d_state_loglik) raise(AASR_ERR_INVALID, "aasr_lna_encode_dev: null input");
    if (S <= 0) raise(AASR_ERR_INVALID, "aasr_lna_encode_dev: S must be positive");
    require_device();
    lna_encode_launch(d_state_loglik, F, S, normalize, lnabytes, d_lp_out, d_bytes_out,
                      (hipStream_t)stream, S);
  });
}

aasr_status aasr_lna_encode_dev_pitched(const float *d_state_loglik, int64_t in_pitch, int64_t F, int32_t S,
                                        int normalize, int lnabytes, float *d_lp_out,
                                        uint8_t *d_bytes_out, void *stream) {
  return guarded([&] {
    if (lnabytes != 2 && lnabytes != 4)
      raise(AASR_ERR_INVALID, "lnabytes must be 2 or 4, got %d", lnabytes);
    if (F > 0 && !d_state_loglik) raise(AASR_ERR_INVALID, "aasr_lna_encode_dev_pitched: null input");
    if (S <= 0 || in_pitch < S) raise(AASR_ERR_INVALID, "aasr_lna_encode_dev_pitched: need 0 < S <= in_pitch");
    require_device();
    lna_encode_launch(d_state_loglik, F, S, normalize, lnabytes, d_lp_out, d_bytes_out,
                      (hipStream_t)stream, in_pitch);
  });
}

int aasr_gmm_score_pitch_ok(const aasr_gmm *h) { return h && gmm_score_pitch_ok(h) ? 1 : 0; }

aasr_status aasr_gmm_score_dev_pitched(aasr_gmm *h, const float *d_frames, int64_t F,
                                       float *d_state_loglik, int64_t pitch, void *stream) {
  return guarded([&] {
    if (!h || (F > 0 && (!d_frames || !d_state_loglik)))
      raise(AASR_ERR_INVALID, "aasr_gmm_score_dev_pitched: null argument");
    if (pitch < h->S) raise(AASR_ERR_INVALID, "aasr_gmm_score_dev_pitched: pitch %ld < %ld states", (long)pitch, (long)h->S);
    gmm_score_launch_pitched(h, d_frames, F, d_state_loglik, pitch, (hipStream_t)stream);
  });
}

aasr_status aasr_lna_encode(const float *state_loglik, int64_t F, int32_t S, int normalize,
                            int lnabytes, float *lp_out, uint8_t *bytes_out) {
  return guarded([&] {
    if (lnabytes != 2 && lnabytes != 4)
      raise(AASR_ERR_INVALID, "lnabytes must be 2 or 4, got %d", lnabytes);
    if (S <= 0) raise(AASR_ERR_INVALID, "aasr_lna_encode: S must be positive");
    if (F <= 0) return;
    if (!state_loglik) raise(AASR_ERR_INVALID, "aasr_lna_encode: null input");
    require_device();
    size_t n = (size_t)F * S;
    DevBuf<float> d_in, d_lp;
    DevBuf<uint8_t> d_by;
    d_in.upload(state_loglik, n);
    if (lp_out) d_lp.alloc(n);
    if (bytes_out) d_by.alloc(n * lnabytes);
    lna_encode_launch(d_in.p, F, S, normalize, lnabytes, d_lp.p, d_by.p, nullptr, S);
    if (lp_out) AASR_HIP(hipMemcpy(lp_out, d_lp.p, n * sizeof(float), hipMemcpyDeviceToHost));
    if (bytes_out) AASR_HIP(hipMemcpy(bytes_out, d_by.p, n * lnabytes, hipMemcpyDeviceToHost));
    AASR_HIP(hipDeviceSynchronize());
  });
}

void aasr_lna_header(int32_t num_states, int lnabytes, uint8_t out[5]) {
  uint32_t i = (uint32_t)num_states;
  out[0] = (i >> 24) & 0xff;
  out[1] = (i >> 16) & 0xff;
  out[2] = (i >> 8) & 0xff;
  out[3] = i & 0xff;
  out[4] = (uint8_t)lnabytes;
}

void aasr_free(void *p) { free(p); }

aasr_status aasr_audio_read(const aasr_feat *feat, const char *path, int16_t **pcm, int64_t *n_samples,
                            int32_t *sample_rate) {
  return guarded([&] {
    if (!path || !pcm || !n_samples) raise(AASR_ERR_INVALID, "aasr_audio_read: null argument");
    int rate = 0;
    std::vector<int16_t> v;
    if (feat) {
      const FeatModule &b = feat->mods[0];
      if (b.type != MOD_AUDIOFILE) raise(AASR_ERR_INVALID, "aasr_audio_read: the graph does not start with an audiofile module");
      v = read_audio_file(path, b.raw_audio != 0, b.sample_rate, b.endian == 2, &rate);
    } else {
      v = read_audio_file(path, false, 0, false, &rate);
    }
    int16_t *out = (int16_t *)malloc(std::max<size_t>(v.size(), 1) * sizeof(int16_t));
    if (!out) raise(AASR_ERR_INVALID, "aasr_audio_read: out of memory");
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int16_t));
    *pcm = out;
    *n_samples = (int64_t)v.size();
    if (sample_rate) *sample_rate = rate;
  });
}

aasr_status aasr_audio_decode(const aasr_feat *feat, const void *data, int64_t n_bytes, int16_t **pcm,
                              int64_t *n_samples, int32_t *sample_rate) {
  return guarded([&] {
    if ((!data && n_bytes > 0) || n_bytes < 0 || !pcm || !n_samples)
      raise(AASR_ERR_INVALID, "aasr_audio_decode: invalid argument");
    int rate = 0;
    const std::vector<char> bytes((const char *)data, (const char *)data + n_bytes);
    std::vector<int16_t> v;
    if (feat && feat->mods[0].type == MOD_PRE) {
      v = decode_input_data(feat, bytes, "(memory)");
    } else if (feat) {
      const FeatModule &b = feat->mods[0];
      if (b.type != MOD_AUDIOFILE) raise(AASR_ERR_INVALID, "aasr_audio_decode: the graph does not start with an audiofile or pre module");
      v = decode_audio(bytes, "(memory)", b.raw_audio != 0, b.endian == 2, b.sample_rate, &rate);
    } else {
      v = decode_audio(bytes, "(memory)", false, false, 0, &rate);
    }
    int16_t *out = (int16_t *)malloc(std::max<size_t>(v.size(), 1) * sizeof(int16_t));
    if (!out) raise(AASR_ERR_INVALID, "aasr_audio_decode: out of memory");
    if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(int16_t));
    *pcm = out;
    *n_samples = (int64_t)v.size();
    if (sample_rate) *sample_rate = rate;
  });
}

}  // extern "C"
