// feat_capi.cc -- extern "C" entry points of the feature chain (include/aasr.h).
#include <cstring>

#include <algorithm>

#include "feat.h"

using namespace aasr;

namespace {

int resolve_target(const aasr_feat *h, const char *module_name) {
  if (!module_name || !*module_name) return (int)h->mods.size() - 1;
  auto it = h->by_name.find(module_name);
  if (it == h->by_name.end())
    raise(AASR_ERR_INVALID, "unknown module requested: %s", module_name);
  return it->second;
}

UttBatch single(int64_t n_samples, int32_t first_frame, int32_t n_frames) {
  UttBatch b;
  b.n_utts = 1;
  b.frame_off = {0, n_frames};
  b.pcm_off = {0, n_samples};
  b.first = {first_frame};
  return b;
}

template <class T>
void run_host(aasr_feat *h, const int16_t *pcm, int64_t n_samples, int32_t first_frame,
              int32_t n_frames, const char *module_name, T *out) {
  if (!h || !pcm || (n_frames > 0 && !out)) raise(AASR_ERR_INVALID, "aasr_feat_run: null argument");
  if (h->mods[0].type == MOD_PRE && (n_samples & 1))
    raise(AASR_ERR_INVALID, "aasr_feat_run: a pre module takes float input (aasr_feat_run_features)");
  if (n_frames < 0 || n_samples < 0) raise(AASR_ERR_INVALID, "aasr_feat_run: negative size");
  if (n_frames == 0) return;
  const int target = resolve_target(h, module_name);
  const int dim = h->mods[target].dim;
  h->d_pcm.ensure((size_t)n_samples);
  // Only the samples the requested frames can reach go to the device (at their absolute offsets): a
  // caller walking a long file block by block (aku::FeatureGenerator::generate) uploads each part
  // once instead of the whole file per block.  Reach = the graph's accumulated look-around, frames
  // the lower end clamped to [0, last_frame] like AudioFileModule's border copy, two frames of slack.
  int64_t s0 = 0, s1 = n_samples;
  const FeatModule &a = h->mods[0];
  if (a.type != MOD_PRE && n_samples >= a.width + 1) {
    int l = 0, r = 0;
    feat_halo(h, target, &l, &r);
    const int64_t last = std::max(0, feat_eof_frame(h, n_samples) - 1);
    auto clampf = [&](int64_t f) { return std::min(std::max<int64_t>(f, 0), last); };
    const int64_t f_lo = clampf((int64_t)first_frame - l - 2);
    // upper end unclamped (the sample bound does it): without copy_borders, frames past the last
    // whole one read the tail of the file
    const int64_t f_hi = std::max<int64_t>(0, (int64_t)first_frame + n_frames - 1 + r + 2);
    s0 = std::max<int64_t>(0, (int64_t)((double)f_lo * (double)a.advance) - 2);
    s1 = std::min<int64_t>(n_samples, (int64_t)((double)f_hi * (double)a.advance) + a.width + 4);
  }
  if (s1 > s0)
    AASR_HIP(hipMemcpy(h->d_pcm.p + s0, pcm + s0, (size_t)(s1 - s0) * sizeof(int16_t), hipMemcpyHostToDevice));
  UttBatch b = single(n_samples, first_frame, n_frames);
  const size_t n = (size_t)n_frames * dim;
  if (sizeof(T) == 4) {
    h->d_out_f32.ensure(n);
    feat_run_batch(h, h->d_pcm.p, b, target, h->d_out_f32.p, nullptr, nullptr);
    AASR_HIP(hipMemcpy(out, h->d_out_f32.p, n * 4, hipMemcpyDeviceToHost));
  } else {
    h->d_out_f64.ensure(n);
    feat_run_batch(h, h->d_pcm.p, b, target, nullptr, h->d_out_f64.p, nullptr);
    AASR_HIP(hipMemcpy(out, h->d_out_f64.p, n * 8, hipMemcpyDeviceToHost));
  }
}

}  // namespace

extern "C" {

aasr_status aasr_feat_create(const char *cfg_text, aasr_feat **out) {
  return guarded([&] {
    if (!cfg_text || !out) raise(AASR_ERR_INVALID, "aasr_feat_create: null argument");
    *out = nullptr;
    *out = feat_create(cfg_text);
  });
}

void aasr_feat_destroy(aasr_feat *h) { delete h; }
int aasr_feat_dim(const aasr_feat *h) { return h ? h->mods.back().dim : -1; }
float aasr_feat_frame_rate(const aasr_feat *h) { return h ? h->mods[0].frame_rate : 0.0f; }
int aasr_feat_sample_rate(const aasr_feat *h) { return h ? h->mods[0].sample_rate : -1; }

int aasr_feat_module_dim(const aasr_feat *h, const char *module_name) {
  if (!h || !module_name) return -1;
  auto it = h->by_name.find(module_name);
  return it == h->by_name.end() ? -1 : h->mods[it->second].dim;
}

void aasr_feat_halo(const aasr_feat *h, int *left, int *right) {
  int l = 0, r = 0;
  if (h) feat_halo(h, (int)h->mods.size() - 1, &l, &r);
  if (left) *left = l;
  if (right) *right = r;
}

int aasr_feat_last_frame(const aasr_feat *h, int64_t n_samples) {
  return h ? feat_last_frame(h, n_samples) : -1;
}
int aasr_feat_eof_frame(const aasr_feat *h, int64_t n_samples) { return h ? feat_eof_frame(h, n_samples) : -1; }

aasr_status aasr_feat_run(aasr_feat *h, const int16_t *pcm, int64_t n_samples,
                          int32_t first_frame, int32_t n_frames, const char *module_name,
                          float *out) {
  return guarded([&] { run_host<float>(h, pcm, n_samples, first_frame, n_frames, module_name, out); });
}

aasr_status aasr_feat_run_f64(aasr_feat *h, const int16_t *pcm, int64_t n_samples,
                              int32_t first_frame, int32_t n_frames, const char *module_name,
                              double *out) {
  return guarded([&] { run_host<double>(h, pcm, n_samples, first_frame, n_frames, module_name, out); });
}

// PreModule input: float frames [n_values / dim x dim] instead of audio
aasr_status aasr_feat_run_features(aasr_feat *h, const float *features, int64_t n_values,
                                   int32_t first_frame, int32_t n_frames,
                                   const char *module_name, float *out) {
  return guarded([&] {
    if (h && h->mods[0].type != MOD_PRE)
      raise(AASR_ERR_INVALID, "aasr_feat_run_features: the first module is not a pre module");
    run_host<float>(h, (const int16_t *)features, 2 * n_values, first_frame, n_frames, module_name, out);
  });
}

aasr_status aasr_feat_run_features_f64(aasr_feat *h, const float *features, int64_t n_values,
                                       int32_t first_frame, int32_t n_frames,
                                       const char *module_name, double *out) {
  return guarded([&] {
    if (h && h->mods[0].type != MOD_PRE)
      raise(AASR_ERR_INVALID, "aasr_feat_run_features: the first module is not a pre module");
    run_host<double>(h, (const int16_t *)features, 2 * n_values, first_frame, n_frames, module_name, out);
  });
}

aasr_status aasr_feat_write_config(const aasr_feat *h, char **text, int64_t *len) {
  return guarded([&] {
    if (!h || !text || !len) raise(AASR_ERR_INVALID, "aasr_feat_write_config: null argument");
    const std::string t = feat_write_configuration(h);
    char *out = (char *)malloc(t.size() + 1);
    if (!out) raise(AASR_ERR_INVALID, "aasr_feat_write_config: out of memory");
    memcpy(out, t.c_str(), t.size() + 1);
    *text = out;
    *len = (int64_t)t.size();
  });
}

int aasr_feat_input_is_features(const aasr_feat *h) { return h && h->mods[0].type == MOD_PRE ? 1 : 0; }
int aasr_feat_pre_legacy(const aasr_feat *h) { return h && h->mods[0].type == MOD_PRE ? h->mods[0].legacy_file : 0; }
int aasr_feat_input_dim(const aasr_feat *h) { return h ? h->mods[0].dim : -1; }

aasr_status aasr_feat_run_dev(aasr_feat *h, const int16_t *d_pcm, int64_t n_samples,
                              int32_t first_frame, int32_t n_frames, float *d_out, void *stream) {
  return guarded([&] {
    if (!h || !d_pcm || (n_frames > 0 && !d_out))
      raise(AASR_ERR_INVALID, "aasr_feat_run_dev: null argument");
    if (n_frames <= 0) return;
    UttBatch b = single(n_samples, first_frame, n_frames);
    feat_run_batch(h, d_pcm, b, (int)h->mods.size() - 1, d_out, nullptr, (hipStream_t)stream);
  });
}

aasr_status aasr_feat_run_batch_dev(aasr_feat *h, const int16_t *d_pcm, const int64_t *pcm_off,
                                    const int64_t *frame_off, int32_t n_utts, float *d_out,
                                    void *stream) {
  return guarded([&] {
    if (!h || !d_pcm || !pcm_off || !frame_off || !d_out)
      raise(AASR_ERR_INVALID, "aasr_feat_run_batch_dev: null argument");
    if (n_utts <= 0) return;
    UttBatch b;
    b.n_utts = n_utts;
    b.frame_off.assign(frame_off, frame_off + n_utts + 1);
    b.pcm_off.assign(pcm_off, pcm_off + n_utts + 1);
    b.first.assign((size_t)n_utts, 0);
    for (int u = 0; u < n_utts; u++) {
      int64_t ns = pcm_off[u + 1] - pcm_off[u];
      int64_t nf = frame_off[u + 1] - frame_off[u];
      if (nf != (int64_t)feat_eof_frame(h, ns))
        raise(AASR_ERR_INVALID,
              "utterance %d: frame_off says %ld frames but %ld samples give %d (aasr_feat_eof_frame)",
              u, (long)nf, (long)ns, feat_eof_frame(h, ns));
    }
    feat_run_batch(h, d_pcm, b, (int)h->mods.size() - 1, d_out, nullptr, (hipStream_t)stream);
  });
}

aasr_status aasr_feat_set_parameters(aasr_feat *h, const char *module_name,
                                     const char *params_text) {
  return guarded([&] {
    if (!h || !module_name || !params_text)
      raise(AASR_ERR_INVALID, "aasr_feat_set_parameters: null argument");
    feat_set_parameters(h, module_name, params_text);
  });
}

aasr_status aasr_feat_get_parameters(const aasr_feat *h, const char *module_name, char **text,
                                     int64_t *len) {
  return guarded([&] {
    if (!h || !module_name || !text || !len)
      raise(AASR_ERR_INVALID, "aasr_feat_get_parameters: null argument");
    ModuleConfig c;
    feat_get_parameters(h, module_name, c);
    std::string t = "{\n";
    for (size_t i = 0; i < c.names.size(); i++) t += "  " + c.names[i] + " " + c.values[i] + "\n";
    t += "}\n";
    char *out = (char *)malloc(t.size() + 1);
    if (!out) raise(AASR_ERR_INVALID, "aasr_feat_get_parameters: out of memory");
    memcpy(out, t.c_str(), t.size() + 1);
    *text = out;
    *len = (int64_t)t.size();
  });
}

aasr_status aasr_feat_register_module_type(const char *type_name, const aasr_host_module *vtbl, void *user) {
  return guarded([&] { (void)register_host_module_type(type_name, vtbl, user); });
}

int aasr_feat_num_modules(const aasr_feat *h) { return h ? (int)h->mods.size() : -1; }
const char *aasr_feat_module_name(const aasr_feat *h, int index) {
  return h && index >= 0 && index < (int)h->mods.size() ? h->mods[(size_t)index].name.c_str() : nullptr;
}
const char *aasr_feat_module_type(const aasr_feat *h, int index) {
  return h && index >= 0 && index < (int)h->mods.size() ? h->mods[(size_t)index].type_str.c_str() : nullptr;
}

}  // extern "C"
