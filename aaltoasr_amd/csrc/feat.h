// feat.h -- compiled feature graph (aasr_feat): host-side mirror of
// aku::FeatureGenerator's module DAG plus the device plan that evaluates it.
//
// The reference evaluates the graph one frame at a time through per-module
// ring buffers (FeatureModule::at, aku/FeatureModules.cc:102-158).  Every
// module's output at frame t is a pure function of t, so here each module is
// evaluated ONCE over the frame range its consumers need (output frames plus
// the accumulated Delta / MeanSubtractor look-around, "halo"), for a whole
// batch of utterances per launch.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "../../include/aasr.h"

namespace aasr {

enum ModType {
  MOD_AUDIOFILE, MOD_FFT, MOD_MEL, MOD_POWER, MOD_DCT, MOD_DELTA,
  MOD_NORMALIZATION, MOD_LIN_TRANSFORM, MOD_MERGE, MOD_MEAN_SUBTRACTOR,
  MOD_CONCAT, MOD_VTLN, MOD_SR_NORM, MOD_MEL_POWER, MOD_QUANTEQ, MOD_PRE,
  MOD_HOST  // a user-registered type evaluated on the host (aasr_feat_register_module_type)
};

// One "{ key value ... }" block (aku::ModuleConfig, aku/ModuleConfig.cc).
struct ModuleConfig {
  std::vector<std::string> names, values;
  std::map<std::string, int> index;
  bool exists(const std::string &k) const { return index.count(k) != 0; }
  bool get(const std::string &k, std::string &v) const;
  bool get(const std::string &k, int &v) const;
  bool get(const std::string &k, float &v) const;
  bool get(const std::string &k, std::vector<float> &v) const;
  bool get(const std::string &k, std::vector<std::string> &v) const;
  // parses one block from `text` starting at *pos (just after the "module"
  // line); advances *pos past the closing brace
  void read(const std::string &text, size_t *pos);
  // ModuleConfig::insert (aku/ModuleConfig.cc:225-238): replaces an existing key
  void insert(const std::string &name, const std::string &value);
  // ModuleConfig::set for float / float-vector values: "%g" formatting (:21-26, 49-60)
  void set(const std::string &name, float value);
  void set(const std::string &name, const std::vector<float> &vec);
};

struct FftPlan {
  int nc = 0;  // complex length = window/2
  int ns = 0;
  int radix[16], sublen[16];
  DevBuf<float> hamming;   // [window]
  DevBuf<float> twiddle;   // [nc][2]
  DevBuf<float> stwiddle;  // [nc/2][2]
  DevBuf<int32_t> perm;    // [nc] digit-reversed source index
};

// rows before a mean subtractor's window that its kernel reads: the window sum is anchored on absolute multiples of 8
constexpr int kCmsLead = 7;

struct FeatModule {
  std::string name, type_str;
  ModType type;
  std::vector<int> sources;
  int dim = 0;
  // audiofile
  int sample_rate = 0, width = 0, copy_borders = 1;
  float emph = 0.97f, frame_rate = 125.0f, advance = 0.0f;
  // pre (PreModule, aku/FeatureModules.cc:572-755): the base module reads float
  // feature frames instead of audio.  Its input travels through the same int16
  // buffers as audio, two units per float.
  int legacy_file = 0;
  int raw_audio = 0, endian = 0;  // audiofile: `raw 1`, `endian little|big` (1 / 2)
  // fft
  int magnitude = 1, take_log = 0;
  FftPlan fft;
  // mel
  int root = 0;
  DevBuf<int32_t> mel_off, mel_t;   // CSR over bins: source index t per term
  DevBuf<int32_t> mel_order;        // the bins by falling term count (the fused kernel deals them to lanes in this order)
  DevBuf<float> mel_scale, mel_sum;  // per term scale, per bin float sum
  // dct
  int zeroth = 0;
  DevBuf<float> dct_cos;  // [dim-bias][src_dim]
  // delta
  int delta_width = 2;
  float delta_norm = 10.0f;
  // normalization
  std::vector<float> mean, scale;
  DevBuf<float> d_mean, d_scale;
  // lin_transform
  int src_dim = 0;
  std::vector<float> matrix, bias;
  std::vector<float> orig_matrix, orig_bias;  // as configured (what write_configuration saves)
  bool matrix_defined = false, bias_defined = false;
  DevBuf<float> d_matrix, d_bias;
  // merge
  DevBuf<int32_t> merge_src_col;  // per output column: source slot, column
  // mean_subtractor (config values; the reference stores left+1 / right+1)
  int cms_left = 75, cms_right = 75;
  // vtln (VtlnModule, aku/FeatureModules.cc:1529-1937)
  int use_pwlin = 0, use_slapt = 0, sinc_rad = 8, all_pass = 0;
  bool lanczos = true;
  float pwlin_turn = 0.8f, warp_factor = 1.0f;
  std::vector<float> slapt_params, vtln_bins;
  DevBuf<float> d_vtln_bins;
  // sr_norm (SRNormModule, :1953-2058)
  int in_frames = 0, out_frames = 0, lanczos_order = 4, frame_dim = 1;
  float speech_rate = 1.0f;
  // windowed interpolation shared by vtln and sr_norm: output group i reads
  // sp_len[i] source groups from sp_start[i] with weights sp_coef[i][.]
  DevBuf<int32_t> sp_start, sp_len;
  DevBuf<float> sp_coef;
  int sp_stride = 0;
  // quanteq (QuantEqModule, :2078-2141)
  std::vector<float> q_alpha, q_gamma, q_max, quant_train;
  DevBuf<float> d_q_alpha, d_q_gamma, d_q_max;
  // MOD_HOST: index into the registry of user types, the user's instance, the option block as read
  int host_type = -1;
  void *host_instance = nullptr;
  std::vector<std::string> opt_names, opt_values;
  // look-around this module itself adds around its sources
  int own_left = 0, own_right = 0;
  int lead_left = 0;   // rows of left look-around the module's kernel wants beyond own_left (mean subtractor: kCmsLead)
};

// aasr_feat_register_module_type: the callbacks of one user module type (see include/aasr.h)
struct HostModuleType {
  std::string name;
  aasr_host_module vtbl;
  void *user;
};
const std::vector<HostModuleType> &host_module_types();
int register_host_module_type(const char *name, const aasr_host_module *vtbl, void *user);

}  // namespace aasr

struct aasr_feat {
  int device = 0;
  std::vector<aasr::FeatModule> mods;
  std::map<std::string, int> by_name;
  // scratch, grown on demand
  std::vector<aasr::DevBuf<double>> bufs;  // one per module
  // batch descriptors, one device block in the layout of the pinned staging block (one upload per call, not four):
  // frame_off[n + 1], pcm_off[n + 1] (int64), first[n], eof[n] (int32)
  aasr::DevBuf<char> d_desc;
  aasr::DevBuf<int16_t> d_pcm;
  aasr::DevBuf<float> d_out_f32;
  aasr::DevBuf<double> d_out_f64;
  void *stage_host = nullptr;   // pinned descriptor staging
  size_t stage_cap = 0;
  hipEvent_t stage_event = nullptr;
  bool stage_busy = false;
  ~aasr_feat() {
    for (aasr::FeatModule &m : mods)
      if (m.type == aasr::MOD_HOST && m.host_instance && aasr::host_module_types()[(size_t)m.host_type].vtbl.destroy)
        aasr::host_module_types()[(size_t)m.host_type].vtbl.destroy(m.host_instance);
    if (stage_host) (void)hipHostFree(stage_host);
    if (stage_event) (void)hipEventDestroy(stage_event);
  }
};

namespace aasr {

struct UttBatch {
  int32_t n_utts = 0;
  std::vector<int64_t> frame_off;  // [n+1] prefix of frames emitted
  std::vector<int64_t> pcm_off;    // [n+1] sample offsets in d_pcm
  std::vector<int32_t> first;      // [n] first emitted frame
};

aasr_feat *feat_create(const std::string &cfg_text);
int feat_last_frame(const aasr_feat *h, int64_t n_samples);
int feat_eof_frame(const aasr_feat *h, int64_t n_samples);
void feat_halo(const aasr_feat *h, int target, int *left, int *right);
// Evaluates module `target` for the batch; exactly one of out_f32 / out_f64
// (device pointers, [total_frames x dim]) is written.
void feat_run_batch(aasr_feat *h, const int16_t *d_pcm, const UttBatch &b, int target,
                    float *out_f32, double *out_f64, hipStream_t stream);
void feat_set_parameters(aasr_feat *h, const std::string &module, const std::string &block);
void feat_set_parameters(aasr_feat *h, const std::string &module, const ModuleConfig &c);
// FeatureModule::get_parameters of `module` into c (existing keys are replaced)
void feat_get_parameters(const aasr_feat *h, const std::string &module, ModuleConfig &c);
// FeatureGenerator::write_configuration (aku/FeatureGenerator.cc:222-243)
std::string feat_write_configuration(const aasr_feat *h);

}  // namespace aasr
