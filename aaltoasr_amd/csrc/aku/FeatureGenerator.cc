// FeatureGenerator.cc -- see FeatureGenerator.hh.
#include "FeatureGenerator.hh"

#include <climits>
#include <cstring>

#include <algorithm>
#include <cerrno>

#include "../feat.h"
#include "../pipeline.h"
#include "FeatureModules.hh"

namespace aku {

static void check(aasr_status st) {
  if (st != AASR_OK) throw std::string(aasr_last_error());
}

// live generators, for find_block (a handful at most; not thread-safe like the rest of aku)
static std::vector<const FeatureGenerator *> &registry() {
  static std::vector<const FeatureGenerator *> r;
  return r;
}

FeatureGenerator::FeatureGenerator()
    : m_feat(nullptr), m_open(false), m_eof_on_last_frame(false), m_block_first(0),
      m_block_count(0), m_block_frames(2048), m_block_serial(0) {
  registry().push_back(this);
}

FeatureGenerator::~FeatureGenerator() {
  close();
  close_configuration();
  std::vector<const FeatureGenerator *> &r = registry();
  r.erase(std::remove(r.begin(), r.end(), this), r.end());
}

void FeatureGenerator::load_configuration(FILE *file) {
  std::string text;
  char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, file)) > 0) text.append(buf, n);
  load_configuration_text(text);
}

void FeatureGenerator::write_configuration(FILE *file) {
  if (!m_feat) throw std::string("no feature modules defined");
  char *text = nullptr;
  int64_t len = 0;
  check(aasr_feat_write_config(m_feat, &text, &len));
  fwrite(text, 1, (size_t)len, file);
  aasr_free(text);
}

void FeatureGenerator::load_configuration_text(const std::string &text) {
  if (m_feat) {
    fprintf(stdout, "FeatureGenerator: loading a new feature configuration\n");
    close_configuration();
  }
  build(text, false);
}

// the module blocks of a written configuration: "module" then "{ ... }" (feat_write_configuration)
static std::vector<ModuleConfig> parse_blocks(const std::string &text) {
  std::vector<ModuleConfig> out;
  size_t pos = 0;
  while ((pos = text.find('{', pos)) != std::string::npos) {
    const size_t end = text.find('}', pos);
    if (end == std::string::npos) break;
    ModuleConfig c;
    c.read_text(text.substr(pos, end - pos + 1) + "\n");
    out.push_back(c);
    pos = end + 1;
  }
  return out;
}

void FeatureGenerator::build(const std::string &text, bool keep_modules) {
  aasr_feat *h = nullptr;
  check(aasr_feat_create(text.c_str(), &h));
  if (m_feat) aasr_feat_destroy(m_feat);
  m_feat = h;
  m_block_count = 0;
  m_epoch++;
  const int n = aasr_feat_num_modules(m_feat);
  if (!keep_modules || (int)m_modules.size() != n) {
    m_modules.clear();
    for (int i = 0; i < n; i++) {
      const std::string type = aasr_feat_module_type(m_feat, i);
      FeatureModule *m;
      if (type == "audiofile") m = new AudioFileModule();
      else if (type == "pre") m = new PreModule();
      else if (type == "vtln") m = new VtlnModule();
      else if (type == "normalization") m = new NormalizationModule();
      else if (type == "lin_transform") m = new LinTransformModule();
      else if (type == "quanteq") m = new QuantEqModule();
      else m = new FeatureModule();
      m_modules.emplace_back(m);
    }
  }
  // every module's own configuration as the engine writes it (defaults made explicit)
  char *written = nullptr;
  int64_t len = 0;
  check(aasr_feat_write_config(m_feat, &written, &len));
  const std::vector<ModuleConfig> blocks = parse_blocks(std::string(written, (size_t)len));
  aasr_free(written);
  for (int i = 0; i < n; i++) {
    FeatureModule &m = *m_modules[(size_t)i];
    m.m_gen = this;
    m.m_name = aasr_feat_module_name(m_feat, i);
    m.m_type_str = aasr_feat_module_type(m_feat, i);
    m.m_dim = aasr_feat_module_dim(m_feat, m.m_name.c_str());
    m.m_count = 0;
    m.m_sources.clear();
    m.m_config = (size_t)i < blocks.size() ? blocks[(size_t)i] : ModuleConfig();
    std::vector<std::string> src;
    m.m_config.get("sources", src);
    for (const std::string &sname : src)
      for (int j = 0; j < i; j++)
        if (m_modules[(size_t)j]->m_name == sname) m.m_sources.push_back(m_modules[(size_t)j].get());
    // look-around the module adds itself (aku/FeatureModules.cc: Delta width, MeanSubtractor
    // left/right, Concat left/right)
    m.m_own_offset_left = m.m_own_offset_right = 0;
    int a = 0, b = 0;
    if (m.m_type_str == "delta") {
      if (m.m_config.get("width", a)) m.m_own_offset_left = m.m_own_offset_right = a;
    } else if (m.m_type_str == "mean_subtractor" || m.m_type_str == "concat") {
      if (m.m_config.get("left", a)) m.m_own_offset_left = a;
      if (m.m_config.get("right", b)) m.m_own_offset_right = b;
    }
    m.m_req_offset_left = m.m_req_offset_right = 0;
  }
  // FeatureModule::set_buffer (aku/FeatureModules.cc:38-69): what the consumers ask for, pushed
  // down to the sources
  for (int i = n - 1; i >= 0; i--) {
    FeatureModule &m = *m_modules[(size_t)i];
    for (FeatureModule *src : m.m_sources) {
      src->m_req_offset_left = std::max(src->m_req_offset_left, m.m_req_offset_left + m.m_own_offset_left);
      src->m_req_offset_right = std::max(src->m_req_offset_right, m.m_req_offset_right + m.m_own_offset_right);
    }
  }
}

void FeatureGenerator::reconfigure_module(const std::string &name, const ModuleConfig &config) {
  if (!m_feat) throw std::string("no feature modules defined");
  std::string text;
  bool found = false;
  for (const std::unique_ptr<FeatureModule> &m : m_modules) {
    ModuleConfig c = m->m_name == name ? config : m->m_config;
    if (m->m_name == name) {
      found = true;
      // name, type and sources are the graph's, not the module's, to change
      std::string v;
      if (!c.exists("name")) c.set("name", m->m_name);
      if (!c.exists("type")) c.set("type", m->m_type_str);
      if (!c.exists("sources") && m->m_config.get("sources", v)) c.set("sources", v);
    }
    text += "module\n" + c.text() + "\n";
  }
  if (!found) throw std::string("unknown module requested: ") + name;
  build(text, true);
}

FeatureModule *FeatureGenerator::module(const std::string &name) {
  for (std::unique_ptr<FeatureModule> &m : m_modules)
    if (m->m_name == name) return m.get();
  throw std::string("unknown module requested: ") + name;
}

const FeatureGenerator *FeatureGenerator::find_block(const double *p, int *frame) {
  for (const FeatureGenerator *g : registry()) {
    if (g->m_block_count == 0 || g->m_block.empty()) continue;
    const double *lo = g->m_block.data();
    const int d = aasr_feat_dim(g->m_feat);
    const double *hi = lo + (size_t)g->m_block_count * d;
    if (p >= lo && p < hi && (p - lo) % d == 0) {
      *frame = g->m_block_first + (int)((p - lo) / d);
      return g;
    }
  }
  return nullptr;
}

void FeatureGenerator::print_dot_graph(FILE *file) {
  fprintf(file, "digraph features {\n");
  fprintf(file, "rankdir=RL;\n");
  for (std::unique_ptr<FeatureModule> &m : m_modules) m->print_dot_node(file);
  for (std::unique_ptr<FeatureModule> &m : m_modules)
    for (FeatureModule *src : m->sources())
      fprintf(file, "\t%s -> %s;\n", m->name().c_str(), src->name().c_str());
  fprintf(file, "}\n");
}

// aku/FeatureModules.cc:202-217.  own / req are the reference's numbers; `init` and `buf` describe
// its ring buffers (initial fill offsets, slots) -- the engine evaluates whole blocks, so they are
// printed as the requirement itself and the window it spans.
void FeatureModule::print_dot_node(FILE *file) {
  fprintf(file, "  %s [label=\"%s\\nown=%d-%d\\nreq=%d-%d\\ninit=%d-%d\\nbuf=%d\\n\"]\n", m_name.c_str(),
          m_name.c_str(), m_own_offset_left, m_own_offset_right, m_req_offset_left, m_req_offset_right,
          m_req_offset_left, m_req_offset_right, m_req_offset_left + m_req_offset_right + 1);
}

void FeatureModule::get_config(ModuleConfig &config) {
  if (m_user) {
    config.set("name", m_name);
    config.set("type", m_type_str);
    get_module_config(config);
    return;
  }
  config = m_config;
}

void FeatureModule::set_config(const ModuleConfig &config) {
  if (m_user) {
    set_module_config(config);
    return;
  }
  m_gen->reconfigure_module(m_name, config);
}

// ---- user-defined module types ------------------------------------------------------------
// One instance per module of the type in a loaded graph: the user's object plus proxy objects
// standing for its sources (they only know their dimension and, during generate(), the rows the
// engine handed over).
struct UserModuleGlue {
  std::unique_ptr<FeatureModule> mod;
  std::vector<std::unique_ptr<FeatureModule>> proxies;
  int left = 0, right = 0;

  static int fail(char *err, int32_t len, const std::string &msg) {
    if (err && len > 0) snprintf(err, (size_t)len, "%s", msg.c_str());
    return 1;
  }

  static int configure(void *user, const char *name, const char *block, int32_t n_sources, const int32_t *source_dims,
                       int32_t *dim, int32_t *left, int32_t *right, void **instance, char *err, int32_t err_len) {
    try {
      std::unique_ptr<UserModuleGlue> g(new UserModuleGlue());
      g->mod.reset(((FeatureModule * (*)()) user)());
      g->mod->m_user = true;
      g->mod->m_name = name;
      for (int k = 0; k < n_sources; k++) {
        g->proxies.emplace_back(new FeatureModule());
        g->proxies.back()->m_user = true;
        g->proxies.back()->m_dim = source_dims[k];
        g->mod->m_sources.push_back(g->proxies.back().get());
      }
      ModuleConfig c;
      c.read_text(block);
      std::string t;
      if (c.get("type", t)) g->mod->m_type_str = t;
      g->mod->set_config(c);
      if (g->mod->m_dim <= 0) return fail(err, err_len, std::string("module ") + name + " did not set its dimension");
      *dim = g->mod->m_dim;
      *left = g->left = g->mod->m_own_offset_left;
      *right = g->right = g->mod->m_own_offset_right;
      g->mod->m_buffer.m_dim = g->mod->m_dim;
      *instance = g.release();
      return 0;
    } catch (std::string &e) {
      return fail(err, err_len, e);
    } catch (std::exception &e) {
      return fail(err, err_len, e.what());
    }
  }

  static int generate(void *instance, int32_t frame, const double *const *sources, double *out, char *err,
                      int32_t err_len) {
    UserModuleGlue *g = (UserModuleGlue *)instance;
    try {
      for (size_t k = 0; k < g->proxies.size(); k++) {
        g->proxies[k]->m_eval_rows = sources[k];
        g->proxies[k]->m_eval_first = frame - g->left;
        g->proxies[k]->m_eval_count = g->left + g->right + 1;
      }
      g->mod->m_buffer.m_row = out;
      g->mod->m_buffer.m_frame = frame;
      g->mod->generate(frame);
      g->mod->m_buffer.m_row = nullptr;
      return 0;
    } catch (std::string &e) {
      return fail(err, err_len, e);
    } catch (std::exception &e) {
      return fail(err, err_len, e.what());
    }
  }

  static void destroy(void *instance) { delete (UserModuleGlue *)instance; }
};

void FeatureGenerator::register_user_type(const char *type, FeatureModule *(*factory)()) {
  aasr_host_module v;
  v.configure = &UserModuleGlue::configure;
  v.generate = &UserModuleGlue::generate;
  v.destroy = &UserModuleGlue::destroy;
  if (aasr_feat_register_module_type(type, &v, (void *)factory) != AASR_OK) throw std::string(aasr_last_error());
}

bool BaseFeaModule::eof(int frame) { return frame >= m_gen->eof_frame(); }
int BaseFeaModule::sample_rate(void) { return m_gen->sample_rate(); }
float BaseFeaModule::frame_rate(void) { return m_gen->frame_rate(); }
int BaseFeaModule::last_frame(void) { return m_gen->last_frame(); }

void FeatureModule::set_parameters(const ModuleConfig &config) {
  if (aasr_feat_set_parameters(m_gen->handle(), m_name.c_str(), config.text().c_str()) != AASR_OK)
    throw std::string(aasr_last_error());
  m_gen->invalidate_block();  // every cached frame downstream is stale
}

// VtlnModule::set_warp_factor / set_slapt_warp (aku/FeatureModules.cc:1603-1622) through the
// parameter block; "%.9g" so that the float arrives bit for bit
void VtlnModule::set_warp_factor(float factor) {
  char buf[64];
  snprintf(buf, sizeof buf, "%.9g", (double)factor);
  ModuleConfig c;
  c.set("warp_factor", std::string(buf));
  set_parameters(c);
}
void VtlnModule::set_slapt_warp(std::vector<float> &params) {
  std::string v;
  char buf[64];
  for (size_t i = 0; i < params.size(); i++) {
    snprintf(buf, sizeof buf, i ? " %.9g" : "%.9g", (double)params[i]);
    v += buf;
  }
  ModuleConfig c;
  c.set("slapt_coef", v);
  set_parameters(c);
}
static std::string exact_floats(const std::vector<float> &v) {
  std::string t;
  char buf[64];
  for (size_t i = 0; i < v.size(); i++) {
    snprintf(buf, sizeof buf, i ? " %.9g" : "%.9g", (double)v[i]);
    t += buf;
  }
  return t;
}

// one key of the module's parameter block replaced (an empty vector removes it), the others kept
static void set_one_parameter(FeatureModule *m, const char *key, const std::vector<float> &v) {
  ModuleConfig cur, next;
  m->get_parameters(cur);
  std::vector<std::string> names;
  cur.get_names(names);
  for (const std::string &n : names) {
    if (n == key) continue;
    std::string val;
    cur.get(n, val);
    if (!val.empty()) next.set(n, val);  // an empty vector and a missing key mean the same to set_parameters
  }
  if (!v.empty()) next.set(key, exact_floats(v));
  m->set_parameters(next);
}

// aku/FeatureModules.cc:1123-1133
void NormalizationModule::set_normalization(const std::vector<float> &mean, const std::vector<float> &scale) {
  if ((int)mean.size() != m_dim || (int)scale.size() != m_dim)
    throw std::string("NormalizationModule: The dimension of the new normalization does not match the input dimension");
  ModuleConfig c;
  c.set("mean", exact_floats(mean));
  c.set("scale", exact_floats(scale));
  set_parameters(c);
}

// aku/FeatureModules.cc:1273-1322: an empty vector puts the identity / zero bias back
void LinTransformModule::set_transformation_matrix(std::vector<float> &t) {
  const int src_dim = m_sources.empty() ? m_dim : m_sources.front()->dim();
  if (!t.empty() && (int)t.size() != m_dim * src_dim)
    throw std::string("LinTransformnModule: The dimension of the new transformation matrix does not match the old dimension");
  set_one_parameter(this, "matrix", t);
}
void LinTransformModule::set_transformation_bias(std::vector<float> &b) {
  if (!b.empty() && (int)b.size() != m_dim)
    throw std::string("LinTransformnModule: The dimension of the new bias does not match the output dimension");
  set_one_parameter(this, "bias", b);
}
const std::vector<float> *LinTransformModule::get_transformation_matrix(void) {
  ModuleConfig c;
  get_parameters(c);
  m_transform.clear();
  c.get("matrix", m_transform);
  return &m_transform;
}
const std::vector<float> *LinTransformModule::get_transformation_bias(void) {
  ModuleConfig c;
  get_parameters(c);
  m_bias.clear();
  c.get("bias", m_bias);
  return &m_bias;
}

// aku/FeatureModules.cc:2104-2120
void QuantEqModule::set_alpha(std::vector<float> &alpha) { set_one_parameter(this, "alpha", alpha); }
void QuantEqModule::set_gamma(std::vector<float> &gamma) { set_one_parameter(this, "gamma", gamma); }
void QuantEqModule::set_quant_max(std::vector<float> &quant_max) { set_one_parameter(this, "quant_max", quant_max); }
std::vector<float> QuantEqModule::get_quant_train(void) {
  ModuleConfig c;
  get_config(c);
  std::vector<float> q;
  c.get("quant_train", q);
  return q;
}

float VtlnModule::get_warp_factor(void) {
  ModuleConfig c;
  get_parameters(c);
  float wf = 1.0f;
  c.get("warp_factor", wf);
  return wf;
}

void FeatureModule::get_parameters(ModuleConfig &config) {
  char *text = nullptr;
  int64_t len = 0;
  if (aasr_feat_get_parameters(m_gen->handle(), m_name.c_str(), &text, &len) != AASR_OK)
    throw std::string(aasr_last_error());
  const std::string t(text, (size_t)len);
  aasr_free(text);
  ModuleConfig got;
  got.read_text(t, true);
  config = got;
}

const FeatureVec FeatureModule::at(int frame) {
  if (m_user) {  // a source of a user module while its generate() runs
    if (!m_eval_rows || frame < m_eval_first || frame >= m_eval_first + m_eval_count)
      throw std::string("FeatureModule::at(): frame outside the look-around the module declared "
                        "(m_own_offset_left / m_own_offset_right)");
    return FeatureVec(m_eval_rows + (size_t)(frame - m_eval_first) * m_dim, m_dim, frame, nullptr);
  }
  if (m_count == 0 || m_epoch != m_gen->epoch() || frame < m_first || frame >= m_first + m_count) {
    const std::vector<int16_t> &in = m_gen->input_units();
    const int n = 256;
    m_block.resize((size_t)n * m_dim);
    if (aasr_feat_run_f64(m_gen->handle(), in.data(), (int64_t)in.size(), frame, n, m_name.c_str(),
                          m_block.data()) != AASR_OK)
      throw std::string(aasr_last_error());
    m_first = frame;
    m_count = n;
    m_epoch = m_gen->epoch();
  }
  return FeatureVec(&m_block[(size_t)(frame - m_first) * m_dim], m_dim, frame, nullptr);
}

void FeatureGenerator::close_configuration() {
  if (m_feat) aasr_feat_destroy(m_feat);
  m_feat = nullptr;
  m_modules.clear();
  m_block_count = 0;
  m_epoch++;
}

void FeatureGenerator::open(const std::string &filename) {
  if (!m_feat) throw std::string("no feature modules defined");
  // AudioFileModule::set_fname / PreModule::set_fname (aku/FeatureModules.cc:244-262, 588-631)
  try {
    m_pcm = aasr::read_input_file(m_feat, filename, false);
  } catch (aasr::Error &e) {
    throw std::string(e.msg);
  }
  m_open = true;
  m_block_count = 0;
  m_epoch++;
  m_eof_on_last_frame = false;
}

void FeatureGenerator::open_fd(const int fd, bool) {
  if (m_file != nullptr) close();
  FILE *file = fdopen(fd, "rb");
  if (file == nullptr) throw std::string("could not open fd ") + ": " + strerror(errno);
  open(file, false, false);
}

void FeatureGenerator::open(FILE *file, bool dont_fclose, bool stream) {
  if (!m_feat) throw std::string("no feature modules defined");
  if (m_file != nullptr) close();
  m_file = file;
  m_dont_fclose = dont_fclose;
  m_streaming = false;
  m_stream_eof = false;
  m_pcm.clear();
  if (stream && !aasr_feat_input_is_features(m_feat)) {
    // AudioReader::open(FILE*, rate, close, stream = true): no container probing, raw PCM16 from a
    // non-seekable stream (aku/AudioReader.cc:112-142); samples are fetched as frames are asked for
    m_streaming = true;
    m_stream_big_endian = m_feat->mods[0].endian == 2;
  } else {
    std::vector<char> data;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, file)) > 0) data.insert(data.end(), buf, buf + n);
    // sf_open_fd on the stream, headerless PCM16 when no container is recognised
    // (aku/AudioReader.cc:112-142); feature data for `pre` graphs
    try {
      m_pcm = aasr::decode_input_data(m_feat, data, "(stream)");
    } catch (aasr::Error &e) {
      throw std::string(e.msg);
    }
  }
  m_open = true;
  m_block_count = 0;
  m_epoch++;
  m_eof_on_last_frame = false;
}

// stream mode: read until the samples cover frame `frame` and the graph's look-ahead behind it (so
// that the frame's value is what it would be with the whole file present), or to the stream's end
void FeatureGenerator::stream_read_until(int frame) {
  if (!m_streaming || m_stream_eof) return;
  int halo_l = 0, halo_r = 0;
  aasr_feat_halo(m_feat, &halo_l, &halo_r);
  const int64_t want_frame = frame == INT_MAX ? (int64_t)INT_MAX : (int64_t)frame + halo_r + 1;
  std::vector<unsigned char> raw;
  while (!m_stream_eof && (want_frame == INT_MAX || aasr_feat_eof_frame(m_feat, (int64_t)m_pcm.size()) - 1 < want_frame)) {
    // samples still missing for that frame: one window advance per frame, at least one window
    int64_t missing = 4096;
    if (want_frame != INT_MAX) {
      const double adv = (double)sample_rate() / (double)frame_rate();
      const int64_t have_frames = std::max(-1, aasr_feat_eof_frame(m_feat, (int64_t)m_pcm.size()) - 1);
      missing = std::max<int64_t>(1, (int64_t)((want_frame - have_frames) * adv));
    }
    raw.resize((size_t)missing * 2);
    const size_t got = fread(raw.data(), 2, (size_t)missing, m_file);
    const size_t at = m_pcm.size();
    m_pcm.resize(at + got);
    for (size_t i = 0; i < got; i++) {
      const unsigned lo = m_stream_big_endian ? raw[2 * i + 1] : raw[2 * i];
      const unsigned hi = m_stream_big_endian ? raw[2 * i] : raw[2 * i + 1];
      m_pcm[at + i] = (int16_t)(lo | (hi << 8));
    }
    if (got < (size_t)missing) m_stream_eof = true;
  }
}

void FeatureGenerator::open_pcm(const int16_t *pcm, int64_t n_samples) {
  if (!m_feat) throw std::string("no feature modules defined");
  m_pcm.assign(pcm, pcm + n_samples);
  m_open = true;
  m_block_count = 0;
  m_epoch++;
  m_eof_on_last_frame = false;
}

void FeatureGenerator::close() {
  // aku/FeatureGenerator.cc:86-93: the file handed to open(FILE*, ...) is ours to close unless
  // the caller kept it
  if (m_file != nullptr && !m_dont_fclose) fclose(m_file);
  m_file = nullptr;
  m_open = false;
  m_streaming = false;
  m_stream_eof = false;
  m_pcm.clear();
  m_block_count = 0;
  m_epoch++;
}

int FeatureGenerator::last_frame() {
  stream_read_until(INT_MAX);  // stream mode: only the stream's end tells
  return aasr_feat_last_frame(m_feat, (int64_t)m_pcm.size());
}
int FeatureGenerator::eof_frame() {
  stream_read_until(INT_MAX);
  return aasr_feat_eof_frame(m_feat, (int64_t)m_pcm.size());
}
int FeatureGenerator::sample_rate() { return aasr_feat_sample_rate(m_feat); }
float FeatureGenerator::frame_rate() { return aasr_feat_frame_rate(m_feat); }
int FeatureGenerator::dim() { return aasr_feat_dim(m_feat); }

void FeatureGenerator::fill_block(int frame) {
  const int d = dim();
  int frames = m_block_frames;
  if (m_streaming) {
    if (!m_stream_eof) stream_read_until(frame + m_stream_block_frames - 1);
    if (!m_stream_eof) frames = m_stream_block_frames;  // the rest of the stream has not arrived yet
    // AudioFileModule::generate, frame 0 crossing the end (aku/FeatureModules.cc:405-410)
    if (aasr_feat_eof_frame(m_feat, (int64_t)m_pcm.size()) < 1) throw std::string("audio shorter than frame");
  }
  m_block.resize((size_t)frames * d);
  check(aasr_feat_run_f64(m_feat, m_pcm.data(), (int64_t)m_pcm.size(), frame, frames, nullptr, m_block.data()));
  m_block_f32.resize(m_block.size());
  for (size_t i = 0; i < m_block.size(); i++) m_block_f32[i] = (float)m_block[i];
  m_block_first = frame;
  m_block_count = frames;
  m_block_serial++;
}

const FeatureVec FeatureGenerator::generate(int frame) {
  if (!m_feat || !m_open) throw std::string("FeatureGenerator: no audio opened");
  if (m_block_count == 0 || frame < m_block_first || frame >= m_block_first + m_block_count)
    fill_block(frame);
  // AudioFileModule::eof (aku/FeatureModules.cc:297-303): true from the first
  // frame whose window crosses the end of the file (aasr_feat_eof_frame)
  m_eof_on_last_frame = (!m_streaming || m_stream_eof) &&
                        frame >= aasr_feat_eof_frame(m_feat, (int64_t)m_pcm.size());
  return FeatureVec(&m_block[(size_t)(frame - m_block_first) * dim()], dim(), frame, this);
}

const double *FeatureGenerator::block_f64(int frame, int *first, int *count) const {
  if (m_block_count == 0 || frame < m_block_first || frame >= m_block_first + m_block_count) return nullptr;
  *first = m_block_first;
  *count = m_block_count;
  return m_block.data();
}

const float *FeatureGenerator::block_f32(int frame, int *first, int *count) const {
  if (m_block_count == 0 || frame < m_block_first || frame >= m_block_first + m_block_count)
    return nullptr;
  *first = m_block_first;
  *count = m_block_count;
  return m_block_f32.data();
}

}  // namespace aku
