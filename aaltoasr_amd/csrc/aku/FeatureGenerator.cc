// FeatureGenerator.cc -- see FeatureGenerator.hh.
#include "FeatureGenerator.hh"

#include <climits>
#include <cstring>

#include "../pipeline.h"

namespace aku {

static void check(aasr_status st) {
  if (st != AASR_OK) throw std::string(aasr_last_error());
}

FeatureGenerator::FeatureGenerator()
    : m_feat(nullptr), m_open(false), m_eof_on_last_frame(false), m_block_first(0),
      m_block_count(0), m_block_frames(2048), m_block_serial(0) {}

FeatureGenerator::~FeatureGenerator() { close_configuration(); }

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
  check(aasr_feat_create(text.c_str(), &m_feat));
  m_modules.clear();
  const int n = aasr_feat_num_modules(m_feat);
  m_modules.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    FeatureModule &m = m_modules[(size_t)i];
    m.m_gen = this;
    m.m_name = aasr_feat_module_name(m_feat, i);
    m.m_type = aasr_feat_module_type(m_feat, i);
    m.m_dim = aasr_feat_module_dim(m_feat, m.m_name.c_str());
  }
}

FeatureModule *FeatureGenerator::module(const std::string &name) {
  for (FeatureModule &m : m_modules)
    if (m.m_name == name) return &m;
  throw std::string("unknown module requested: ") + name;
}

void FeatureModule::set_parameters(const ModuleConfig &config) {
  if (aasr_feat_set_parameters(m_gen->handle(), m_name.c_str(), config.text().c_str()) != AASR_OK)
    throw std::string(aasr_last_error());
  m_gen->invalidate_block();  // every cached frame downstream is stale
}

void FeatureModule::get_parameters(ModuleConfig &config) const {
  char *text = nullptr;
  int64_t len = 0;
  if (aasr_feat_get_parameters(m_gen->handle(), m_name.c_str(), &text, &len) != AASR_OK)
    throw std::string(aasr_last_error());
  const std::string t(text, (size_t)len);
  aasr_free(text);
  ModuleConfig got;
  got.read_text(t);
  config = got;
}

const FeatureVec FeatureModule::at(int frame) {
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

void FeatureGenerator::open(FILE *file, bool) {
  if (!m_feat) throw std::string("no feature modules defined");
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
  m_open = true;
  m_block_count = 0;
  m_epoch++;
  m_eof_on_last_frame = false;
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
  m_open = false;
  m_pcm.clear();
  m_block_count = 0;
  m_epoch++;
}

int FeatureGenerator::last_frame() { return aasr_feat_last_frame(m_feat, (int64_t)m_pcm.size()); }
int FeatureGenerator::sample_rate() { return aasr_feat_sample_rate(m_feat); }
float FeatureGenerator::frame_rate() { return aasr_feat_frame_rate(m_feat); }
int FeatureGenerator::dim() { return aasr_feat_dim(m_feat); }

void FeatureGenerator::fill_block(int frame) {
  const int d = dim();
  m_block.resize((size_t)m_block_frames * d);
  check(aasr_feat_run_f64(m_feat, m_pcm.data(), (int64_t)m_pcm.size(), frame, m_block_frames,
                          nullptr, m_block.data()));
  m_block_f32.resize(m_block.size());
  for (size_t i = 0; i < m_block.size(); i++) m_block_f32[i] = (float)m_block[i];
  m_block_first = frame;
  m_block_count = m_block_frames;
  m_block_serial++;
}

const FeatureVec FeatureGenerator::generate(int frame) {
  if (!m_feat || !m_open) throw std::string("FeatureGenerator: no audio opened");
  if (m_block_count == 0 || frame < m_block_first || frame >= m_block_first + m_block_count)
    fill_block(frame);
  // AudioFileModule::eof (aku/FeatureModules.cc:297-303): true from the first
  // frame whose window crosses the end of the file, i.e. last_frame()+1
  m_eof_on_last_frame = frame >= last_frame() + 1;
  return FeatureVec(&m_block[(size_t)(frame - m_block_first) * dim()], dim(), frame, this);
}

const float *FeatureGenerator::block_f32(int frame, int *first, int *count) const {
  if (m_block_count == 0 || frame < m_block_first || frame >= m_block_first + m_block_count)
    return nullptr;
  *first = m_block_first;
  *count = m_block_count;
  return m_block_f32.data();
}

}  // namespace aku
