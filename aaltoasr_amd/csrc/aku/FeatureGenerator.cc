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
}

void FeatureGenerator::close_configuration() {
  if (m_feat) aasr_feat_destroy(m_feat);
  m_feat = nullptr;
  m_block_count = 0;
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
  m_eof_on_last_frame = false;
}

void FeatureGenerator::open_pcm(const int16_t *pcm, int64_t n_samples) {
  if (!m_feat) throw std::string("no feature modules defined");
  m_pcm.assign(pcm, pcm + n_samples);
  m_open = true;
  m_block_count = 0;
  m_eof_on_last_frame = false;
}

void FeatureGenerator::close() {
  m_open = false;
  m_pcm.clear();
  m_block_count = 0;
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
