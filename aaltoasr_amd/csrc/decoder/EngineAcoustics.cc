// EngineAcoustics.cc -- see EngineAcoustics.hh.
#include "EngineAcoustics.hh"

#include <cstring>

#include "../pipeline.h"

namespace aasr {
std::vector<int16_t> read_input_file(const aasr_feat *feat, const std::string &path, bool force_raw);
}

EngineAcoustics::EngineAcoustics(aasr_feat *feat, aasr_gmm *gmm, int lnabytes, bool normalize,
                                 int block_frames)
    : m_feat(feat), m_gmm(gmm), m_lnabytes(lnabytes), m_block_frames(block_frames > 0 ? block_frames : 1),
      m_normalize(normalize), m_eof_frame(0), m_block_first(0), m_block_count(0) {
  if (lnabytes != 2 && lnabytes != 4) throw std::string("Invalid number of LNA bytes");
  m_num_models = aasr_gmm_num_states(gmm);
}

void EngineAcoustics::open_pcm(const int16_t *pcm, int64_t n_samples) {
  m_pcm.assign(pcm, pcm + n_samples);
  m_eof_frame = aasr_feat_last_frame(m_feat, n_samples) + 1;
  if (m_eof_frame < 1) throw std::string("audio shorter than frame");
  m_block_count = 0;
  m_log_prob = nullptr;
}

void EngineAcoustics::open_file(const std::string &path) {
  std::vector<int16_t> pcm;
  try {
    pcm = aasr::read_input_file(m_feat, path, false);
  } catch (aasr::Error &e) {
    throw std::string(e.msg);
  }
  open_pcm(pcm.data(), (int64_t)pcm.size());
}

void EngineAcoustics::close() {
  m_pcm.clear();
  m_eof_frame = 0;
  m_block_count = 0;
  m_log_prob = nullptr;
}

void EngineAcoustics::fill(int frame) {
  const int last = frame + m_block_frames < m_eof_frame ? frame + m_block_frames : m_eof_frame;
  uint8_t *lna = nullptr;
  int64_t len = 0, frames = 0;
  if (aasr_run_utterance(m_feat, m_gmm, m_pcm.data(), (int64_t)m_pcm.size(), frame, last,
                         m_normalize ? 1 : 0, m_lnabytes, &lna, &len, &frames) != AASR_OK)
    throw std::string(aasr_last_error());
  const size_t n = (size_t)frames * m_num_models;
  m_block.resize(n);
  const uint8_t *body = lna + 5;
  if (m_lnabytes == 4) {
    memcpy(m_block.data(), body, n * 4);
  } else {
    // LnaReaderCircular::go_to (decoder/src/LnaReaderCircular.cc:187-196)
    for (size_t i = 0; i < n; i++) m_block[i] = (body[2 * i] * 256 + body[2 * i + 1]) / -1820.0;
  }
  aasr_free(lna);
  m_block_first = frame;
  m_block_count = (int)frames;
}

bool EngineAcoustics::go_to(int frame) {
  if (m_pcm.empty()) throw std::string("EngineAcoustics::go_to(): nothing opened yet");
  if (frame < 0) throw std::string("EngineAcoustics::go_to(): negative frame");
  if (frame >= m_eof_frame) return false;
  if (m_block_count == 0 || frame < m_block_first || frame >= m_block_first + m_block_count) fill(frame);
  m_log_prob = &m_block[(size_t)(frame - m_block_first) * m_num_models];
  return true;
}
