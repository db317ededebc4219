// SpeakerConfig.cc -- see SpeakerConfig.hh.
#include "SpeakerConfig.hh"

namespace aku {

SpeakerConfig::SpeakerConfig(FeatureGenerator &fea_gen, HmmSet *model)
    : m_fea_gen(fea_gen), m_model(model), m_h(nullptr), m_model_attached(false) {}

SpeakerConfig::~SpeakerConfig() { aasr_spkc_destroy(m_h); }

void SpeakerConfig::ensure() {
  if (m_h) return;
  if (!m_fea_gen.handle()) throw std::string("SpeakerConfig: no feature configuration loaded");
  if (aasr_spkc_create(m_fea_gen.handle(), nullptr, &m_h) != AASR_OK)
    throw std::string(aasr_last_error());
}

// phone_probs reads the speaker file before the model files (aku/phone_probs.cc:94-101):
// the model handle is only fetched when a speaker is set
void SpeakerConfig::attach_model() {
  if (m_model_attached || !m_model) return;
  if (aasr_spkc_set_model(m_h, m_model->handle()) != AASR_OK) throw std::string(aasr_last_error());
  m_model_attached = true;
}

void SpeakerConfig::read_speaker_file(FILE *file) {
  ensure();
  std::string text;
  char buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, file)) > 0) text.append(buf, n);
  if (aasr_spkc_read_text(m_h, text.c_str()) != AASR_OK) throw std::string(aasr_last_error());
}

void SpeakerConfig::set_speaker(const std::string &speaker_id) {
  ensure();
  attach_model();
  const aasr_status st = aasr_spkc_set_speaker(m_h, speaker_id.c_str());
  m_fea_gen.invalidate_block();
  if (m_model) m_model->invalidate_block();
  if (st != AASR_OK) throw std::string(aasr_last_error());
  m_cur_speaker = speaker_id;
  m_cur_utterance = "";
}

void SpeakerConfig::set_utterance(const std::string &utterance_id) {
  ensure();
  attach_model();
  const aasr_status st = aasr_spkc_set_utterance(m_h, utterance_id.c_str());
  m_fea_gen.invalidate_block();
  if (st != AASR_OK) throw std::string(aasr_last_error());
  m_cur_utterance = utterance_id;
}

void SpeakerConfig::write_speaker_file(FILE *file, std::set<std::string> *speakers, std::set<std::string> *utterances) {
  ensure();
  std::vector<const char *> sp, ut;
  if (speakers) for (const std::string &x : *speakers) sp.push_back(x.c_str());
  if (utterances) for (const std::string &x : *utterances) ut.push_back(x.c_str());
  char *text = nullptr;
  int64_t len = 0;
  if (aasr_spkc_write_text(m_h, sp.data(), speakers ? (int32_t)sp.size() : -1, ut.data(),
                           utterances ? (int32_t)ut.size() : -1, &text, &len) != AASR_OK)
    throw std::string(aasr_last_error());
  fwrite(text, 1, (size_t)len, file);
  aasr_free(text);
}

}  // namespace aku
