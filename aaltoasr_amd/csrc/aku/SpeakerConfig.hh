// SpeakerConfig.hh -- aku::SpeakerConfig adapter on the C ABI (aasr_spkc_*).
//
// Same constructor and methods as aku/SpeakerConfig.hh:21-35 for the
// recognition side: read_speaker_file, set_speaker, set_utterance,
// get_cur_speaker / get_cur_utterance; errors are thrown std::string like the
// reference's; write_speaker_file (:26-28) for the adaptation tools that estimate and save
// (aku/vtln.cc).  get_model_transformer belongs to the MLLR trainer and is not built.
#ifndef AKU_AMD_SPEAKERCONFIG_HH
#define AKU_AMD_SPEAKERCONFIG_HH

#include <cstdio>
#include <set>
#include <string>

#include "FeatureGenerator.hh"
#include "HmmSet.hh"

namespace aku {

class SpeakerConfig {
public:
  SpeakerConfig(FeatureGenerator &fea_gen, HmmSet *model = NULL);
  ~SpeakerConfig();

  void read_speaker_file(FILE *file);
  /** aku/SpeakerConfig.cc:156-236; a NULL set writes every entry */
  void write_speaker_file(FILE *file, std::set<std::string> *speakers = NULL,
                          std::set<std::string> *utterances = NULL);

  void set_speaker(const std::string &speaker_id);
  const std::string &get_cur_speaker(void) { return m_cur_speaker; }

  void set_utterance(const std::string &utterance_id);
  const std::string &get_cur_utterance(void) { return m_cur_utterance; }

private:
  void ensure();
  void attach_model();
  FeatureGenerator &m_fea_gen;
  HmmSet *m_model;
  aasr_spkc *m_h;
  bool m_model_attached;
  std::string m_cur_speaker, m_cur_utterance;
};

}  // namespace aku

#endif
