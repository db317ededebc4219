// pipeline.h -- recipe structures shared by pipeline.cc and the aku adapters.
#pragma once
#include <string>
#include <vector>

#include "common.h"

namespace aasr {

// the fields of aku::Recipe::Info the hot path consumes (aku/Recipe.hh)
struct RecipeInfo {
  std::string audio_path, alt_audio_path, transcript_path, alignment_path, hmmnet_path, den_hmmnet_path;
  std::string lna_path, speaker_id, utterance_id;
  int start_line = 0, end_line = 0;
  // float, as Recipe::Info (aku/Recipe.hh:48-49): atof narrowed on assignment
  float start_time = 0, end_time = 0;
};

std::string str_clean(const std::string &s, const char *chars);
std::vector<std::string> str_split(const std::string &s, const char *delims, bool group, int num_fields = 0);
void recipe_batch_range(int total, int num_batches, int batch_index, int *first, int *count);
std::vector<RecipeInfo> recipe_read(const std::string &text, int num_batches, int batch_index,
                                    bool cluster_speakers = false);
// audio_reader.cc
std::vector<int16_t> read_audio_file(const std::string &path, bool force_raw, int expect_rate,
                                     bool big_endian = false, int *rate_out = nullptr);
std::vector<int16_t> decode_audio(const std::vector<char> &data, const std::string &name, bool force_raw,
                                  bool big_endian, int expect_rate, int *rate_out = nullptr);
std::vector<int16_t> parse_feature_data(const std::vector<char> &data, int dim, bool legacy);
std::vector<int16_t> read_feature_file(const std::string &path, int dim, bool legacy);

}  // namespace aasr
struct aasr_feat;
namespace aasr {
// FeatureGenerator::open(filename) / open(FILE*) for whichever base module the graph has
// (audiofile: container detection + the module's `raw` / `endian` options; pre: feature file)
std::vector<int16_t> read_input_file(const aasr_feat *feat, const std::string &path, bool force_raw);
std::vector<int16_t> decode_input_data(const aasr_feat *feat, const std::vector<char> &data,
                                       const std::string &name);

}  // namespace aasr

#include <functional>
struct aasr_spkc;
namespace aasr {
// speaker_config.cc
void spkc_read_text(aasr_spkc *h, const std::string &text);
void spkc_set_speaker(aasr_spkc *h, const std::string &speaker_id);
void spkc_set_utterance(aasr_spkc *h, const std::string &utterance_id);
// called right before any module's device parameters change
void spkc_set_before_change(aasr_spkc *h, std::function<void()> fn);

}  // namespace aasr
