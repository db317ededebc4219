// pipeline.h -- recipe structures shared by pipeline.cc and the aku adapters.
#pragma once
#include <string>
#include <vector>

#include "common.h"

namespace aasr {

// the fields of aku::Recipe::Info the hot path consumes (aku/Recipe.hh)
struct RecipeInfo {
  std::string audio_path, lna_path, speaker_id, utterance_id;
  double start_time = 0, end_time = 0;
};

void recipe_batch_range(int total, int num_batches, int batch_index, int *first, int *count);
std::vector<RecipeInfo> recipe_read(const std::string &text, int num_batches, int batch_index);
std::vector<int16_t> read_audio_file(const std::string &path, bool force_raw, int expect_rate);
std::vector<int16_t> parse_feature_data(const std::vector<char> &data, int dim, bool legacy);
std::vector<int16_t> read_feature_file(const std::string &path, int dim, bool legacy);

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
