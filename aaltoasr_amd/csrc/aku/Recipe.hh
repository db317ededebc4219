// Recipe.hh -- aku::Recipe (aku/Recipe.hh:36-118) on the engine's recipe reader
// (aasr_recipe_read_all, csrc/pipeline.cc: the reference's line cleaning, key persistence,
// batch walk and cluster_speakers rule): read(FILE*, num_batches, batch_index,
// cluster_speakers), infos with every Info field, clear(), sort_infos().
// Info::init_phn_files (aku/Recipe.cc:152-194) is declared when the aligners' PhnReader.hh is on
// the include path -- i.e. when the reference's own aku/PhnReader.{hh,cc} / aku/Viterbi.{hh,cc} are
// being compiled against these adapters (oracle/Makefile: align_refmain); init_hmmnet_files
// (:197-229) likewise when aku/HmmNetBaumWelch.hh is (logl_refmain).
#ifndef AKU_AMD_RECIPE_HH
#define AKU_AMD_RECIPE_HH

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "FeatureGenerator.hh"
#if defined(__has_include)
#if __has_include("PhnReader.hh")
#include "PhnReader.hh"
#define AKU_AMD_HAVE_PHNREADER 1
#endif
#if __has_include("HmmNetBaumWelch.hh")
#include "HmmNetBaumWelch.hh"
#define AKU_AMD_HAVE_HMMNETBW 1
#endif
#endif

namespace aku {

class Recipe {
public:
  class Info {
  public:
    std::string audio_path;
    std::string alt_audio_path;
    std::string transcript_path;
    std::string alignment_path;
    std::string hmmnet_path;
    std::string den_hmmnet_path;
    std::string lna_path;
    float start_time;
    float end_time;
    int start_line;
    int end_line;
    std::string speaker_id;
    std::string utterance_id;

    Info() : start_time(0), end_time(0), start_line(0), end_line(0) {}
#ifdef AKU_AMD_HAVE_PHNREADER
    /** aku/Recipe.cc:152-194: opens the audio (when fea_gen is given) and the transcript or
     * alignment PHN file, with the recipe line's frame and line limits */
    PhnReader *init_phn_files(HmmSet *model, bool relative_sample_nums, bool state_num_labels, bool out_phn,
                              FeatureGenerator *fea_gen, PhnReader *phn_reader) {
      float frame_rate = 125;
      if (fea_gen != NULL) fea_gen->open(audio_path);
      if (phn_reader == NULL) {
        if (model == NULL)
          throw std::string("recipe::Info::init_phn_files: HMM model is required if phn_reader==NULL");
        phn_reader = new PhnReader(model);
      }
      phn_reader->set_state_num_labels(state_num_labels);
      phn_reader->set_relative_sample_numbers(relative_sample_nums);
      if (fea_gen != NULL) frame_rate = fea_gen->frame_rate();
      phn_reader->set_frame_rate(frame_rate);
      phn_reader->open(out_phn ? alignment_path : transcript_path);
      if (start_time > 0 || end_time > 0)
        phn_reader->set_frame_limits((int)(start_time * frame_rate), (int)(end_time * frame_rate));
      if (start_line > 0 || end_line > 0) phn_reader->set_line_limits(start_line, end_line);
      return phn_reader;
    }
#endif
#ifdef AKU_AMD_HAVE_HMMNETBW
    /** aku/Recipe.cc:197-229: opens the audio and the (numerator or denominator) HMM network, with the
     * recipe line's frame limits */
    HmmNetBaumWelch *init_hmmnet_files(HmmSet *model, bool den_hmmnet, FeatureGenerator *fea_gen,
                                       HmmNetBaumWelch *hnbw) {
      fea_gen->open(audio_path);
      if (hnbw == NULL) {
        if (model == NULL)
          throw std::string("Recipe::Info::init_hmmnet_files: HMM model is required if hnbw==NULL");
        hnbw = new HmmNetBaumWelch(*fea_gen, *model);
      }
      if (den_hmmnet) {
        hnbw->open(den_hmmnet_path);
      } else {
        if (hmmnet_path.empty())
          throw std::string("Recipe::Info::init_hmmnet_files: hmmnet not specified in recipe.");
        hnbw->open(hmmnet_path);
      }
      if (start_time > 0 || end_time > 0) {
        const float frame_rate = fea_gen->frame_rate();
        hnbw->set_frame_limits((int)(start_time * frame_rate), (int)(end_time * frame_rate));
      }
      return hnbw;
    }
#endif
    bool operator<(const Info &i) const { return (speaker_id < i.speaker_id); }
  };

  void clear() { infos.clear(); }

  void read(FILE *f, int num_batches, int batch_index, bool cluster_speakers) {
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    char *table = NULL;
    int64_t len = 0;
    if (aasr_recipe_read_all(text.c_str(), num_batches, batch_index, cluster_speakers ? 1 : 0, &table, &len) != AASR_OK)
      throw std::string(aasr_last_error());
    const std::string t(table, (size_t)len);
    aasr_free(table);
    size_t pos = 0;
    while (pos < t.size()) {
      size_t eol = t.find('\n', pos);
      if (eol == std::string::npos) eol = t.size();
      std::vector<std::string> f13;
      size_t a = pos;
      while (a <= eol) {
        size_t b = t.find('\x1f', a);
        if (b == std::string::npos || b > eol) b = eol;
        f13.push_back(t.substr(a, b - a));
        a = b + 1;
      }
      if (f13.size() == 13) {
        Info i;
        i.audio_path = f13[0];
        i.alt_audio_path = f13[1];
        i.transcript_path = f13[2];
        i.alignment_path = f13[3];
        i.hmmnet_path = f13[4];
        i.den_hmmnet_path = f13[5];
        i.lna_path = f13[6];
        i.start_time = (float)atof(f13[7].c_str());   // "%.9g": exact for a float
        i.end_time = (float)atof(f13[8].c_str());
        i.start_line = atoi(f13[9].c_str());
        i.end_line = atoi(f13[10].c_str());
        i.speaker_id = f13[11];
        i.utterance_id = f13[12];
        infos.push_back(i);
      }
      pos = eol + 1;
    }
  }

  std::vector<Info> infos;
  void sort_infos() { std::stable_sort(infos.begin(), infos.end()); }
};

}  // namespace aku

#endif
