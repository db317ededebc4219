// aku_adapter_check -- the reference's own per-frame calling sequence
// (aku/phone_probs.cc:209-263: gen.open, generate(f), eof(), reset_cache,
// precompute_likelihoods, state_likelihood, float normalisation, safe_log,
// 2/4-byte packing) written against the adapter classes.  Used by the tests to
// show that code shaped like the reference's callers runs unchanged on the
// engine and produces the same LNA as the batched entry point.
//
//   aku_adapter_check CFG MODEL_BASE AUDIO OUT.lna LNABYTES [GCL MINC MING]
//                     [spkc SPKC SPEAKER UTTERANCE]
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "FeatureGenerator.hh"
#include "HmmSet.hh"
#include "SpeakerConfig.hh"

static double safe_log(double x) {  // aku/util.hh:132-139
  const double tiny_for_log = 1e-50;
  return x < tiny_for_log ? log(tiny_for_log) : log(x);
}

int main(int argc, char **argv) {
  if (argc != 6 && argc != 9 && argc != 10 && argc != 13) {
    fprintf(stderr, "usage: aku_adapter_check CFG MODEL_BASE AUDIO OUT.lna LNABYTES [GCL MINC MING] "
                    "[spkc SPKC SPEAKER UTTERANCE]\n");
    return 2;
  }
  try {
    aku::FeatureGenerator gen;
    aku::HmmSet model;
    FILE *cf = fopen(argv[1], "r");
    if (!cf) throw std::string("could not open config");
    gen.load_configuration(cf);
    fclose(cf);
    aku::SpeakerConfig speaker_conf(gen, &model);
    int arg = 6;
    const bool with_gcl = arg < argc && std::string(argv[arg]) != "spkc";
    const int spk_arg = with_gcl ? arg + 3 : arg;
    const bool with_spk = spk_arg < argc && std::string(argv[spk_arg]) == "spkc";
    if (with_spk) {  // aku/phone_probs.cc:94-95: before the model is read
      FILE *sf = fopen(argv[spk_arg + 1], "r");
      if (!sf) throw std::string("could not open speaker file");
      speaker_conf.read_speaker_file(sf);
      fclose(sf);
    }
    model.read_all(argv[2]);
    if (with_gcl) {  // aku/phone_probs.cc:112-117
      model.read_clustering(argv[arg]);
      model.set_clustering_min_evals(atof(argv[arg + 1]), atof(argv[arg + 2]));
    }
    if (with_spk) {  // aku/phone_probs.cc:191-196
      speaker_conf.set_speaker(argv[spk_arg + 2]);
      if (argv[spk_arg + 3][0]) speaker_conf.set_utterance(argv[spk_arg + 3]);
    }
    const int lnabytes = atoi(argv[5]);
    if (model.dim() != gen.dim()) throw std::string("dimension mismatch");
    gen.open(argv[3]);
    FILE *ofp = fopen(argv[4], "wb");
    if (!ofp) throw std::string("could not open output");
    unsigned int ns = model.num_states();
    unsigned char hdr[5] = {(unsigned char)(ns >> 24), (unsigned char)(ns >> 16),
                            (unsigned char)(ns >> 8), (unsigned char)ns, (unsigned char)lnabytes};
    fwrite(hdr, 1, 5, ofp);
    std::vector<float> obs_log_probs;
    unsigned char buffer[4];
    for (int f = 0; f < INT_MAX; f++) {
      const aku::FeatureVec fea_vec = gen.generate(f);
      if (gen.eof()) break;
      model.reset_cache();
      model.precompute_likelihoods(fea_vec);
      obs_log_probs.resize(model.num_states());
      double log_normalizer = 0;
      for (int i = 0; i < model.num_states(); i++) {
        obs_log_probs[i] = model.state_likelihood(i, fea_vec);
        log_normalizer += obs_log_probs[i];
      }
      if (log_normalizer == 0) log_normalizer = 1;
      for (int i = 0; i < (int)obs_log_probs.size(); i++)
        obs_log_probs[i] = safe_log(obs_log_probs[i] / log_normalizer);
      for (int i = 0; i < model.num_states(); i++) {
        if (lnabytes == 4) {
          unsigned char *p = (unsigned char *)&obs_log_probs[i];
          for (int j = 0; j < 4; j++) buffer[j] = p[j];
        } else {
          if (obs_log_probs[i] < -36.008) {
            buffer[0] = 255;
            buffer[1] = 255;
          } else {
            int temp = (int)(-1820.0 * obs_log_probs[i] + .5);
            buffer[0] = (unsigned char)((temp >> 8) & 255);
            buffer[1] = (unsigned char)(temp & 255);
          }
        }
        fwrite(buffer, 1, lnabytes, ofp);
      }
    }
    gen.close();
    fclose(ofp);
  } catch (std::exception &e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  } catch (std::string &str) {
    fprintf(stderr, "exception: %s\n", str.c_str());
    return 1;
  }
  return 0;
}
