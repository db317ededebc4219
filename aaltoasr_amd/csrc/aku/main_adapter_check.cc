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
#include "FeatureModules.hh"
#include "HmmSet.hh"
#include "SpeakerConfig.hh"

static double safe_log(double x) {  // aku/util.hh:132-139
  const double tiny_for_log = 1e-50;
  return x < tiny_for_log ? log(tiny_for_log) : log(x);
}

// aku_adapter_check modules CFG AUDIO OUT: the FeatureModule view of every module of the graph
// (FeatureGenerator::module(name): name, type_str, dim, at(frame) for frames -3 .. 20), then the
// get_parameters -> set_parameters round trip on every module that has parameters and the
// generator's output after it.  OUT is text: one line per value group.
static int modules_mode(const char *cfg, const char *audio, const char *out_path) {
  aku::FeatureGenerator gen;
  FILE *cf = fopen(cfg, "r");
  if (!cf) throw std::string("could not open config");
  gen.load_configuration(cf);
  fclose(cf);
  gen.open(audio);
  FILE *out = fopen(out_path, "w");
  if (!out) throw std::string("could not open output");
  FILE *wf = tmpfile();
  gen.write_configuration(wf);
  rewind(wf);
  // module names in configuration order, from the written configuration ("  name X" lines)
  std::vector<std::string> names;
  char line[4096];
  while (fgets(line, sizeof line, wf)) {
    std::string l(line);
    if (l.compare(0, 7, "  name ") == 0) names.push_back(l.substr(7, l.size() - 8));
  }
  fclose(wf);
  for (const std::string &nm : names) {
    aku::FeatureModule *m = gen.module(nm);
    fprintf(out, "module %s %s %d\n", m->name().c_str(), m->type_str().c_str(), m->dim());
    for (int f = -3; f <= 20; f++) {
      const aku::FeatureVec v = m->at(f);
      for (int i = 0; i < v.dim(); i++) fprintf(out, "%.17g ", v[i]);
      fprintf(out, "\n");
    }
  }
  bool unknown_thrown = false;
  try {
    gen.module("no-such-module");
  } catch (std::string &e) {
    unknown_thrown = e == "unknown module requested: no-such-module";
  }
  fprintf(out, "unknown %d\n", unknown_thrown ? 1 : 0);
  for (const std::string &nm : names) {
    aku::FeatureModule *m = gen.module(nm);
    aku::ModuleConfig c;
    m->get_parameters(c);
    std::vector<float> mean;
    if (c.get("mean", mean)) {  // a normalization module: shift every mean, read back
      for (float &x : mean) x += 0.25f;
      c.set("mean", mean);
      m->set_parameters(c);
      aku::ModuleConfig back;
      m->get_parameters(back);
      fprintf(out, "params %s %s", nm.c_str(), back.text().c_str());
    }
  }
  for (int f = 0; f <= 5; f++) {
    const aku::FeatureVec v = gen.generate(f);
    for (int i = 0; i < v.dim(); i++) fprintf(out, "%.17g ", v[i]);
    fprintf(out, "\n");
  }
  fclose(out);
  return 0;
}

// aku_adapter_check pool CFG MODEL_BASE AUDIO OUT: per-Gaussian log-likelihoods (PDFPool view of
// HmmSet) for two frames, one value per line.
static int pool_mode(const char *cfg, const char *base, const char *audio, const char *out_path) {
  aku::FeatureGenerator gen;
  aku::HmmSet model;
  FILE *cf = fopen(cfg, "r");
  if (!cf) throw std::string("could not open config");
  gen.load_configuration(cf);
  fclose(cf);
  model.read_all(base);
  gen.open(audio);
  FILE *out = fopen(out_path, "w");
  if (!out) throw std::string("could not open output");
  fprintf(out, "%d\n", model.num_pool_pdfs());
  for (int f : {0, 7}) {
    const aku::FeatureVec v = gen.generate(f);
    for (int g = model.num_pool_pdfs() - 1; g >= 0; g--)  // any order, one device pass per frame
      fprintf(out, "%.9g %.9g\n", model.pool_log_likelihood(g, v), model.pool_likelihood(g, v));
  }
  fclose(out);
  return 0;
}

// The Distributions surface as the reference's training tools use it: for a frame f
//   model.get_emission_pdf(s)->compute_likelihood(*f.get_vector())     (aku/logl.cc:58-60 through
//                                                                         HmmSet::state_likelihood)
//   mixture->get_base_pdf(k)->compute_likelihood(*f.get_vector()), get_base_pdf_index, size,
//   get_mixture_coefficient, Gaussian::get_mean / get_covariance        (aku/MllrTrainer.cc:40-56)
//   model.get_pool()->compute_likelihood(*f.get_vector(), g)
// plus the same calls on a Vector that is nobody's block (a copy).  One line per value.
static int dist_mode(const char *cfg, const char *base, const char *audio, const char *out_path) {
  aku::FeatureGenerator gen;
  aku::HmmSet model;
  FILE *cf = fopen(cfg, "r");
  if (!cf) throw std::string("could not open config");
  gen.load_configuration(cf);
  fclose(cf);
  model.read_all(base);
  gen.open(audio);
  FILE *out = fopen(out_path, "w");
  if (!out) throw std::string("could not open output");
  fprintf(out, "%d %d\n", model.num_states(), model.get_pool()->size());
  for (int f : {3, 11}) {
    aku::FeatureVec fea = gen.generate(f);
    model.reset_cache();  // the caller's duty whenever the frame changes (aku/HmmSet.cc:444-457)
    for (int s = 0; s < model.num_states(); s++) {
      aku::Mixture *mixture = model.get_emission_pdf(model.emission_pdf_index(s));
      double by_parts = 0;
      for (int k = 0; k < mixture->size(); k++) {
        aku::Gaussian *gaussian = dynamic_cast<aku::Gaussian *>(mixture->get_base_pdf(k));
        by_parts += mixture->get_mixture_coefficient(k) * gaussian->compute_likelihood(*fea.get_vector());
      }
      const aku::Vector own = *fea.get_vector();  // a copy: not a frame of any block
      fprintf(out, "%.9g %.9g %.9g %.9g %.9g\n", mixture->compute_likelihood(*fea.get_vector()),
              mixture->compute_log_likelihood(*fea.get_vector()), model.state_likelihood(s, fea), by_parts,
              mixture->compute_likelihood(own));
    }
    aku::Mixture *m0 = model.get_emission_pdf(0);
    aku::Vector mean, covar;
    dynamic_cast<aku::Gaussian *>(m0->get_base_pdf(0))->get_mean(mean);
    dynamic_cast<aku::Gaussian *>(m0->get_base_pdf(0))->get_covariance(covar);
    fprintf(out, "%d %.17g %.17g %.9g\n", m0->get_base_pdf_index(0), mean(0), covar(covar.size() - 1),
            model.get_pool()->compute_likelihood(*fea.get_vector(), m0->get_base_pdf_index(0)));
  }
  fclose(out);
  return 0;
}

// random [seed steps block]: the per-frame API driven out of order -- frames forwards, backwards and far apart,
// before 0 and past the end, the eager style (precompute_likelihoods, every state) and the lazy one
// (reset_cache, a few states), a module tap in between -- with a small adapter block so that almost every
// call lands on a block edge.  One text line per value; tests compare them with the batch entry points.
static int random_mode(const char *cfg, const char *base, const char *audio, const char *out_path, unsigned seed,
                       int steps, int block) {
  aku::FeatureGenerator gen;
  aku::HmmSet model;
  FILE *cf = fopen(cfg, "r");
  if (!cf) throw std::string("could not open config");
  gen.load_configuration(cf);
  fclose(cf);
  model.read_all(base);
  gen.set_block_frames(block);
  gen.open(audio);
  FILE *out = fopen(out_path, "w");
  if (!out) throw std::string("could not open output");
  unsigned long long x = seed * 2654435761ull + 12345;
  auto rnd = [&](int n) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    return (int)((x >> 33) % (unsigned long long)n);
  };
  const int S = model.num_states();
  int f = 0, last = 400;
  for (int i = 0; i < steps; i++) {
    switch (rnd(6)) {
      case 0: f += 1; break;
      case 1: f -= 1 + rnd(3); break;
      case 2: f = rnd(last + 12) - 6; break;
      case 3: f += block - 1 + rnd(3); break;
      case 4: f = last - rnd(8) + 3; break;
      default: break;  // the same frame again
    }
    if (f < -8) f = -8;
    const aku::FeatureVec fea = gen.generate(f);
    const bool eof = gen.eof();
    fprintf(out, "F %d %d", f, eof ? 1 : 0);
    for (int d = 0; d < fea.dim(); d++) fprintf(out, " %a", fea[d]);
    fprintf(out, "\n");
    if (eof && f < last) last = f;
    const int style = rnd(3);
    if (style == 0) {
      model.reset_cache();
      model.precompute_likelihoods(fea);
      fprintf(out, "E %d", f);
      for (int s = 0; s < S; s++) fprintf(out, " %a", model.state_likelihood(s, fea));
      fprintf(out, "\n");
    } else if (style == 1) {
      model.reset_cache();
      for (int k = 0; k < 3; k++) {
        const int s = rnd(S);
        fprintf(out, "L %d %d %a\n", f, s, model.state_likelihood(s, fea));
      }
    } else {
      const aku::FeatureVec tap = gen.module("mfcc")->at(f);
      fprintf(out, "T %d", f);
      for (int d = 0; d < tap.dim(); d++) fprintf(out, " %a", tap[d]);
      fprintf(out, "\n");
    }
  }
  fclose(out);
  gen.close();
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 5 && std::string(argv[1]) == "setters") {
    // aku_adapter_check setters CFG AUDIO OUT: the module classes the estimation tools reach through
    // dynamic_cast (aku/feanorm.cc:72-101, 264, 380; aku/vtln.cc:208): NormalizationModule::
    // set_normalization, LinTransformModule::set_transformation_matrix / _bias / get_*, VtlnModule::
    // set_warp_factor / get_warp_factor.  The values are fixed functions of the index so that the
    // test can apply the same ones to the oracle; OUT = 8 output frames as "%.17g".
    aku::FeatureGenerator gen;
    FILE *cf = fopen(argv[2], "r");
    if (!cf) throw std::string("could not open config");
    gen.load_configuration(cf);
    fclose(cf);
    gen.open(argv[3]);
    aku::NormalizationModule *norm = dynamic_cast<aku::NormalizationModule *>(gen.module("norm"));
    aku::LinTransformModule *lin = dynamic_cast<aku::LinTransformModule *>(gen.module("mllr"));
    aku::VtlnModule *vtln = dynamic_cast<aku::VtlnModule *>(gen.module("vtln"));
    if (!norm || !lin || !vtln) throw std::string("module classes missing");
    if (dynamic_cast<aku::VtlnModule *>(gen.module("norm"))) throw std::string("wrong class");
    const int d = norm->dim();
    std::vector<float> mean(d), scale(d), mat((size_t)d * d), bias(d);
    for (int i = 0; i < d; i++) {
      mean[i] = 0.1f * (float)i - 0.7f;
      scale[i] = 1.0f / (1.0f + 0.03f * (float)i);
      bias[i] = 0.01f * (float)(i % 5) - 0.02f;
      for (int j = 0; j < d; j++) mat[(size_t)i * d + j] = (i == j ? 1.0f : 0.0f) + 0.001f * (float)((i * 7 + j * 3) % 11);
    }
    norm->set_normalization(mean, scale);
    lin->set_transformation_matrix(mat);
    lin->set_transformation_bias(bias);
    vtln->set_warp_factor(1.0f + 0.03f);
    bool dim_thrown = false;
    try {
      std::vector<float> bad(3);
      norm->set_normalization(bad, bad);
    } catch (std::string &) {
      dim_thrown = true;
    }
    FILE *out = fopen(argv[4], "w");
    if (!out) throw std::string("could not open output");
    fprintf(out, "%d %.9g %d %d\n", dim_thrown ? 1 : 0, (double)vtln->get_warp_factor(),
            (int)lin->get_transformation_matrix()->size(), (int)lin->get_transformation_bias()->size());
    for (int f = 0; f < 8; f++) {
      const aku::FeatureVec v = gen.generate(f);
      for (int i = 0; i < v.dim(); i++) fprintf(out, "%.17g ", v[i]);
      fprintf(out, "\n");
    }
    // an empty matrix puts the identity back (aku/FeatureModules.cc:1276-1285)
    std::vector<float> none;
    lin->set_transformation_matrix(none);
    lin->set_transformation_bias(none);
    for (int f = 0; f < 2; f++) {
      const aku::FeatureVec v = gen.generate(f);
      for (int i = 0; i < v.dim(); i++) fprintf(out, "%.17g ", v[i]);
      fprintf(out, "\n");
    }
    fclose(out);
    return 0;
  }
  if (argc == 4 && std::string(argv[1]) == "stream") {
    // aku_adapter_check stream CFG OUT < raw PCM16: decode-stream.cc's reading loop
    // (decoder/decode-stream.cc:81, 238-276: gen.open(stdin, true, true); generate(f) until eof()).
    // Each frame goes to OUT as doubles and "frame N" to stdout, both flushed at once, so the test
    // can see frames arrive while it is still feeding the pipe.
    aku::FeatureGenerator gen;
    FILE *cf = fopen(argv[2], "r");
    if (!cf) throw std::string("could not open config");
    gen.load_configuration(cf);
    fclose(cf);
    gen.open(stdin, true, true);
    FILE *out = fopen(argv[3], "wb");
    if (!out) throw std::string("could not open output");
    for (int f = 0;; f++) {
      const aku::FeatureVec v = gen.generate(f);
      if (gen.eof()) break;
      for (int i = 0; i < v.dim(); i++) {
        const double x = v[i];
        fwrite(&x, sizeof x, 1, out);
      }
      fflush(out);
      printf("frame %d\n", f);
      fflush(stdout);
    }
    fclose(out);
    gen.close();
    return 0;
  }
  if (argc == 9 && std::string(argv[1]) == "random") {
    try {
      return random_mode(argv[2], argv[3], argv[4], argv[5], (unsigned)atoi(argv[6]), atoi(argv[7]), atoi(argv[8]));
    } catch (std::string &e) {
      fprintf(stderr, "exception: %s\n", e.c_str());
      return 1;
    }
  }
  if (argc == 6 && std::string(argv[1]) == "dist") {
    try {
      return dist_mode(argv[2], argv[3], argv[4], argv[5]);
    } catch (std::string &e) {
      fprintf(stderr, "exception: %s\n", e.c_str());
      return 1;
    }
  }
  if (argc == 6 && std::string(argv[1]) == "pool") {
    try {
      return pool_mode(argv[2], argv[3], argv[4], argv[5]);
    } catch (std::string &e) {
      fprintf(stderr, "exception: %s\n", e.c_str());
      return 1;
    }
  }
  if (argc == 5 && std::string(argv[1]) == "modules") {
    try {
      return modules_mode(argv[2], argv[3], argv[4]);
    } catch (std::string &e) {
      fprintf(stderr, "exception: %s\n", e.c_str());
      return 1;
    }
  }
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
