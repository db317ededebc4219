// phone_probs -- command-line front end with the reference tool's flags
// (aku/phone_probs.cc:60-81) driving the MI355X engine through the C ABI.
//
//   phone_probs (-b BASE | -g GK -m MC -p PH) -c CFG -r RECIPE [-o DIR]
//               [--lnabytes 2|4] [-a] [-n] [-N] [-B n -I k] [-i level]
//               [-C GCL --eval-minc R --eval-ming R] [-S SPKC]
//               [--sort-recipe] [--model-cache FILE]
// One process drives one GPU (--device N or
// HIP_VISIBLE_DEVICES); run N processes with -B N -I k for N GPUs, exactly as
// the reference scales over CPU cores.
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "../../../include/aasr.h"
#include "conf.hh"

static void die(const std::string &msg) {
  fprintf(stderr, "exception: %s\n", msg.c_str());
  exit(1);
}

int main(int argc, char *argv[]) {
  // the reference's option table (aku/phone_probs.cc:60-81) and grammar (conf.hh), plus the two
  // options this engine adds
  aku::conf::Config config;
  config("usage: phone_probs [OPTION...]\n")
    ('h', "help", "", "", "display help")
    ('b', "base=BASENAME", "arg", "", "base filename for model files")
    ('g', "gk=FILE", "arg", "", "Gaussian kernels")
    ('m', "mc=FILE", "arg", "", "kernel indices for states")
    ('p', "ph=FILE", "arg", "", "HMM definitions")
    ('c', "config=FILE", "arg must", "", "feature configuration")
    ('r', "recipe=FILE", "arg must", "", "recipe file")
    ('o', "output-dir=DIR", "arg", "", "output directory (default: use filenames from recipe)")
    ('\0', "lnabytes=INT", "arg", "2", "number of bytes for probabilities, 2 (default) or 4")
    ('a', "afname", "", "", "use audio file name")
    ('n', "no-overwrite", "", "", "prevent overwriting existing files")
    ('S', "speakers=FILE", "arg", "", "speaker configuration file")
    ('C', "clusters=FILE", "arg", "", "Gaussian clustering file")
    ('\0', "eval-minc=FLOAT", "arg", "0", "minimum ratio of top clusters to evaluate")
    ('\0', "eval-ming=FLOAT", "arg", "0.1", "minimum ratio of Gaussians to evaluate")
    ('\0', "sort-recipe", "", "", "sort recipe lines, useful with adaptation")
    ('N', "no-normalization", "", "", "do not normalize the likelihoods")
    ('B', "batch=INT", "arg", "0", "number of batch processes with the same recipe")
    ('I', "bindex=INT", "arg", "0", "batch process index")
    ('i', "info=INT", "arg", "0", "info level")
    ('\0', "device=INT", "arg", "-1", "GPU ordinal (default: the first visible device)")
    ('\0', "model-cache=FILE", "arg", "", "binary model image, written on first use");
  config.default_parse(argc, argv);

  const int info = config["info"].get_int();
  const std::string cfg = config["config"].get_str(), recipe = config["recipe"].get_str();
  const int lnabytes = config["lnabytes"].get_int();
  if (lnabytes != 2 && lnabytes != 4) die("Invalid number of LNA bytes");
  const bool no_overwrite = config["no-overwrite"].specified, afname = config["afname"].specified;
  const bool no_norm = config["no-normalization"].specified, sort_recipe = config["sort-recipe"].specified;
  const std::string speakers = config["speakers"].specified ? config["speakers"].get_str() : "";
  const std::string clusters = config["clusters"].specified ? config["clusters"].get_str() : "";
  const std::string model_cache = config["model-cache"].get_str();
  std::string gk, mc, ph, out_dir;
  if (config["base"].specified) {
    const std::string base = config["base"].get_str();
    gk = base + ".gk";
    mc = base + ".mc";
    ph = base + ".ph";
  } else if (config["gk"].specified && config["mc"].specified && config["ph"].specified) {
    gk = config["gk"].get_str();
    mc = config["mc"].get_str();
    ph = config["ph"].get_str();
  } else {
    die("Must give either --base or all --gk, --mc and --ph");
  }
  const double eval_minc = config["eval-minc"].get_double(), eval_ming = config["eval-ming"].get_double();
  if (config["output-dir"].specified) out_dir = config["output-dir"].get_str();
  if (config["batch"].specified ^ config["bindex"].specified) die("Must give both --batch and --bindex");
  const int batch = config["batch"].get_int(), bindex = config["bindex"].get_int();
  const int device = config["device"].get_int();
  if (device >= 0 && aasr_set_device(device) != AASR_OK) die(aasr_last_error());

  std::ifstream cin_(cfg);
  if (!cin_) die("could not open " + cfg);
  std::stringstream ss;
  ss << cin_.rdbuf();
  aasr_feat *feat = nullptr;
  aasr_gmm *gmm = nullptr;
  if (aasr_feat_create(ss.str().c_str(), &feat) != AASR_OK) die(aasr_last_error());
  // --model-cache FILE (new): binary image of the parsed model; created on the first run
  // and whenever it does not match the .gk/.mc/.ph named on the command line
  if (!model_cache.empty() &&
      aasr_gmm_create_from_cache_checked(model_cache.c_str(), gk.c_str(), mc.c_str(), ph.c_str(), &gmm) == AASR_OK) {
    if (info > 0) printf("Model read from cache %s\n", model_cache.c_str());
  } else {
    if (!model_cache.empty() && access(model_cache.c_str(), F_OK) == 0)
      fprintf(stderr, "WARNING: not using the model cache: %s\n", aasr_last_error());
    if (aasr_gmm_create_from_files(gk.c_str(), mc.c_str(), ph.c_str(), &gmm) != AASR_OK)
      die(aasr_last_error());
    if (!model_cache.empty() && aasr_gmm_write_cache(gmm, model_cache.c_str()) != AASR_OK)
      fprintf(stderr, "WARNING: could not write model cache: %s\n", aasr_last_error());
  }
  if (!clusters.empty()) {
    // aku/phone_probs.cc:112-117
    if (aasr_gmm_read_clustering(gmm, clusters.c_str()) != AASR_OK) die(aasr_last_error());
    if (aasr_gmm_set_clustering_min_evals(gmm, eval_minc, eval_ming) != AASR_OK)
      die(aasr_last_error());
  }
  aasr_spkc *spk = nullptr;
  if (!speakers.empty()) {
    if (aasr_spkc_create(feat, gmm, &spk) != AASR_OK) die(aasr_last_error());
    if (aasr_spkc_read_file(spk, speakers.c_str()) != AASR_OK) die(aasr_last_error());
  }
  aasr_run_options opt;
  memset(&opt, 0, sizeof opt);
  opt.lnabytes = lnabytes;
  opt.normalize = no_norm ? 0 : 1;
  opt.num_batches = batch;
  opt.batch_index = bindex;
  opt.no_overwrite = no_overwrite;
  opt.info = info;
  opt.afname = afname;
  opt.out_dir = out_dir.empty() ? nullptr : out_dir.c_str();
  opt.speakers = spk;
  opt.sort_recipe = sort_recipe;
  aasr_run_stats st;
  memset(&st, 0, sizeof st);
  if (aasr_run_recipe(feat, gmm, recipe.c_str(), &opt, &st) != AASR_OK) die(aasr_last_error());
  if (info > 0)
    fprintf(stderr, "{\"utterances\": %ld, \"frames\": %ld, \"seconds\": %.3f, \"device_seconds\": %.3f, \"frames_per_s\": %.1f}\n",
            (long)st.utterances, (long)st.frames, st.seconds_total, st.seconds_device,
            st.seconds_total > 0 ? st.frames / st.seconds_total : 0.0);
  aasr_spkc_destroy(spk);
  aasr_gmm_destroy(gmm);
  aasr_feat_destroy(feat);
  return 0;
}
