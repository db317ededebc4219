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
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "../../../include/aasr.h"

static void die(const std::string &msg) {
  fprintf(stderr, "exception: %s\n", msg.c_str());
  exit(1);
}

int main(int argc, char *argv[]) {
  std::string base, gk, mc, ph, cfg, recipe, out_dir;
  std::string clusters, speakers, model_cache;
  double eval_minc = 0.0, eval_ming = 0.1;  // defaults of aku/phone_probs.cc:74-75
  int lnabytes = 2, info = 0, batch = 0, bindex = 0, device = -1;
  bool sort_recipe = false;
  bool afname = false, no_overwrite = false, no_norm = false, batch_set = false, bindex_set = false;
  static struct option opts[] = {
      {"help", no_argument, 0, 'h'},          {"base", required_argument, 0, 'b'},
      {"gk", required_argument, 0, 'g'},      {"mc", required_argument, 0, 'm'},
      {"ph", required_argument, 0, 'p'},      {"config", required_argument, 0, 'c'},
      {"recipe", required_argument, 0, 'r'},  {"output-dir", required_argument, 0, 'o'},
      {"lnabytes", required_argument, 0, 1},  {"afname", no_argument, 0, 'a'},
      {"no-overwrite", no_argument, 0, 'n'},  {"speakers", required_argument, 0, 'S'},
      {"clusters", required_argument, 0, 'C'}, {"eval-minc", required_argument, 0, 2},
      {"eval-ming", required_argument, 0, 3}, {"sort-recipe", no_argument, 0, 4},
      {"no-normalization", no_argument, 0, 'N'}, {"batch", required_argument, 0, 'B'},
      {"bindex", required_argument, 0, 'I'},  {"info", required_argument, 0, 'i'},
      {"device", required_argument, 0, 5},    {"model-cache", required_argument, 0, 6},
      {0, 0, 0, 0}};
  int c;
  while ((c = getopt_long(argc, argv, "hb:g:m:p:c:r:o:anS:C:NB:I:i:", opts, nullptr)) != -1) {
    switch (c) {
      case 'h':
        printf("usage: phone_probs [OPTION...]\n"
               "  -b BASE | -g GK -m MC -p PH   model files\n  -c CFG   feature configuration\n"
               "  -r RECIPE  recipe file\n  -o DIR   output directory\n  --lnabytes=2|4\n"
               "  -a  use audio file name\n  -n  no overwrite\n  -N  no normalization\n"
               "  -B n -I k  batch k of n\n  -i level  info\n  --device=N  GPU ordinal\n"
               "  --model-cache=FILE  binary model image (written on first use)\n  -S SPKC  speaker configuration file\n  -C GCL  Gaussian clustering file\n  --eval-minc=R  minimum ratio of top clusters\n"
               "  --eval-ming=R  minimum ratio of Gaussians to evaluate\n");
        return 0;
      case 'b': base = optarg; break;
      case 'g': gk = optarg; break;
      case 'm': mc = optarg; break;
      case 'p': ph = optarg; break;
      case 'c': cfg = optarg; break;
      case 'r': recipe = optarg; break;
      case 'o': out_dir = optarg; break;
      case 1: lnabytes = atoi(optarg); break;
      case 'a': afname = true; break;
      case 'n': no_overwrite = true; break;
      case 'N': no_norm = true; break;
      case 'B': batch = atoi(optarg); batch_set = true; break;
      case 'I': bindex = atoi(optarg); bindex_set = true; break;
      case 'i': info = atoi(optarg); break;
      case 5: device = atoi(optarg); break;
      case 6: model_cache = optarg; break;
      case 'S': speakers = optarg; break;
      case 'C': clusters = optarg; break;
      case 2: eval_minc = atof(optarg); break;
      case 3: eval_ming = atof(optarg); break;
      case 4: sort_recipe = true; break;
      default: return 2;
    }
  }
  if (cfg.empty() || recipe.empty()) die("options --config and --recipe are required");
  if (lnabytes != 2 && lnabytes != 4) die("Invalid number of LNA bytes");
  if (!base.empty()) {
    gk = base + ".gk";
    mc = base + ".mc";
    ph = base + ".ph";
  } else if (gk.empty() || mc.empty() || ph.empty()) {
    die("Must give either --base or all --gk, --mc and --ph");
  }
  if (batch_set != bindex_set) die("Must give both --batch and --bindex");
  if (device >= 0 && aasr_set_device(device) != AASR_OK) die(aasr_last_error());

  std::ifstream cin_(cfg);
  if (!cin_) die("could not open " + cfg);
  std::stringstream ss;
  ss << cin_.rdbuf();
  aasr_feat *feat = nullptr;
  aasr_gmm *gmm = nullptr;
  if (aasr_feat_create(ss.str().c_str(), &feat) != AASR_OK) die(aasr_last_error());
  // --model-cache FILE (new): binary image of the parsed model; created on the first run
  if (!model_cache.empty() && aasr_gmm_create_from_cache(model_cache.c_str(), &gmm) == AASR_OK) {
    if (info > 0) printf("Model read from cache %s\n", model_cache.c_str());
  } else {
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
