// feacat -- the reference's feature dump tool (aku/feacat.cc) on the adapter
// classes: prints the features of one audio (or, with a `pre` base module,
// feature) file as text or raw float32, with the same flags:
//
//   feacat -c CFG [--raw-output] [-H] [-s START] [-e END]
//          [-S SPKC -d SPEAKER [-u UTTERANCE]] FILE|-
//
// FILE "-" reads standard input like the reference's io::Stream.  Text output is
// "%8.4f " per value (aku/feacat.cc:27-31); raw output is float32 with an
// optional int32 dimension header (-H), the format PreModule reads back.
// Not built: -w/--write-config, -G/--gaussian-std.
#include <getopt.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "FeatureGenerator.hh"
#include "SpeakerConfig.hh"

static void die(const std::string &msg) {
  fprintf(stderr, "exception: %s\n", msg.c_str());
  exit(1);
}

int main(int argc, char *argv[]) {
  std::string cfg, speakers, speaker_id, utterance_id;
  bool raw_output = false, header = false, utt_set = false;
  int start_frame = 0, end_frame = INT_MAX;
  static struct option opts[] = {
      {"help", no_argument, 0, 'h'},           {"config", required_argument, 0, 'c'},
      {"write-config", required_argument, 0, 'w'}, {"raw-output", no_argument, 0, 1},
      {"header", no_argument, 0, 'H'},         {"start-frame", required_argument, 0, 's'},
      {"end-frame", required_argument, 0, 'e'}, {"speakers", required_argument, 0, 'S'},
      {"speaker-id", required_argument, 0, 'd'}, {"utterance-id", required_argument, 0, 'u'},
      {"gaussian-std", required_argument, 0, 'G'}, {0, 0, 0, 0}};
  int c;
  while ((c = getopt_long(argc, argv, "hc:w:Hs:e:S:d:u:G:", opts, nullptr)) != -1) {
    switch (c) {
      case 'h':
        printf("usage: feacat [OPTION...] FILE\n  -c CFG  feature configuration\n  --raw-output  raw float output\n"
               "  -H  write a header (feature dim, 32 bits) in raw output\n  -s INT  start frame\n"
               "  -e INT  end frame\n  -S FILE  speaker configuration\n  -d NAME  speaker ID\n"
               "  -u NAME  utterance ID\n");
        return 0;
      case 'c': cfg = optarg; break;
      case 1: raw_output = true; break;
      case 'H': header = true; break;
      case 's': start_frame = atoi(optarg); break;
      case 'e': end_frame = atoi(optarg); break;
      case 'S': speakers = optarg; break;
      case 'd': speaker_id = optarg; break;
      case 'u': utterance_id = optarg; utt_set = true; break;
      case 'w': die("--write-config is not built in this engine yet");
      case 'G': die("--gaussian-std is not built in this engine yet");
      default: return 2;
    }
  }
  if (cfg.empty()) die("option --config is required");
  if (argc - optind != 1) die("usage: feacat [OPTION...] FILE");
  if (header && !raw_output) fprintf(stderr, "Warning: header is only written in raw output mode\n");
  try {
    aku::FeatureGenerator gen;
    aku::SpeakerConfig speaker_conf(gen);
    FILE *cf = fopen(cfg.c_str(), "r");
    if (!cf) throw std::string("could not open ") + cfg;
    gen.load_configuration(cf);
    fclose(cf);
    std::string in = argv[optind];
    if (in == "-") gen.open(stdin, true);
    else gen.open(in);
    if (!speakers.empty()) {
      FILE *sf = fopen(speakers.c_str(), "r");
      if (!sf) throw std::string("could not open ") + speakers;
      speaker_conf.read_speaker_file(sf);
      fclose(sf);
      speaker_conf.set_speaker(speaker_id);
      if (utt_set) speaker_conf.set_utterance(utterance_id);
    }
    if (raw_output && header) {
      int dim = gen.dim();
      fwrite(&dim, sizeof(int), 1, stdout);
    }
    auto print_feature = [&](const aku::FeatureVec &fea) {
      if (raw_output) {
        for (int i = 0; i < fea.dim(); i++) {
          float tmp = fea[i];
          fwrite(&tmp, sizeof(float), 1, stdout);
        }
      } else {
        for (int i = 0; i < fea.dim(); i++) printf("%8.4f ", fea[i]);
        printf("\n");
      }
    };
    if (start_frame < end_frame) {
      if (end_frame != INT_MAX) gen.set_block_frames(end_frame - start_frame + 1);
      for (int f = start_frame; f <= end_frame; f++) {
        const aku::FeatureVec fea = gen.generate(f);
        if (end_frame == INT_MAX && gen.eof()) break;
        print_feature(fea);
      }
    } else {
      // one device block covers the whole (descending) range
      gen.set_block_frames(start_frame - end_frame + 1);
      gen.generate(end_frame);
      for (int f = start_frame; f >= end_frame; f--) print_feature(gen.generate(f));
    }
    gen.close();
  } catch (std::string &str) {
    die(str);
  }
  return 0;
}
