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
// -G/--gaussian-std adds N(0, std) noise to every value (aku/feacat.cc:38-42; the generator's
// stream is this engine's own, see ziggurat.hh).
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "FeatureGenerator.hh"
#include "conf.hh"
#include "SpeakerConfig.hh"
#include "ziggurat.hh"

static void die(const std::string &msg) {
  fprintf(stderr, "exception: %s\n", msg.c_str());
  exit(1);
}

int main(int argc, char *argv[]) {
  // option table of aku/feacat.cc:50-63, grammar of conf.hh
  aku::conf::Config config;
  config("usage: feacat [OPTION...] FILE\n")
    ('h', "help", "", "", "display help")
    ('c', "config=FILE", "arg must", "", "read feature configuration")
    ('w', "write-config=FILE", "arg", "", "write feature configuration")
    ('\0', "raw-output", "", "", "raw float output")
    ('H', "header", "", "", "write a header (feature dim, 32 bits) in raw output")
    ('s', "start-frame=INT", "arg", "", "audio start frame")
    ('e', "end-frame=INT", "arg", "", "audio end frame")
    ('S', "speakers=FILE", "arg", "", "speaker configuration file")
    ('d', "speaker-id=NAME", "arg", "", "speaker ID")
    ('u', "utterance-id=NAME", "arg", "", "utterance ID")
    ('G', "gaussian-std=FLOAT", "arg", "", "Gaussian noise std added to features");
  config.default_parse(argc, argv);
  if (config.arguments.size() != 1) config.print_help(stderr, 1);
  const double noise_std = config["gaussian-std"].specified ? config["gaussian-std"].get_double() : 0.0;
  const bool raw_output = config["raw-output"].specified, header = config["header"].specified;
  const std::string cfg = config["config"].get_str();
  const std::string speakers = config["speakers"].specified ? config["speakers"].get_str() : "";
  const std::string speaker_id = config["speaker-id"].get_str(), utterance_id = config["utterance-id"].get_str();
  const bool utt_set = config["utterance-id"].specified;
  const int start_frame = config["start-frame"].specified ? config["start-frame"].get_int() : 0;
  const int end_frame = config["end-frame"].specified ? config["end-frame"].get_int() : INT_MAX;
  if (header && !raw_output) fprintf(stderr, "Warning: header is only written in raw output mode\n");
  try {
    aku::FeatureGenerator gen;
    aku::SpeakerConfig speaker_conf(gen);
    FILE *cf = fopen(cfg.c_str(), "r");
    if (!cf) throw std::string("could not open ") + cfg;
    gen.load_configuration(cf);
    fclose(cf);
    const std::string in = config.arguments[0];
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
    if (config["write-config"].specified) {
      // after the speaker settings, like the reference (aku/feacat.cc:85-87)
      FILE *wf = fopen(config["write-config"].get_c_str(), "w");
      if (!wf) throw std::string("could not open ") + config["write-config"].get_str();
      gen.write_configuration(wf);
      fclose(wf);
    }
    if (raw_output && header) {
      int dim = gen.dim();
      fwrite(&dim, sizeof(int), 1, stdout);
    }
    auto print_feature = [&](aku::FeatureVec fea) {
      if (noise_std > 0.0)  // on the generator's block, like the reference writes into its ring buffer
        for (int i = 0; i < fea.dim(); i++) fea[i] += ziggurat::rnd.rnor() * noise_std;
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
