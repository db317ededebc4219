// acoustics_check -- walks an EngineAcoustics object the way the decoder does
// (go_to(frame) then log_prob(model) for every model; decoder/src/TokenPassSearch
// reads m_acoustics->log_prob(...) after go_to) and dumps the floats, so the tests
// can compare them with LnaReaderCircular's view of the LNA file.
//
//   acoustics_check CFG MODEL_BASE AUDIO LNABYTES BLOCK_FRAMES OUT.f32
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

#include "EngineAcoustics.hh"

int main(int argc, char **argv) {
  if (argc != 7) {
    fprintf(stderr, "usage: acoustics_check CFG MODEL_BASE AUDIO LNABYTES BLOCK_FRAMES OUT.f32\n");
    return 2;
  }
  try {
    std::ifstream cin_(argv[1]);
    std::stringstream ss;
    ss << cin_.rdbuf();
    aasr_feat *feat = nullptr;
    aasr_gmm *gmm = nullptr;
    if (aasr_feat_create(ss.str().c_str(), &feat) != AASR_OK) throw std::string(aasr_last_error());
    const std::string base = argv[2];
    if (aasr_gmm_create_from_files((base + ".gk").c_str(), (base + ".mc").c_str(), (base + ".ph").c_str(),
                                   &gmm) != AASR_OK)
      throw std::string(aasr_last_error());
    EngineAcoustics ac(feat, gmm, atoi(argv[4]), true, atoi(argv[5]));
    ac.open_file(argv[3]);
    Acoustics &a = ac;  // through the interface the decoder sees
    FILE *out = fopen(argv[6], "wb");
    if (!out) throw std::string("could not open output");
    int frames = 0;
    // out-of-order access first: the decoder may step back within its window
    if (a.go_to(7) && a.go_to(3)) {
    }
    for (int f = 0; a.go_to(f); f++, frames++)
      for (int m = 0; m < a.num_models(); m++) {
        float v = a.log_prob(m);
        fwrite(&v, 4, 1, out);
      }
    fclose(out);
    printf("%d frames, %d models, eof %d\n", frames, a.num_models(), ac.eof_frame());
    aasr_gmm_destroy(gmm);
    aasr_feat_destroy(feat);
  } catch (std::string &s) {
    fprintf(stderr, "exception: %s\n", s.c_str());
    return 1;
  }
  return 0;
}
