// EngineAcoustics.hh -- in-memory hand-off of acoustic log-probabilities to the
// decoder.
//
// The reference decoder pulls per-frame state log-probabilities through the
// `Acoustics` interface (decoder/src/Acoustics.hh:7-25): go_to(frame) positions
// the source, log_prob(model) reads one value.  Its stock implementations read
// an LNA file written earlier by phone_probs (LnaReaderCircular,
// decoder/src/LnaReaderCircular.cc:129-209) or take one frame at a time from the
// caller (OneFrameAcoustics).  EngineAcoustics is a third implementation: it
// owns the audio of one utterance and computes blocks of frames on the GPU on
// demand, so the 2 x frames x states bytes per utterance never touch the disk.
// What log_prob() returns is exactly what LnaReaderCircular returns for the LNA
// file phone_probs would have written: with lnabytes = 4 the float log-prob,
// with lnabytes = 2 the quantised value -(hi*256 + lo)/1820.
//
// To use it in the reference decoder: include the decoder's own Acoustics.hh
// BEFORE this header (it then derives from that class) and pass the object to
// TokenPassSearch::set_acoustics / Toolbox (decoder/src/Toolbox.hh:120-129).
// Stand-alone builds (this repository's tests) get an interface with the same
// members from this header.
#ifndef AASR_ENGINE_ACOUSTICS_HH
#define AASR_ENGINE_ACOUSTICS_HH

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/aasr.h"

#ifndef ACOUSTICS_HH
#define ACOUSTICS_HH
// Same members as decoder/src/Acoustics.hh so that either definition can be used.
class Acoustics {
public:
  Acoustics() : m_log_prob(0), m_num_models(0) {}
  virtual ~Acoustics() {}
  virtual bool go_to(int frame) = 0;
  float log_prob(int model) const { return m_log_prob[model]; }
  int num_models() const { return m_num_models; }

protected:
  float *m_log_prob;
  int m_num_models;
};
#endif

class EngineAcoustics : public Acoustics {
public:
  /** The handles are borrowed.  block_frames = frames computed per device call. */
  EngineAcoustics(aasr_feat *feat, aasr_gmm *gmm, int lnabytes = 4, bool normalize = true,
                  int block_frames = 4096);
  virtual ~EngineAcoustics() {}

  /** One utterance: PCM16 samples at the feature configuration's sample rate
   * (or, for a `pre` base module, float frames reinterpreted -- see aasr.h). */
  void open_pcm(const int16_t *pcm, int64_t n_samples);
  /** WAV / raw PCM16 file (or feature file for a `pre` base module) */
  void open_file(const std::string &path);
  void close();

  /** Acoustics::go_to: false from the first frame past the end of the audio, like
   * LnaReaderCircular::go_to on the LNA file phone_probs writes. */
  virtual bool go_to(int frame);

  int eof_frame() const { return m_eof_frame; }

private:
  void fill(int frame);
  aasr_feat *m_feat;
  aasr_gmm *m_gmm;
  int m_lnabytes, m_block_frames;
  bool m_normalize;
  std::vector<int16_t> m_pcm;
  int m_eof_frame;
  int m_block_first, m_block_count;
  std::vector<float> m_block;  // [count x num_models]
};

#endif
