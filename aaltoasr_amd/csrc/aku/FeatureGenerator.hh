// FeatureGenerator.hh -- aku::FeatureGenerator / aku::FeatureVec adapters on
// top of the C ABI (include/aasr.h).
//
// Same method names, argument meaning and error behaviour (thrown std::string)
// as aku/FeatureGenerator.hh:23-91 and aku/FeatureBuffer.hh:15-89, so callers
// written against aku -- phone_probs.cc:85-267, PhoneProbsToolbox.cc:42-222,
// decoder/decode-stream.cc:238-276 -- compile against this header unchanged
// for the scoring path.  generate(frame) is served from a block of frames the
// device computed in one go; the block is refilled on a miss.
#ifndef AKU_AMD_FEATUREGENERATOR_HH
#define AKU_AMD_FEATUREGENERATOR_HH

#include <cstdint>
#include <cstdio>
#include <string>
#include "ModuleConfig.hh"
#include <vector>

#include "../../../include/aasr.h"

namespace aku {

class FeatureGenerator;

/** Borrowed view of one frame's feature vector (double, like LaVectorDouble).
 * Valid until the generator refills its block, mirroring the reference where a
 * FeatureVec points into a module ring buffer (aku/FeatureModules.cc:102-158). */
class FeatureVec {
public:
  FeatureVec() : m_data(nullptr), m_dim(0), m_frame(0), m_owner(nullptr) {}
  FeatureVec(const double *data, int dim, int frame, const FeatureGenerator *owner)
      : m_data(data), m_dim(dim), m_frame(frame), m_owner(owner) {}
  const double &operator[](int index) const {
    if (index < 0 || index >= m_dim) throw std::string("FeatureVec out of range");
    return m_data[index];
  }
  int dim() const { return m_dim; }
  void get(std::vector<float> &vec) const {
    vec.resize(m_dim);
    for (int i = 0; i < m_dim; i++) vec[i] = (float)m_data[i];
  }
  void get(std::vector<double> &vec) const { vec.assign(m_data, m_data + m_dim); }
  const double *data() const { return m_data; }
  /** frame index and generator this vector came from (block-cache key) */
  int frame() const { return m_frame; }
  const FeatureGenerator *owner() const { return m_owner; }

private:
  const double *m_data;
  int m_dim, m_frame;
  const FeatureGenerator *m_owner;
};

/** One module of the graph with the calls of aku::FeatureModule that users of a generator reach
 * through FeatureGenerator::module(name) (aku/FeatureModule.hh:47-154): name / type_str / dim,
 * set_parameters / get_parameters (speaker adaptation, aku/FeatureModules.cc per module) and at(frame),
 * the module's own output.  The arithmetic runs on the device; this object is a view. */
class FeatureModule {
public:
  const std::string &name() const { return m_name; }
  const std::string &type_str() const { return m_type; }
  int dim() const { return m_dim; }
  void set_parameters(const ModuleConfig &config);
  void get_parameters(ModuleConfig &config) const;
  /** this module's feature vector at `frame` (computed through the graph, cached in blocks) */
  const FeatureVec at(int frame);

private:
  friend class FeatureGenerator;
  FeatureGenerator *m_gen = nullptr;
  std::string m_name, m_type;
  int m_dim = 0;
  int m_first = 0, m_count = 0;
  uint64_t m_epoch = 0;
  std::vector<double> m_block;
};

class FeatureGenerator {
public:
  FeatureGenerator();
  ~FeatureGenerator();

  /** aku/FeatureGenerator.cc:96-219 */
  void load_configuration(FILE *file);
  void load_configuration_text(const std::string &text);
  void close_configuration();
  /** aku/FeatureGenerator.cc:222-243 */
  void write_configuration(FILE *file);
  /** aku/FeatureGenerator.cc:257-265: throws std::string("unknown module requested: " + name) */
  FeatureModule *module(const std::string &name);

  /** aku/FeatureGenerator.cc:30-52: opens a PCM16 WAV (or raw) file */
  void open(const std::string &filename);
  void open(FILE *file, bool stream = false);
  /** in-memory audio (new: lets callers hand over samples they already hold) */
  void open_pcm(const int16_t *pcm, int64_t n_samples);
  void close();

  const FeatureVec generate(int frame);
  bool eof() const { return m_eof_on_last_frame; }
  int last_frame();
  int sample_rate();
  float frame_rate();
  int dim();

  /** handle access for sibling adapters */
  aasr_feat *handle() const { return m_feat; }
  /** float32 features of the cached block that holds `frame` (nullptr if the
   * frame is not cached); used by HmmSet to score whole blocks */
  const float *block_f32(int frame, int *first, int *count) const;
  uint64_t block_serial() const { return m_block_serial; }
  void set_block_frames(int n) { m_block_frames = n > 0 ? n : 1; }
  /** module parameters changed (SpeakerConfig): cached frames are stale */
  void invalidate_block() {
    m_block_count = 0;
    m_epoch++;
  }
  /** bumped whenever cached frames become stale (parameters, new input) */
  uint64_t epoch() const { return m_epoch; }
  /** the open input in the engine's int16 units */
  const std::vector<int16_t> &input_units() const { return m_pcm; }

private:
  void fill_block(int frame);
  aasr_feat *m_feat;
  std::vector<int16_t> m_pcm;
  bool m_open, m_eof_on_last_frame;
  int m_block_first, m_block_count, m_block_frames;
  uint64_t m_block_serial;
  uint64_t m_epoch = 0;
  std::vector<FeatureModule> m_modules;
  std::vector<double> m_block;     // [count x dim]
  std::vector<float> m_block_f32;  // same block, float32 (device scoring input)
};

}  // namespace aku

#endif
