// FeatureGenerator.hh -- aku::FeatureGenerator / aku::FeatureVec adapters on
// top of the C ABI (include/aasr.h).
//
// Same method names, signatures, argument meaning and error behaviour (thrown std::string)
// as aku/FeatureGenerator.hh:23-91, so callers written against aku -- phone_probs.cc:85-267,
// feacat.cc, PhoneProbsToolbox.cc:42-222, decoder/decode-stream.cc:70-117,177-206 -- compile
// against this header unchanged: tests/test_reference_callers.py feeds the reference's own
// source text to the compiler against these adapters.  generate(frame) is served from a block of frames the
// device computed in one go; the block is refilled on a miss.
#ifndef AKU_AMD_FEATUREGENERATOR_HH
#define AKU_AMD_FEATUREGENERATOR_HH

#include <cstdint>
#include <cstdio>
#include <string>
#include <memory>
#include <vector>

#include "FeatureBuffer.hh"
#include "FeatureModule.hh"
#include "ModuleConfig.hh"

#include "../../../include/aasr.h"

namespace aku {

class BaseFeaModule;

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

  /** aku/FeatureGenerator.cc:30-52: opens an audio file (or a feature file for `pre` graphs) */
  void open(const std::string &filename);
  /** aku/FeatureGenerator.hh:35, FeatureGenerator.cc:54-66: fdopen + open(file, false, false).
   * (The reference ignores raw_audio here; so does this.) */
  void open_fd(const int fd, bool raw_audio);
  /** aku/FeatureGenerator.hh:43, FeatureGenerator.cc:69-84.  With dont_fclose == false the FILE is
   * closed by close(), as in the reference.  stream == false: the file is read to its end here (the
   * engine computes whole blocks).  stream == true (decode-stream.cc:81: stdin): headerless 16-bit
   * samples in the audiofile module's byte order, as AudioReader::open(FILE*, ..., stream) reads a
   * non-seekable stream (aku/AudioReader.cc:112-142); the samples are read as generate() needs them
   * -- a block of set_stream_block_frames() frames plus the graph's look-ahead at a time -- so
   * frames come out while the producer is still writing.  eof() turns true on the same frame as
   * for a file. */
  void open(FILE *file, bool dont_fclose, bool stream = false);
  /** frames computed per device pass in stream mode (default 16 = 0.128 s at 125 frames/s) */
  void set_stream_block_frames(int n) { m_stream_block_frames = n > 0 ? n : 1; }
  /** in-memory audio (new: lets callers hand over samples they already hold) */
  void open_pcm(const int16_t *pcm, int64_t n_samples);
  void close();

  const FeatureVec generate(int frame);
  bool eof() const { return m_eof_on_last_frame; }
  int last_frame();
  /** first frame whose window crosses the end of the input (what eof() reports against) */
  int eof_frame();
  int sample_rate();
  float frame_rate();
  int dim();

  /** aku/FeatureGenerator.cc:389-408: the module structure in DOT format */
  void print_dot_graph(FILE *file);

  /** Makes the module class T usable as `type <T::type_str()>` in feature configurations -- what
   * adding an `else if` to FeatureGenerator::load_configuration does in the reference
   * (aku/FeatureGenerator.cc:145-175).  T derives from FeatureModule, has a default constructor
   * and a static type_str(), and overrides set_module_config / generate (FeatureModule.hh). */
  template <class T>
  static void register_module_type() {
    register_user_type(T::type_str(), []() -> FeatureModule * { return new T(); });
  }
  static void register_user_type(const char *type, FeatureModule *(*factory)());

  /** handle access for sibling adapters */
  aasr_feat *handle() const { return m_feat; }
  /** float32 features of the cached block that holds `frame` (nullptr if the
   * frame is not cached); used by HmmSet to score whole blocks */
  const float *block_f32(int frame, int *first, int *count) const;
  /** the same block in double (AASR_PREC_F64 scoring) */
  const double *block_f64(int frame, int *first, int *count) const;
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

  /** the generator whose cached block holds `p` (a frame handed out as FeatureVec / Vector),
   * and the frame it is; NULL when p is nobody's */
  static const FeatureGenerator *find_block(const double *p, int *frame);
  /** FeatureModule::set_config: rebuild the graph with one module's block replaced */
  void reconfigure_module(const std::string &name, const ModuleConfig &config);

private:
  void fill_block(int frame);
  void build(const std::string &text, bool keep_modules);
  void read_all(FILE *file);
  void stream_read_until(int frame);
  bool m_streaming = false, m_stream_eof = false, m_stream_big_endian = false;
  int m_stream_block_frames = 16;
  FILE *m_file = nullptr;
  bool m_dont_fclose = false;
  aasr_feat *m_feat;
  std::vector<int16_t> m_pcm;
  bool m_open, m_eof_on_last_frame;
  int m_block_first, m_block_count, m_block_frames;
  uint64_t m_block_serial;
  uint64_t m_epoch = 0;
  std::vector<std::unique_ptr<FeatureModule>> m_modules;
  std::vector<double> m_block;     // [count x dim]
  std::vector<float> m_block_f32;  // same block, float32 (device scoring input)
};

}  // namespace aku

#endif
