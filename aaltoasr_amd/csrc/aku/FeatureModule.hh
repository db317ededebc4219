// FeatureModule.hh -- aku::FeatureModule as users of a FeatureGenerator see it
// (aku/FeatureModule.hh:47-154): name(), type_str(), dim(), sources(), get_config() /
// set_config(), set_parameters() / get_parameters() (speaker adaptation) and at(frame), the
// module's own output.  The arithmetic of the built-in module types runs on the device
// (csrc/feat_kernels.hip); an object of this class is the host-side handle of one module of the
// loaded graph.  BaseFeaModule / AudioFileModule / PreModule (FeatureModules.hh) derive from it so
// that `dynamic_cast<AudioFileModule*>(gen.module("audiofile"))` works as in
// decoder/decode-stream.cc:94-98.
#ifndef AKU_AMD_FEATUREMODULE_HH
#define AKU_AMD_FEATUREMODULE_HH

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "FeatureBuffer.hh"
#include "ModuleConfig.hh"

namespace aku {

class FeatureGenerator;
class FeatureModule;

/** aku/FeatureBuffer.hh:92-143.  Two uses: (a) a circular store of feature vectors a caller keeps for
 * itself -- resize(num_frames, dim), then operator[](frame) addresses slot frame mod num_frames, as
 * HmmNetBaumWelch does with the frames of its window (aku/HmmNetBaumWelch.cc:806-835); (b) what a
 * user module's generate(frame) writes into, m_buffer[frame], reduced to the one slot that is being
 * computed. */
class FeatureBuffer {
public:
  FeatureBuffer() {}
  void resize(int num_frames, int dim) {
    assert(num_frames > 0 && dim > 0);
    m_num_frames = num_frames;
    m_dim = dim;
    m_store.assign((size_t)num_frames * dim, 0.0);
    m_row = nullptr;
  }
  void clear(void) { resize(1, 1); }
  int dim() const { return m_dim; }
  int num_frames() const { return m_num_frames; }
  const FeatureVec operator[](int frame) const {
    if (m_store.empty()) throw std::string("FeatureBuffer: empty");
    return FeatureVec(&m_store[(size_t)util::modulo(frame, m_num_frames) * m_dim], m_dim, frame, nullptr);
  }
  FeatureVec operator[](int frame) {
    if (m_row) {  // (b): the slot of the frame being generated
      if (frame != m_frame) throw std::string("FeatureBuffer: only the frame being generated can be written");
      return FeatureVec(m_row, m_dim, frame, nullptr);
    }
    if (m_store.empty()) throw std::string("FeatureBuffer: only the frame being generated can be written");
    return FeatureVec(&m_store[(size_t)util::modulo(frame, m_num_frames) * m_dim], m_dim, frame, nullptr);
  }

private:
  friend class FeatureModule;
  friend struct UserModuleGlue;
  double *m_row = nullptr;
  int m_dim = 0, m_frame = 0, m_num_frames = 0;
  std::vector<double> m_store;
};

class FeatureModule {
public:
  FeatureModule() {}
  virtual ~FeatureModule() {}

  std::string name() const { return m_name; }
  std::string type_str() const { return m_type_str; }
  int dim(void) { return m_dim; }
  const std::vector<FeatureModule *> &sources() const { return m_sources; }

  /** aku/FeatureModule.hh:88-93: everything needed to configure an identical module ("name"
   * and "type" included, as FeatureModule::get_config writes them, aku/FeatureModules.cc:173-179) */
  void get_config(ModuleConfig &config);
  /** reconfigures this module: the graph is rebuilt with the new settings (module objects stay) */
  void set_config(const ModuleConfig &config);

  virtual void set_parameters(const ModuleConfig &params);
  virtual void get_parameters(ModuleConfig &params);

  /** this module's feature vector at `frame` (computed through the graph, cached in blocks) */
  const FeatureVec at(int frame);

  /** aku/FeatureModules.cc:202-217 */
  void print_dot_node(FILE *file);

private:
  // A user-defined module type overrides these exactly as it would in the reference
  // (aku/FeatureModule.hh:131-139) and is made known with
  // FeatureGenerator::register_module_type<T>(): set_module_config checks its sources
  // (m_sources[k]->dim()) and sets m_dim and m_own_offset_left / _right; generate(frame) reads
  // m_sources[k]->at(frame + d), |d| within those offsets, and fills m_buffer[frame].  It runs on
  // the host, once per frame (include/aasr.h: aasr_feat_register_module_type).
  virtual void set_module_config(const ModuleConfig &config) { (void)config; }
  virtual void get_module_config(ModuleConfig &config) { (void)config; }
  virtual void reset_module() {}
  virtual void generate(int frame) { (void)frame; }

protected:
  friend class FeatureGenerator;
  friend struct UserModuleGlue;
  FeatureBuffer m_buffer;
  // while a user module's generate() runs, its sources serve at() from the rows the engine handed over
  const double *m_eval_rows = nullptr;
  int m_eval_first = 0, m_eval_count = 0;
  bool m_user = false;  // a user module (or one of its source proxies), not a handle of a loaded graph
  FeatureGenerator *m_gen = nullptr;
  std::string m_name, m_type_str;
  int m_dim = 0;
  int m_own_offset_left = 0, m_own_offset_right = 0;   // look-around of the module itself
  int m_req_offset_left = 0, m_req_offset_right = 0;   // what its consumers need on top
  std::vector<FeatureModule *> m_sources;
  ModuleConfig m_config;
  int m_first = 0, m_count = 0;
  uint64_t m_epoch = 0;
  std::vector<double> m_block;
};

}  // namespace aku

#endif
