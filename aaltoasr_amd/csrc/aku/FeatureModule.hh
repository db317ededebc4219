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

protected:
  friend class FeatureGenerator;
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
