// PhoneProbsToolbox.hh -- aku::PPToolbox (aku/PhoneProbsToolbox.hh:13-29) on
// top of the C ABI: same four entry points the SWIG wrapper
// aku/swig/PPToolbox.i:57-75 exposes.  Always normalised, 2-byte LNA, like
// the reference (aku/PhoneProbsToolbox.cc:84-131,160-207).
#ifndef AKU_AMD_PHONEPROBSTOOLBOX_HH
#define AKU_AMD_PHONEPROBSTOOLBOX_HH

#include <string>

#include "FeatureGenerator.hh"
#include "HmmSet.hh"

namespace aku {

class PPToolbox {
public:
  void read_configuration(const std::string &cfgname);
  void read_models(const std::string &base);
  /** aku/PhoneProbsToolbox.hh:18 */
  void set_clustering(const std::string &clfile_name, double eval_minc, double eval_ming);
  /** audio file (or raw PCM16 when raw is set) -> LNA file */
  void generate(const std::string &input, const std::string &output, bool raw = false);
  void generate_from_file_to_fd(const std::string &input, int out_fd, bool raw = false);
  void generate_to_fd(int in_fd, int out_fd, bool raw = true);

private:
  FeatureGenerator m_gen;
  HmmSet m_model;
};

}  // namespace aku

#endif
