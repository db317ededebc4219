// FeatureModules.hh -- the module classes callers name explicitly: BaseFeaModule
// (aku/BaseFeaModule.hh:10-27) with eof / sample_rate / frame_rate / last_frame, and its two
// implementations AudioFileModule (aku/AudioFileModule.hh) and PreModule, and VtlnModule with its
// warp-factor accessors.  All other module types of a graph are plain FeatureModule handles (their
// classes have no members callers use).
#ifndef AKU_AMD_FEATUREMODULES_HH
#define AKU_AMD_FEATUREMODULES_HH

#include "FeatureGenerator.hh"
#include "FeatureModule.hh"

namespace aku {

class BaseFeaModule : public FeatureModule {
public:
  /** true for frames whose window crosses the end of the input (aku/FeatureModules.cc:297-303) */
  virtual bool eof(int frame);
  virtual int sample_rate(void);
  virtual float frame_rate(void);
  virtual int last_frame(void);
};

class AudioFileModule : public BaseFeaModule {};
class PreModule : public BaseFeaModule {};

/** aku/FeatureModules.hh:215-225: the warp factor of a `vtln` module as the VTLN estimation tool
 * drives it (aku/vtln.cc:70-75); both go through the module's parameter block */
class VtlnModule : public FeatureModule {
public:
  static const char *type_str() { return "vtln"; }
  void set_warp_factor(float factor);
  void set_slapt_warp(std::vector<float> &params);
  float get_warp_factor(void);
};

/** aku/FeatureModules.hh:129-144, 147-171, 279-301: the module classes whose parameters the
 * estimation tools set directly (feanorm: set_normalization / set_transformation_matrix, quanteq:
 * set_alpha / set_gamma / set_quant_max).  Each setter goes through the module's parameter block
 * (floats as "%.9g", i.e. bit for bit); the arithmetic stays on the device. */
class NormalizationModule : public FeatureModule {
public:
  static const char *type_str() { return "normalization"; }
  void set_normalization(const std::vector<float> &mean, const std::vector<float> &scale);
};

class LinTransformModule : public FeatureModule {
public:
  static const char *type_str() { return "lin_transform"; }
  const std::vector<float> *get_transformation_matrix(void);
  const std::vector<float> *get_transformation_bias(void);
  void set_transformation_matrix(std::vector<float> &t);
  void set_transformation_bias(std::vector<float> &b);

private:
  std::vector<float> m_transform, m_bias;  // what the accessors point to
};

class QuantEqModule : public FeatureModule {
public:
  static const char *type_str() { return "quanteq"; }
  void set_alpha(std::vector<float> &alpha);
  void set_gamma(std::vector<float> &gamma);
  void set_quant_max(std::vector<float> &quant_max);
  std::vector<float> get_quant_train(void);
};

}  // namespace aku

#endif
