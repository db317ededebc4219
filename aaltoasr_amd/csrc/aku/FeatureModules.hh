// FeatureModules.hh -- the module classes callers name explicitly: BaseFeaModule
// (aku/BaseFeaModule.hh:10-27) with eof / sample_rate / frame_rate / last_frame, and its two
// implementations AudioFileModule (aku/AudioFileModule.hh) and PreModule.  All other module
// types of a graph are plain FeatureModule handles (their classes have no members callers use).
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

}  // namespace aku

#endif
