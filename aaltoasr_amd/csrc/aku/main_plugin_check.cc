// plugin_check -- user-defined feature module types on the adapter classes.
//
// Two modules written the way a module is written for the reference (a FeatureModule subclass
// with a static type_str(), set_module_config checking its source and setting m_dim and the
// offsets, generate(frame) reading m_sources.back()->at(frame + k) and filling m_buffer[frame]):
//   my_delta   the arithmetic of DeltaModule (aku/FeatureModules.cc:998-1037), so that a graph
//              using it must print exactly what the same graph with the built-in `delta` prints;
//   my_gain    y = gain * x + offset, reading two sources (x from the first, the offset from
//              element 0 of the second).
// usage: plugin_check CFG AUDIO FIRST_FRAME N_FRAMES   (prints "%.17g " per value, one frame per line)
#include <cstdio>
#include <cstdlib>
#include <string>

#include "FeatureGenerator.hh"
#include "FeatureModules.hh"

class MyDelta : public aku::FeatureModule {
public:
  static const char *type_str() { return "my_delta"; }

private:
  virtual void get_module_config(aku::ModuleConfig &config) { config.set("width", m_delta_width); }
  virtual void set_module_config(const aku::ModuleConfig &config) {
    m_dim = m_sources.back()->dim();
    m_delta_width = 2;
    config.get("width", m_delta_width);
    if (m_delta_width < 1) throw std::string("MyDelta: Delta width must be greater than zero");
    m_own_offset_left = m_delta_width;
    m_own_offset_right = m_delta_width;
    m_delta_norm = 2 * m_delta_width * (m_delta_width + 1) * (2 * m_delta_width + 1) / 6;
  }
  virtual void generate(int frame) {
    aku::FeatureVec target_fea = m_buffer[frame];
    for (int i = 0; i < m_dim; i++) target_fea[i] = 0;
    for (int k = 1; k <= m_delta_width; k++) {
      const aku::FeatureVec left = m_sources.back()->at(frame - k);
      const aku::FeatureVec right = m_sources.back()->at(frame + k);
      for (int i = 0; i < m_dim; i++) target_fea[i] += k * (right[i] - left[i]);
    }
    for (int i = 0; i < m_dim; i++) target_fea[i] /= m_delta_norm;
  }
  int m_delta_width;
  float m_delta_norm;
};

class MyGain : public aku::FeatureModule {
public:
  static const char *type_str() { return "my_gain"; }

private:
  virtual void set_module_config(const aku::ModuleConfig &config) {
    if (m_sources.size() != 2) throw std::string("MyGain: two sources expected");
    m_dim = m_sources[0]->dim();
    m_gain = 1;
    config.get("gain", m_gain);
  }
  virtual void generate(int frame) {
    aku::FeatureVec out = m_buffer[frame];
    const aku::FeatureVec x = m_sources[0]->at(frame);
    const aku::FeatureVec o = m_sources[1]->at(frame);
    for (int i = 0; i < m_dim; i++) out[i] = (double)m_gain * x[i] + o[0];
  }
  float m_gain;
};

int main(int argc, char **argv) {
  if (argc != 5) {
    fprintf(stderr, "usage: plugin_check CFG AUDIO FIRST_FRAME N_FRAMES\n");
    return 2;
  }
  try {
    aku::FeatureGenerator::register_module_type<MyDelta>();
    aku::FeatureGenerator::register_module_type<MyGain>();
    aku::FeatureGenerator gen;
    FILE *cf = fopen(argv[1], "r");
    if (!cf) throw std::string("could not open config");
    gen.load_configuration(cf);
    fclose(cf);
    gen.open(argv[2]);
    const int first = atoi(argv[3]), n = atoi(argv[4]);
    for (int f = first; f < first + n; f++) {
      const aku::FeatureVec fea = gen.generate(f);
      for (int i = 0; i < fea.dim(); i++) printf("%.17g ", fea[i]);
      printf("\n");
    }
    gen.write_configuration(stderr);
  } catch (std::string &e) {
    fprintf(stderr, "exception: %s\n", e.c_str());
    return 1;
  }
  return 0;
}
