// ziggurat.hh -- ziggurat::rnd.rnor(), the standard-normal generator feacat -G draws its feature
// noise from (aku/feacat.cc:38-42).  The reference's generator is Marsaglia & Tsang's ziggurat
// with a global state; its stream is not part of any output contract (noise), so this adapter
// draws from std::mt19937_64 / std::normal_distribution with a fixed seed.
#ifndef AKU_AMD_ZIGGURAT_HH
#define AKU_AMD_ZIGGURAT_HH

#include <random>

namespace ziggurat {

struct Ziggurat {
  Ziggurat() : m_gen(20260928u), m_norm(0.0, 1.0) {}
  /** N(0, 1) */
  double rnor() { return m_norm(m_gen); }
  void seed(unsigned long s) { m_gen.seed(s); }

private:
  std::mt19937_64 m_gen;
  std::normal_distribution<double> m_norm;
};

// one generator per process, like the reference's global
inline Ziggurat &instance() {
  static Ziggurat z;
  return z;
}
static Ziggurat &rnd = instance();

}  // namespace ziggurat

#endif
