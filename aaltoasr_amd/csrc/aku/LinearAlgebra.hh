// LinearAlgebra.hh -- aku::Vector for the adapters.
//
// In the reference `Vector` is LapackPP's LaVectorDouble (aku/LinearAlgebra.hh:13-15).  Callers on
// the scoring path only index it (`v(i)`), ask its size and hand it on
// (`pdf->compute_likelihood(*f.get_vector())`, aku/MllrTrainer.cc:44), so the adapter's Vector is
// a plain double array: owning when constructed with a size, a borrowed view when constructed
// over memory (the feature block of a FeatureGenerator).
#ifndef AKU_AMD_LINEARALGEBRA_HH
#define AKU_AMD_LINEARALGEBRA_HH

#include <vector>

namespace aku {

class Vector {
public:
  Vector() : m_p(nullptr), m_n(0) {}
  explicit Vector(int n) : m_own((size_t)n, 0.0), m_p(m_own.data()), m_n(n) {}
  /** borrowed view (not copied, not freed) */
  Vector(double *p, int n) : m_p(p), m_n(n) {}
  Vector(const Vector &o) { assign(o); }
  Vector &operator=(const Vector &o) {
    if (this != &o) assign(o);
    return *this;
  }
  /** LaVectorDouble: scalar assignment fills */
  Vector &operator=(double s) {
    for (int i = 0; i < m_n; i++) m_p[i] = s;
    return *this;
  }
  double &operator()(int i) { return m_p[i]; }
  const double &operator()(int i) const { return m_p[i]; }
  int size() const { return m_n; }
  void resize(int n) {
    m_own.assign((size_t)n, 0.0);
    m_p = m_own.data();
    m_n = n;
  }
  /** LaVectorDouble::ref: become a view of memory held elsewhere */
  void ref(double *p, int n) {
    m_own.clear();
    m_p = p;
    m_n = n;
  }
  /** LaVectorDouble::copy */
  void copy(const Vector &o) { assign(o); }
  const double *addr() const { return m_p; }
  double *addr() { return m_p; }

private:
  void assign(const Vector &o) {  // copies own their data, like LaVectorDouble::copy
    m_own.assign(o.m_p, o.m_p + o.m_n);
    m_p = m_own.data();
    m_n = o.m_n;
  }
  std::vector<double> m_own;
  double *m_p;
  int m_n;
};

}  // namespace aku

#endif
