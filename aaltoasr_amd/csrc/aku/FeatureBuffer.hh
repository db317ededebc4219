// FeatureBuffer.hh -- aku::FeatureVec for the adapters (aku/FeatureBuffer.hh:15-89).
//
// In the reference a FeatureVec is a pointer to one Vector of a module's ring buffer, valid until
// that slot is regenerated (aku/FeatureModules.cc:102-158).  Here it is a view of one frame of the
// block a FeatureGenerator (or FeatureModule::at) computed on the device: same accessors -- const
// and mutable operator[] with the "FeatureVec out of bounds" std::string, dim(), get(), set(),
// copy(), get_vector() -- valid until the block is refilled.  A default-constructed FeatureVec
// can be assigned to (decoder/decode-stream.cc:177-181).
#ifndef AKU_AMD_FEATUREBUFFER_HH
#define AKU_AMD_FEATUREBUFFER_HH

#include <assert.h>
#include <string>
#include <vector>

#include "LinearAlgebra.hh"
#include "util.hh"

namespace aku {

class FeatureGenerator;

class FeatureVec {
public:
  FeatureVec() : m_data(nullptr), m_dim(0), m_frame(0), m_owner(nullptr) {}
  /** aku/FeatureBuffer.hh:26: a vector held elsewhere */
  FeatureVec(const Vector *ptr, int dim)
      : m_data(ptr ? ptr->addr() : nullptr), m_dim(dim), m_frame(0), m_owner(nullptr) {}
  /** one frame of a generator's block */
  FeatureVec(const double *data, int dim, int frame, const FeatureGenerator *owner)
      : m_data(data), m_dim(dim), m_frame(frame), m_owner(owner) {}
  FeatureVec(const FeatureVec &o) : m_data(o.m_data), m_dim(o.m_dim), m_frame(o.m_frame), m_owner(o.m_owner) {}
  FeatureVec &operator=(const FeatureVec &o) {
    m_data = o.m_data;
    m_dim = o.m_dim;
    m_frame = o.m_frame;
    m_owner = o.m_owner;
    return *this;
  }

  void copy(const FeatureVec &vec) {
    assert(vec.dim() == m_dim);
    for (int i = 0; i < m_dim; i++) mut()[i] = vec[i];
  }
  void set(const std::vector<double> &vec) {
    assert((int)vec.size() == m_dim);
    for (int i = 0; i < m_dim; i++) mut()[i] = vec[i];
  }
  void set(const std::vector<float> &vec) {
    assert((int)vec.size() == m_dim);
    for (int i = 0; i < m_dim; i++) mut()[i] = vec[i];
  }
  void get(std::vector<double> &vec) const { vec.assign(m_data, m_data + m_dim); }
  void get(std::vector<float> &vec) const {
    vec.resize(m_dim);
    for (int i = 0; i < m_dim; i++) vec[i] = (float)m_data[i];
  }
  const double &operator[](int index) const {
    if (index < 0 || index >= m_dim) throw std::string("FeatureVec out of bounds");
    return m_data[index];
  }
  double &operator[](int index) {
    if (index < 0 || index >= m_dim) throw std::string("FeatureVec out of bounds");
    return mut()[index];
  }
  int dim() const { return m_dim; }

  /** the frame as a Vector: a view of the same memory (LaVectorDouble::ref), valid as long as
   * this object and the block it points into */
  const Vector *get_vector() const {
    m_view.ref(mut(), m_dim);
    return &m_view;
  }

  const double *data() const { return m_data; }
  /** frame index and generator this vector came from (block-cache key of HmmSet) */
  int frame() const { return m_frame; }
  const FeatureGenerator *owner() const { return m_owner; }

private:
  double *mut() const { return const_cast<double *>(m_data); }
  const double *m_data;
  int m_dim, m_frame;
  const FeatureGenerator *m_owner;
  mutable Vector m_view;
};

}  // namespace aku

#endif
