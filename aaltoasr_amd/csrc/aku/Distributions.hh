// Distributions.hh -- the likelihood surface of aku/Distributions.hh on the engine:
//   PDF::compute_likelihood / compute_log_likelihood(const Vector&)      (:66-69)
//   PDFPool::compute_likelihood(f, index), precompute_likelihoods, reset_cache, size, dim, get_pdf (:122-160)
//   Mixture::size, get_base_pdf_index, get_mixture_coefficient, get_base_pdf,
//            compute_likelihood / compute_log_likelihood                 (:795-812, 872-873)
//   Gaussian::get_mean / get_covariance (diagonal)                       (the E-step of MllrTrainer.cc:44-56)
// reached through HmmSet::get_pool / get_pool_pdf / get_emission_pdf (aku/HmmSet.hh:216-228), as
// aku/logl.cc:58-60 and aku/MllrTrainer.cc:44 do.  These are views: an object names one pool
// Gaussian or one mixture of the HmmSet's device-resident model.  A likelihood asked for a Vector
// that is a frame of a FeatureGenerator's cached block (FeatureVec::get_vector()) is served from
// the scores of that whole block, computed on the device the first time any of its frames is
// asked for; any other Vector is scored alone (one launch), cached by value until the next one.
// Training-side members (accumulate, estimate_parameters, sampling, KLD ...) are not built.
#ifndef AKU_AMD_DISTRIBUTIONS_HH
#define AKU_AMD_DISTRIBUTIONS_HH

#include <vector>

#include "FeatureBuffer.hh"
#include "LinearAlgebra.hh"

namespace aku {

class HmmSet;
class PDFPool;

class PDF {
public:
  virtual ~PDF() {}
  /* The likelihood of the current feature given this model */
  virtual double compute_likelihood(const Vector &f) const = 0;
  /* The log likelihood of the current feature given this model */
  virtual double compute_log_likelihood(const Vector &f) const = 0;
  int dim() const;

protected:
  friend class HmmSet;
  friend class PDFPool;
  HmmSet *m_set = nullptr;
  int m_index = 0;
};

/** one Gaussian of the pool */
class Gaussian : public PDF {
public:
  virtual double compute_likelihood(const Vector &f) const;
  virtual double compute_log_likelihood(const Vector &f) const;
  void get_mean(Vector &mean) const;
  /** diagonal of the covariance (the variances of a DiagonalGaussian) */
  void get_covariance(Vector &covariance) const;
};
typedef Gaussian DiagonalGaussian;

class PDFPool {
public:
  int size() const;
  int dim() const;
  PDF *get_pdf(int index) const;
  void reset_cache() {}
  /** Uses cache (aku/Distributions.cc:2636-2644) */
  double compute_likelihood(const Vector &f, int index);
  void precompute_likelihoods(const Vector &f);

private:
  friend class HmmSet;
  HmmSet *m_set = nullptr;
};

class Mixture : public PDF {
public:
  int size() const { return (int)m_pointers.size(); }
  int get_base_pdf_index(int index) const { return m_pointers[(size_t)index]; }
  double get_mixture_coefficient(int index) const { return m_weights[(size_t)index]; }
  PDF *get_base_pdf(int index);
  /** sum_k w_k N_k(f) (aku/Distributions.cc:2078-2086).  Served from the block scores, which
   * carry HmmSet's floor: values below 1e-50 read as 1e-50 (aku/HmmSet.cc:497-498). */
  virtual double compute_likelihood(const Vector &f) const;
  /** util::safe_log of it (aku/Distributions.cc:2089-2097) */
  virtual double compute_log_likelihood(const Vector &f) const;

private:
  friend class HmmSet;
  std::vector<int> m_pointers;
  std::vector<double> m_weights;
};

}  // namespace aku

#endif
