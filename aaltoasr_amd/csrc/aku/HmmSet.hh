// HmmSet.hh -- aku::HmmSet scoring surface on top of the C ABI.
//
// Mirrors the subset of aku/HmmSet.hh callers on the scoring path use
// (:120,163,246-271,299-321,474-493,511-534): read_all / read_gk / read_mc /
// read_ph, dim, num_states, reset_cache, precompute_likelihoods,
// state_likelihood, pdf_likelihood, and the OpenError / ReadError exceptions.
// Likelihood caches in the reference are caller-managed per frame
// (reset_cache() whenever the frame changes, HmmSet.cc:444-457); here a frame
// that came from a FeatureGenerator block is scored together with its whole
// block on the device the first time any of its frames is asked for, and
// reset_cache() only drops the current-frame view.
#ifndef AKU_AMD_HMMSET_HH
#define AKU_AMD_HMMSET_HH

#include <exception>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include <assert.h>
#include <memory>

#include "Distributions.hh"
#include "FeatureGenerator.hh"

namespace aku {

/** aku/HmmSet.hh:19-82: the topology half of the model -- states with their transitions, phone
 * HMMs as lists of state indices.  Host-side data read from the .ph file; what the aligners
 * (aku/Viterbi.cc, aku/PhnReader.cc, aku/HmmNetBaumWelch.cc) walk. */
class HmmState {
public:
  HmmState() {}
  HmmState(int pdf_index) : emission_pdf(pdf_index) {}
  inline std::vector<int> &transitions() { return m_transitions; }
  int emission_pdf = 0;             //!< index of the emission PDF
  std::vector<int> m_transitions;   //!< indices into HmmSet::transition()
};

struct HmmTransition {
  HmmTransition() {}
  HmmTransition(int source, int target, double prob) : source_index(source), target_offset(target), prob(prob) {}
  int source_index = 0;   //!< source state
  int target_offset = 0;  //!< target relative to the source's position in its HMM
  double prob = 0;
};

class Hmm {
public:
  std::string label;
  void resize(int states) { m_states.resize(states); }
  inline int num_states() const { return (int)m_states.size(); }
  inline int &state(int index) { return m_states[index]; }
  /** aku/HmmSet.cc:21-40: "a-b+c" -> "b" */
  std::string get_center_phone();

private:
  std::vector<int> m_states;
};

/** aku/HmmSet.hh:86-89 */
struct ResetCacheInterface {
  virtual void reset_cache() = 0;
  virtual ~ResetCacheInterface() {}
};

class HmmSet {
public:
  struct OpenError : public std::exception {
    virtual const char *what() const throw() { return "HmmSet: open error"; }
  };
  struct ReadError : public std::exception {
    virtual const char *what() const throw() { return "HmmSet: read error"; }
  };
  struct DuplicateHmm : public std::exception {
    virtual const char *what() const throw() { return "HmmSet: duplicate hmm"; }
  };
  struct UnknownHmm : public std::exception {
    virtual const char *what() const throw() { return "HmmSet: unknown hmm"; }
  };

  /** aku/HmmSet.hh:126-214 -- the HMM inventory of the .ph file (aku/HmmSet.cc:183-329) */
  Hmm &new_hmm(const std::string &label);
  Hmm &add_hmm(const std::string &label, int num_states);
  int num_hmms() const { return (int)m_hmms.size(); }
  Hmm &hmm(int hmm) { return m_hmms[hmm]; }
  Hmm &hmm(const std::string &label) { return m_hmms[hmm_index(label)]; }
  int hmm_index(const std::string &label) const;
  HmmState &state(int state) { return m_states[state]; }
  int add_transition(int source, int target, double prob);
  int add_state(int pdf_index);
  int num_transitions() const { return (int)m_transitions.size(); }
  HmmTransition &transition(int t) { return m_transitions[t]; }
  /** aku/HmmSet.hh:300-302: objects reset together with the likelihood cache */
  void register_reset_cache_object(ResetCacheInterface *obj) { m_reset_cache_objects.insert(obj); }
  void unregister_reset_cache_object(ResetCacheInterface *obj) { m_reset_cache_objects.erase(obj); }

  HmmSet();
  ~HmmSet();

  void read_all(const std::string &base);
  void read_gk(const std::string &filename);
  void read_mc(const std::string &filename);
  bool read_ph(const std::string &filename);

  int dim();
  int num_states();
  int num_emission_pdfs() { return num_states(); }

  /** aku/HmmSet.hh:477,492 -- Gaussian clustering for the likelihood precomputation */
  void read_clustering(const std::string &filename);
  void set_clustering_min_evals(double min_clusters = 1.0, double min_gaussians = 1.0);

  void reset_cache();
  void precompute_likelihoods(const FeatureVec &f);
  double state_likelihood(const int s, const FeatureVec &f);
  double pdf_likelihood(const int p, const FeatureVec &f) { return state_likelihood(p, f); }

  /** aku/HmmSet.hh:216-228: the pool, one of its Gaussians, one state's mixture -- views on the
   * device-resident model (Distributions.hh) */
  PDFPool *get_pool();
  PDF *get_pool_pdf(int index);
  Mixture *get_emission_pdf(int index);
  /** aku/HmmSet.hh:120: emission pdf of a state (the legacy .ph format: the state's own index) */
  int emission_pdf_index(int state) const {
    return state < (int)m_states.size() ? m_states[(size_t)state].emission_pdf : state;
  }

  /** likelihoods for a raw vector (what the PDF / Mixture views call): the row of state
   * log-likelihoods, from the block cache when `x` is a frame of a generator's block */
  const float *state_loglik_row(const double *x, int dim);
  double pool_log_likelihood(const int g, const double *x, int dim);

  /** PDFPool::size / compute_likelihood / compute_log_likelihood for one pool Gaussian
   * (aku/Distributions.hh:262-263, aku/Distributions.cc:2636-2644, 1033-1062): the whole pool is
   * scored for the frame on first use (aasr_gmm_gauss_loglik) and kept until the frame changes */
  int num_pool_pdfs();
  double pool_log_likelihood(const int g, const FeatureVec &f);
  double pool_likelihood(const int g, const FeatureVec &f);

private:
  std::vector<float> m_pool_ll;   // per-Gaussian log-likelihoods of m_pool_frame
  std::vector<double> m_pool_x;   // the frame they belong to (by value)
public:
  aasr_gmm *handle() {
    ensure_model();
    return m_gmm;
  }
  /** the model changed underneath (SpeakerConfig loaded a transform) */
  void invalidate_block() {
    m_count = 0;
    m_row = nullptr;
    m_row64 = nullptr;
  }
  /** log state likelihoods of the cached block row for f (S floats) */
  const float *state_loglik_row(const FeatureVec &f);

private:
  void ensure_model();
  void drop_model();
  void read_legacy_ph(std::ifstream &in);
  std::map<std::string, int> m_hmm_map;
  std::vector<Hmm> m_hmms;
  std::vector<HmmState> m_states;
  std::vector<HmmTransition> m_transitions;
  std::set<ResetCacheInterface *> m_reset_cache_objects;
  std::string m_gk, m_mc, m_ph;
  aasr_gmm *m_gmm;
  // block cache
  const FeatureGenerator *m_owner;
  uint64_t m_serial;
  int m_first, m_count;
  std::vector<float> m_block_ll;  // [count x S]
  // AASR_PREC_F64 (aasr_gmm_set_precision / AASR_PREC=1): the same rows in double, what
  // state_likelihood() then returns -- the aligners and trainers get the reference's arithmetic
  std::vector<double> m_block_ll64, m_single_ll64;
  const double *m_row64 = nullptr;
  const double *row64_for(const float *row);
  std::vector<float> m_single_ll;  // one frame, for vectors without a block
  std::vector<double> m_single_x;  // the vector m_single_ll belongs to
  const float *m_row;
  // Distributions views, created on first use
  PDFPool m_pool_view;
  std::vector<std::unique_ptr<Gaussian>> m_gauss_views;
  std::vector<std::unique_ptr<Mixture>> m_mix_views;
};

}  // namespace aku

#endif
