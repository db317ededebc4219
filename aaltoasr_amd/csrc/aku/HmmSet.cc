// HmmSet.cc -- see HmmSet.hh.
#include "HmmSet.hh"
#include "str.hh"

#include <cmath>
#include <cstdio>

namespace aku {

HmmSet::HmmSet()
    : m_gmm(nullptr), m_owner(nullptr), m_serial(0), m_first(0), m_count(0), m_row(nullptr) {}

HmmSet::~HmmSet() { drop_model(); }

void HmmSet::drop_model() {
  m_gauss_views.clear();
  m_mix_views.clear();
  m_single_x.clear();
  m_pool_x.clear();
  if (m_gmm) aasr_gmm_destroy(m_gmm);
  m_gmm = nullptr;
  m_owner = nullptr;
  m_count = 0;
  m_row = nullptr;
}

static bool readable(const std::string &p) {
  FILE *f = fopen(p.c_str(), "r");
  if (!f) return false;
  fclose(f);
  return true;
}

void HmmSet::read_gk(const std::string &filename) {
  if (!readable(filename))
    throw std::string("PDFPool::read_gk(): could not open ") + filename + "\n";
  m_gk = filename;
  drop_model();
}

void HmmSet::read_mc(const std::string &filename) {
  if (!readable(filename)) {
    fprintf(stderr, "HmmSet::read_mc(): could not open %s\n", filename.c_str());
    throw OpenError();
  }
  m_mc = filename;
  drop_model();
}

bool HmmSet::read_ph(const std::string &filename) {
  if (!readable(filename)) {
    fprintf(stderr, "HmmSet::read_ph(): could not open %s\n", filename.c_str());
    throw OpenError();
  }
  m_ph = filename;
  drop_model();
  // the topology (aku/HmmSet.cc:183-205): "PHONE" first, anything else is a ReadError
  std::ifstream in(filename.c_str());
  std::string word;
  in >> word;
  if (word != "PHONE") throw ReadError();
  m_hmm_map.clear();
  m_hmms.clear();
  m_states.clear();
  m_transitions.clear();
  read_legacy_ph(in);
  return true;  // legacy PHONE format is the only one the reference reads too
}

// aku/HmmSet.cc:208-329.  Per phone: "index states label", the two dummy states' numbers, one pdf
// index per real state, then per source (dummies included) "source n" and n "target prob" pairs.
// States are tied by their pdf: the first phone that mentions a pdf defines that state's
// transitions (target 1 = the sink, stored as the offset that leaves the HMM), later mentions are
// only checked.  States are then created in pdf order, transitions numbered state by state.
void HmmSet::read_legacy_ph(std::ifstream &in) {
  std::string label;
  int phonemes = 0;
  std::vector<std::vector<HmmTransition>> state_info;
  in >> phonemes;
  m_hmms.reserve(phonemes > 0 ? phonemes : 0);
  for (int h = 0; h < phonemes; h++) {
    int index = 0, states = 0;
    in >> index >> states >> label;
    if (!in) throw ReadError();
    states -= 2;  // the dummy entry / exit states
    Hmm &hmm = add_hmm(label, states);
    int dummy, pdf;
    std::vector<bool> load_transitions;
    in >> dummy >> dummy;
    for (int s = 0; s < states; s++) {
      in >> pdf;
      // (the reference indexes with whatever it read: a negative or garbage index is a ReadError here)
      if (!in || pdf < 0 || pdf > (1 << 24)) throw ReadError();
      if (pdf >= (int)state_info.size()) state_info.resize((size_t)pdf + 1);
      hmm.state(s) = pdf;
      load_transitions.push_back(state_info[(size_t)pdf].empty());
    }
    for (int s = -2; s < states; s++) {
      int transitions = 0, source = 0;
      in >> source >> transitions;
      source -= 2;
      if (source >= states)
        throw str::fmt(128, "HmmSet::read_legacy_ph: Invalid source state number %i (only %i states)", source,
                       states);
      for (int t = 0; t < transitions; t++) {
        int target;
        double prob;
        in >> target >> prob;
        if (prob <= 0)
          throw str::fmt(128,
                         "HmmSet::read_legacy_ph: Phone %i (%s) transition from %i to %i has nonpositive "
                         "probability %f.",
                         index, label.c_str(), source, target, prob);
        if (source >= 0 && load_transitions[(size_t)source]) {
          if (target == 1) {
            target = states - source;  // the sink
          } else {
            target -= 2;
            if (target > states)
              throw str::fmt(128, "HmmSet::read_legacy_ph: Invalid target state number %i (only %i states)",
                             source, states);
            target -= source;  // relative
          }
          state_info[(size_t)hmm.state(source)].push_back(HmmTransition(hmm.state(source), target, prob));
        }
      }
      if (source >= 0 && !load_transitions[(size_t)source])
        for (const HmmTransition &tr : state_info[(size_t)hmm.state(source)])
          if (source + tr.target_offset > states)
            throw str::fmt(128,
                           "HmmSet::read_legacy_ph: Invalid target state number %i on existing state %i (only "
                           "%i states)",
                           source, hmm.state(source), states);
    }
  }
  for (int s = 0; s < (int)state_info.size(); s++) {
    add_state(s);
    for (const HmmTransition &tr : state_info[(size_t)s]) add_transition(s, tr.target_offset, tr.prob);
  }
}

std::string Hmm::get_center_phone() {
  const size_t minus = label.find_last_of('-'), plus = label.find_first_of('+');
  std::string c;
  if (minus != std::string::npos && plus != std::string::npos) {
    if (plus > minus + 1) c = label.substr(minus + 1, plus - minus - 1);
  } else if (minus != std::string::npos) {
    c = label.substr(minus + 1);
  } else if (plus != std::string::npos) {
    c = label.substr(0, plus);
  } else {
    c = label;
  }
  if (c.empty()) throw std::string("Invalid phone label ") + label;
  return c;
}

Hmm &HmmSet::new_hmm(const std::string &label) {
  if (m_hmm_map.count(label)) throw DuplicateHmm();
  m_hmm_map[label] = (int)m_hmms.size();
  m_hmms.push_back(Hmm());
  m_hmms.back().label = label;
  return m_hmms.back();
}

Hmm &HmmSet::add_hmm(const std::string &label, int num_states) {
  Hmm &h = new_hmm(label);
  h.resize(num_states);
  return h;
}

int HmmSet::hmm_index(const std::string &label) const {
  const auto it = m_hmm_map.find(label);
  if (it == m_hmm_map.end()) {
    fprintf(stderr, "HmmSet::hmm_index(): unknown hmm '%s'\n", label.c_str());
    throw UnknownHmm();
  }
  return it->second;
}

int HmmSet::add_transition(int source, int target, double prob) {
  const int index = (int)m_transitions.size();
  m_transitions.push_back(HmmTransition(source, target, prob));
  m_states[(size_t)source].m_transitions.push_back(index);
  return index;
}

int HmmSet::add_state(int pdf_index) {
  m_states.push_back(HmmState(pdf_index));
  return (int)m_states.size() - 1;
}

void HmmSet::read_all(const std::string &base) {
  // same order as aku/HmmSet.cc:351-357
  read_mc(base + ".mc");
  read_ph(base + ".ph");
  read_gk(base + ".gk");
  ensure_model();
}

void HmmSet::ensure_model() {
  if (m_gmm) return;
  if (m_gk.empty() || m_mc.empty())
    throw std::string("HmmSet: model files have not been read");
  aasr_status st = aasr_gmm_create_from_files(m_gk.c_str(), m_mc.c_str(),
                                              m_ph.empty() ? nullptr : m_ph.c_str(), &m_gmm);
  if (st == AASR_ERR_IO) throw OpenError();
  if (st != AASR_OK) {
    std::string msg = aasr_last_error();
    if (msg.find("read_ph") != std::string::npos) throw ReadError();
    throw msg;
  }
}

void HmmSet::read_clustering(const std::string &filename) {
  ensure_model();
  // PDFPool::read_clustering throws std::string (aku/Distributions.cc:3118-3140)
  if (aasr_gmm_read_clustering(m_gmm, filename.c_str()) != AASR_OK)
    throw std::string(aasr_last_error());
  m_count = 0;  // cached block rows were scored without the clustering
  m_row = nullptr;
  m_row64 = nullptr;
}

void HmmSet::set_clustering_min_evals(double min_clusters, double min_gaussians) {
  ensure_model();
  if (aasr_gmm_set_clustering_min_evals(m_gmm, min_clusters, min_gaussians) != AASR_OK)
    throw std::string(aasr_last_error());
  m_count = 0;
  m_row = nullptr;
  m_row64 = nullptr;
}

int HmmSet::dim() {
  ensure_model();
  return aasr_gmm_dim(m_gmm);
}

int HmmSet::num_states() {
  ensure_model();
  return aasr_gmm_num_states(m_gmm);
}

void HmmSet::reset_cache() {
  m_row = nullptr;
  m_row64 = nullptr;
  for (ResetCacheInterface *o : m_reset_cache_objects) o->reset_cache();
}

int HmmSet::num_pool_pdfs() {
  ensure_model();
  return aasr_gmm_num_gaussians(m_gmm);
}

double HmmSet::pool_log_likelihood(const int g, const FeatureVec &f) {
  return pool_log_likelihood(g, f.data(), f.dim());
}

double HmmSet::pool_log_likelihood(const int g, const double *px, int dim) {
  ensure_model();
  const int G = aasr_gmm_num_gaussians(m_gmm);
  if (g < 0 || g >= G) throw std::string("PDFPool: Gaussian index out of range");
  std::vector<double> x(px, px + dim);
  if (x != m_pool_x || (int)m_pool_ll.size() != G) {
    if (dim != aasr_gmm_dim(m_gmm))
      throw std::string("HmmSet: feature dimension does not match the model");
    std::vector<float> xf(x.begin(), x.end());
    m_pool_ll.resize((size_t)G);
    if (aasr_gmm_gauss_loglik(m_gmm, xf.data(), 1, m_pool_ll.data()) != AASR_OK)
      throw std::string(aasr_last_error());
    m_pool_x = x;
  }
  return (double)m_pool_ll[(size_t)g];
}

double HmmSet::pool_likelihood(const int g, const FeatureVec &f) {
  return std::exp(pool_log_likelihood(g, f));
}

const float *HmmSet::state_loglik_row(const FeatureVec &f) {
  ensure_model();
  const int S = aasr_gmm_num_states(m_gmm);
  const FeatureGenerator *own = f.owner();
  if (own) {
    int first = 0, count = 0;
    const float *blk = own->block_f32(f.frame(), &first, &count);
    if (blk) {
      if (m_owner != own || m_serial != own->block_serial() || m_count == 0) {
        m_block_ll.resize((size_t)count * S);
        int f1 = 0, c1 = 0;
        const double *blk64 = aasr_gmm_get_precision(m_gmm) == AASR_PREC_F64 ? own->block_f64(f.frame(), &f1, &c1) : nullptr;
        if (blk64) {  // double features in, double log-likelihoods out; the float rows are their roundings
          m_block_ll64.resize((size_t)count * S);
          if (aasr_gmm_score_f64(m_gmm, blk64, count, m_block_ll64.data()) != AASR_OK)
            throw std::string(aasr_last_error());
          for (size_t i = 0; i < m_block_ll64.size(); i++) m_block_ll[i] = (float)m_block_ll64[i];
        } else {
          m_block_ll64.clear();
          if (aasr_gmm_score(m_gmm, blk, count, m_block_ll.data()) != AASR_OK)
            throw std::string(aasr_last_error());
        }
        m_owner = own;
        m_serial = own->block_serial();
        m_first = first;
        m_count = count;
      }
      return &m_block_ll[(size_t)(f.frame() - m_first) * S];
    }
  }
  return state_loglik_row(f.data(), f.dim());
}

const float *HmmSet::state_loglik_row(const double *px, int dim) {
  ensure_model();
  const int S = aasr_gmm_num_states(m_gmm);
  // a Vector handed out by a generator (FeatureVec::get_vector()) still lies in its block
  int frame = 0;
  const FeatureGenerator *own = FeatureGenerator::find_block(px, &frame);
  if (own) return state_loglik_row(FeatureVec(px, dim, frame, own));
  // a vector that does not come from a cached block: score it alone, keep it until the next one
  if (dim != aasr_gmm_dim(m_gmm))
    throw std::string("HmmSet: feature dimension does not match the model");
  std::vector<double> xd(px, px + dim);
  if (xd == m_single_x && (int)m_single_ll.size() == S) return m_single_ll.data();
  std::vector<float> x(xd.begin(), xd.end());
  m_single_ll.resize(S);
  if (aasr_gmm_get_precision(m_gmm) == AASR_PREC_F64) {
    m_single_ll64.resize(S);
    if (aasr_gmm_score_f64(m_gmm, xd.data(), 1, m_single_ll64.data()) != AASR_OK)
      throw std::string(aasr_last_error());
    for (int i = 0; i < S; i++) m_single_ll[(size_t)i] = (float)m_single_ll64[(size_t)i];
  } else {
    m_single_ll64.clear();
    if (aasr_gmm_score(m_gmm, x.data(), 1, m_single_ll.data()) != AASR_OK)
      throw std::string(aasr_last_error());
  }
  m_single_x = xd;
  return m_single_ll.data();
}

// ---- Distributions views ----------------------------------------------------------------

PDFPool *HmmSet::get_pool() {
  ensure_model();
  m_pool_view.m_set = this;
  return &m_pool_view;
}

PDF *HmmSet::get_pool_pdf(int index) {
  ensure_model();
  const int G = aasr_gmm_num_gaussians(m_gmm);
  if (index < 0 || index >= G) throw std::string("PDFPool: Gaussian index out of range");
  if ((int)m_gauss_views.size() != G) {
    m_gauss_views.clear();
    m_gauss_views.resize((size_t)G);
  }
  std::unique_ptr<Gaussian> &v = m_gauss_views[(size_t)index];
  if (!v) {
    v.reset(new Gaussian());
    v->m_set = this;
    v->m_index = index;
  }
  return v.get();
}

Mixture *HmmSet::get_emission_pdf(int index) {
  ensure_model();
  const int S = aasr_gmm_num_states(m_gmm);
  if (index < 0 || index >= S) throw std::string("HmmSet: state index out of range");
  if ((int)m_mix_views.size() != S) {
    m_mix_views.clear();
    m_mix_views.resize((size_t)S);
  }
  std::unique_ptr<Mixture> &v = m_mix_views[(size_t)index];
  if (!v) {
    v.reset(new Mixture());
    v->m_set = this;
    v->m_index = index;
    const int n = aasr_gmm_mixture_size(m_gmm, index);
    std::vector<int32_t> idx((size_t)n);
    v->m_weights.resize((size_t)n);
    if (aasr_gmm_mixture_get(m_gmm, index, idx.data(), v->m_weights.data()) != AASR_OK)
      throw std::string(aasr_last_error());
    v->m_pointers.assign(idx.begin(), idx.end());
  }
  return v.get();
}

int PDF::dim() const { return m_set->dim(); }

double Gaussian::compute_log_likelihood(const Vector &f) const {
  return m_set->pool_log_likelihood(m_index, f.addr(), f.size());
}
double Gaussian::compute_likelihood(const Vector &f) const { return std::exp(compute_log_likelihood(f)); }

void Gaussian::get_mean(Vector &mean) const {
  mean.resize(m_set->dim());
  if (aasr_gmm_gaussian_get(m_set->handle(), m_index, mean.addr(), nullptr) != AASR_OK)
    throw std::string(aasr_last_error());
}
void Gaussian::get_covariance(Vector &covariance) const {
  covariance.resize(m_set->dim());
  if (aasr_gmm_gaussian_get(m_set->handle(), m_index, nullptr, covariance.addr()) != AASR_OK)
    throw std::string(aasr_last_error());
}

int PDFPool::size() const { return m_set->num_pool_pdfs(); }
int PDFPool::dim() const { return m_set->dim(); }
PDF *PDFPool::get_pdf(int index) const { return m_set->get_pool_pdf(index); }
double PDFPool::compute_likelihood(const Vector &f, int index) {
  return std::exp(m_set->pool_log_likelihood(index, f.addr(), f.size()));
}
void PDFPool::precompute_likelihoods(const Vector &f) { (void)m_set->pool_log_likelihood(0, f.addr(), f.size()); }

PDF *Mixture::get_base_pdf(int index) { return m_set->get_pool_pdf(m_pointers[(size_t)index]); }
double Mixture::compute_likelihood(const Vector &f) const {
  return std::exp((double)m_set->state_loglik_row(f.addr(), f.size())[m_index]);
}
double Mixture::compute_log_likelihood(const Vector &f) const { return util::safe_log(compute_likelihood(f)); }

// the double row that belongs to a float row handed out by state_loglik_row (AASR_PREC_F64), or null
const double *HmmSet::row64_for(const float *row) {
  if (!row) return nullptr;
  if (!m_block_ll64.empty() && row >= m_block_ll.data() && row < m_block_ll.data() + m_block_ll.size())
    return m_block_ll64.data() + (row - m_block_ll.data());
  if (!m_single_ll64.empty() && row == m_single_ll.data()) return m_single_ll64.data();
  return nullptr;
}

void HmmSet::precompute_likelihoods(const FeatureVec &f) {
  reset_cache();
  m_row = state_loglik_row(f);
  m_row64 = row64_for(m_row);
}

double HmmSet::state_likelihood(const int s, const FeatureVec &f) {
  if (!m_row) {  // lazy style: reset_cache() then single states
    m_row = state_loglik_row(f);
    m_row64 = row64_for(m_row);
  }
  if (s < 0 || s >= aasr_gmm_num_states(m_gmm)) throw std::string("HmmSet: state index out of range");
  // the row holds log(max(lik, 1e-50)) (aku/HmmSet.cc:497-498)
  return std::exp(m_row64 ? m_row64[s] : (double)m_row[s]);
}

}  // namespace aku
