// gmm_model.cc -- model file readers and host-side packing of the streamed
// Gaussian operand.
//
// File formats: PDFPool::read_gk (aku/Distributions.cc:2811-2910),
// DiagonalGaussian::read (:1131-1150), HmmSet::read_mc (aku/HmmSet.cc:156-180),
// Mixture::read (aku/Distributions.cc:2418-2434), HmmSet::read_legacy_ph
// (aku/HmmSet.cc:194-329).  Constants: DiagonalGaussian::set_constant
// (aku/Distributions.cc:1273-1288) -- no (2*pi)^(-d/2) term.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "gmm.h"

namespace aasr {

static const double kLog2e = 1.4426950408889634073599246810019;
// log2-domain value standing in for log(0): exp2(x - max) underflows to 0 for
// any live component, and an all-null segment still reduces to a finite value
// that the 1e-50 floor then clamps.
static const float kNullConst = -1.0e30f;

HostModel read_model_files(const char *gk, const char *mc, const char *ph) {
  HostModel m;
  {
    std::ifstream in(gk);
    if (!in) raise(AASR_ERR_IO, "PDFPool::read_gk(): could not open %s", gk);
    long pdfs = 0;
    std::string type;
    in >> pdfs >> m.dim >> type;
    if (!in || pdfs < 0 || m.dim <= 0)
      raise(AASR_ERR_INVALID, "PDFPool::read_gk(): error reading file: %s", gk);
    bool variable = (type == "variable");
    if (!variable && type != "diagonal_cov") {
      if (type == "full_cov" || type == "pcgmm" || type == "scgmm")
        raise(AASR_ERR_UNSUPPORTED,
              "gk type '%s' (non-diagonal Gaussians) is not built in this engine yet", type.c_str());
      raise(AASR_ERR_INVALID, "Unknown model type");
    }
    m.G = pdfs;
    m.mean.resize((size_t)pdfs * m.dim);
    m.var.resize((size_t)pdfs * m.dim);
    for (long g = 0; g < pdfs; g++) {
      if (variable) {
        in >> type;
        if (type != "diag") {
          if (type == "full" || type == "pcgmm" || type == "scgmm" ||
              type == "precision_subspace" || type == "exponential_subspace")
            raise(AASR_ERR_UNSUPPORTED,
                  "Gaussian type '%s' is not built in this engine yet", type.c_str());
          raise(AASR_ERR_INVALID, "Unknown model type\n%s", type.c_str());
        }
      }
      for (int i = 0; i < m.dim; i++) in >> m.mean[(size_t)g * m.dim + i];
      for (int i = 0; i < m.dim; i++) in >> m.var[(size_t)g * m.dim + i];
      if (in.fail())
        raise(AASR_ERR_INVALID, "Error in reading Gaussian specifications");
    }
  }
  {
    std::ifstream in(mc);
    if (!in) raise(AASR_ERR_IO, "HmmSet::read_mc(): could not open %s", mc);
    long pdfs = 0;
    in >> pdfs;
    if (!in || pdfs < 0) raise(AASR_ERR_INVALID, "HmmSet::read_mc(): bad header in %s", mc);
    m.S = pdfs;
    m.mix_off.assign(1, 0);
    for (long s = 0; s < pdfs; s++) {
      int n = 0;
      in >> n;
      for (int k = 0; k < n; k++) {
        int idx;
        double w;
        in >> idx >> w;
        if (in.fail())
          raise(AASR_ERR_INVALID, "Error in reading mixture specifications");
        m.mix_idx.push_back(idx);
        m.mix_w.push_back(w);
      }
      m.mix_off.push_back((int32_t)m.mix_idx.size());
    }
  }
  if (ph) {
    // Legacy PHONE file: only the state inventory matters for scoring.  State
    // index == emission pdf index (aku/HmmSet.cc:245,319-322); the number of
    // states is 1 + the largest pdf index referenced.
    std::ifstream in(ph);
    if (!in) raise(AASR_ERR_IO, "HmmSet::read_ph(): could not open %s", ph);
    std::string buf;
    in >> buf;
    if (buf != "PHONE") raise(AASR_ERR_INVALID, "HmmSet::read_ph(): not a PHONE file: %s", ph);
    int phonemes = 0;
    in >> phonemes;
    long max_pdf = -1;
    for (int h = 0; h < phonemes; h++) {
      int index = 0, states = 0;
      std::string label;
      in >> index >> states >> label;
      if (!in) raise(AASR_ERR_INVALID, "HmmSet::read_ph(): read error in %s", ph);
      states -= 2;
      int dummy;
      in >> dummy >> dummy;
      for (int s = 0; s < states; s++) {
        int pdf;
        in >> pdf;
        if (pdf > max_pdf) max_pdf = pdf;
      }
      for (int s = -2; s < states; s++) {
        int source = 0, transitions = 0;
        in >> source >> transitions;
        for (int t = 0; t < transitions; t++) {
          int target;
          double prob;
          in >> target >> prob;
        }
      }
      if (!in) raise(AASR_ERR_INVALID, "HmmSet::read_ph(): read error in %s", ph);
    }
    long nstates = max_pdf + 1;
    if (nstates > m.S)
      raise(AASR_ERR_INVALID, "ph file references pdf %ld but mc file has %ld mixtures", max_pdf, (long)m.S);
    // states beyond the ph inventory are not emitted (num_states() = ph count)
    if (nstates < m.S) {
      m.S = nstates;
      m.mix_off.resize((size_t)nstates + 1);
      m.mix_idx.resize((size_t)m.mix_off.back());
      m.mix_w.resize((size_t)m.mix_off.back());
    }
  }
  return m;
}

// ---------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------

struct RowSpec {
  int64_t g;        // pool Gaussian
  double logw;      // log mixture weight (natural), -inf for zero weight
};

// Write rows into the [tiles][nkk/2][64][4] layout the kernel streams:
// lane l = h*32 + r32 of kk-pair q holds
//   { A[r32][2(2q)+h], A[r32][2(2q+1)+h], A[32+r32][2(2q)+h], A[32+r32][2(2q+1)+h] }
// with K index k = 2*kk + h:  kk<dim: h=0 -> p*mu'*log2e, h=1 -> -p/2*log2e;
// kk==dim: h=0 -> constant*log2e; everything else 0.
static void pack_rows(const aasr_gmm *g, const std::vector<RowSpec> &rows,
                      PackedRows &out, std::vector<double> *a64_host) {
  const HostModel &m = g->host;
  const int D = m.dim;
  const int nkk = pick_nkk(D);
  if (nkk < 0)
    raise(AASR_ERR_UNSUPPORTED, "feature dimension %d > 63 is not built in this engine yet", D);
  out.nkk = nkk;
  out.rows = (int64_t)rows.size();
  out.tiles = std::max<int64_t>(1, (out.rows + TILE_ROWS - 1) / TILE_ROWS);
  const size_t tile_floats = (size_t)(nkk / 2) * 64 * 4;
  std::vector<float> a((size_t)out.tiles * tile_floats, 0.0f);
  if (a64_host) a64_host->assign((size_t)out.tiles * TILE_ROWS * (2 * D + 1), 0.0);
  std::vector<double> coef(2 * (size_t)nkk);
  for (int64_t r = 0; r < out.tiles * TILE_ROWS; r++) {
    std::fill(coef.begin(), coef.end(), 0.0);
    if (r < out.rows) {
      const RowSpec &rs = rows[(size_t)r];
      const double *mu = &m.mean[(size_t)rs.g * D];
      const double *var = &m.var[(size_t)rs.g * D];
      // DiagonalGaussian::read + set_constant (Distributions.cc:1144-1147, 1273-1288)
      double prod = 1, quad = 0;
      for (int d = 0; d < D; d++) {
        double p = (var[d] > 0) ? 1 / var[d] : 0;
        prod *= p;
      }
      double cst = (prod > 0) ? std::log(std::sqrt(prod)) : prod;
      for (int d = 0; d < D; d++) {
        double p = (var[d] > 0) ? 1 / var[d] : 0;
        double muc = mu[d] - (double)g->pivot[d];
        coef[2 * d] = p * muc * kLog2e;
        coef[2 * d + 1] = -0.5 * p * kLog2e;
        quad += p * muc * muc;
      }
      double c = cst + rs.logw - 0.5 * quad;
      if (!std::isfinite(c)) {
        if (c > 0 || std::isnan(c))
          raise(AASR_ERR_INVALID,
                "Gaussian %ld has a non-finite constant (precision product overflow)", (long)rs.g);
        coef[2 * D] = kNullConst;
      } else {
        coef[2 * D] = c * kLog2e;
      }
    } else {
      coef[2 * D] = kNullConst;  // padding row: contributes exp2(-1e30 - max) = 0
    }
    int64_t t = r / TILE_ROWS;
    int j = (int)(r % TILE_ROWS);
    int mb = j / 32, r32 = j % 32;
    for (int kk = 0; kk < nkk; kk++) {
      int q = kk / 2, e = kk % 2;
      for (int h = 0; h < 2; h++) {
        size_t idx = (size_t)t * tile_floats + ((size_t)q * 64 + (size_t)(h * 32 + r32)) * 4 + (size_t)(mb * 2 + e);
        a[idx] = (float)coef[2 * kk + h];
      }
    }
    if (a64_host)
      for (int k = 0; k < 2 * D + 1; k++)
        (*a64_host)[(size_t)r * (2 * D + 1) + k] = coef[k];
  }
  out.a.upload(a.data(), a.size());
}

void gmm_build(aasr_gmm *g, const HostModel &model) {
  require_device();
  g->host = model;
  HostModel &m = g->host;
  g->dim = m.dim;
  g->G = m.G;
  g->S = m.S;
  if (m.dim <= 0 || m.G <= 0 || m.S <= 0)
    raise(AASR_ERR_INVALID, "empty model (dim %d, %ld Gaussians, %ld states)", m.dim, (long)m.G, (long)m.S);
  if ((int64_t)m.mix_off.size() != m.S + 1)
    raise(AASR_ERR_INVALID, "mix_off must hold num_states+1 entries");
  if (m.dim + 1 > 64)
    raise(AASR_ERR_UNSUPPORTED, "feature dimension %d > 63 is not built in this engine yet", m.dim);
  for (size_t k = 0; k < m.mix_idx.size(); k++)
    if (m.mix_idx[k] < 0 || m.mix_idx[k] >= m.G)
      raise(AASR_ERR_INVALID, "mixture component %zu points at Gaussian %d outside the pool of %ld",
            k, m.mix_idx[k], (long)m.G);
  // Mixture::normalize_weights (Distributions.cc:2067-2075)
  for (int64_t s = 0; s < m.S; s++) {
    double sum = 0;
    for (int32_t k = m.mix_off[s]; k < m.mix_off[s + 1]; k++) sum += m.mix_w[k];
    for (int32_t k = m.mix_off[s]; k < m.mix_off[s + 1]; k++) m.mix_w[k] /= sum;
  }
  // centring pivot: per-dimension mean of the pool means, rounded to float so
  // the device subtracts exactly the value the constants were built with
  g->pivot.assign(m.dim, 0.0f);
  for (int d = 0; d < m.dim; d++) {
    double acc = 0;
    for (int64_t i = 0; i < m.G; i++) acc += m.mean[(size_t)i * m.dim + d];
    g->pivot[d] = (float)(acc / (double)m.G);
  }
  g->d_pivot.upload(g->pivot.data(), g->pivot.size());

  // component-expanded rows in state order + segment metadata
  std::vector<RowSpec> rows;
  rows.reserve(m.mix_idx.size());
  std::vector<int32_t> chunk_seg_begin;
  std::vector<uint32_t> seg_desc;
  std::vector<int32_t> seg_out;
  int64_t total_rows = (int64_t)m.mix_idx.size();
  int64_t n_chunks = std::max<int64_t>(1, (total_rows + TILE_ROWS - 1) / TILE_ROWS) * (TILE_ROWS / CHUNK_ROWS);
  std::vector<std::vector<std::pair<uint32_t, int32_t>>> per_chunk((size_t)n_chunks);
  int64_t row = 0;
  for (int64_t s = 0; s < m.S; s++) {
    int32_t a = m.mix_off[s], b = m.mix_off[s + 1];
    if (b <= a) {
      // a state without components scores the floor; emit a zero-length
      // closing segment so the column is still written
      int64_t c = std::min<int64_t>(row / CHUNK_ROWS, n_chunks - 1);
      uint32_t rb = (uint32_t)(row - c * CHUNK_ROWS);
      if (rb > CHUNK_ROWS) rb = CHUNK_ROWS;
      per_chunk[(size_t)c].push_back({rb | (rb << 8), (int32_t)s});
      continue;
    }
    for (int32_t k = a; k < b; k++) {
      double w = m.mix_w[k];
      rows.push_back({m.mix_idx[k], (w > 0) ? std::log(w) : -INFINITY});
    }
    int64_t r0 = row, r1 = row + (b - a);
    for (int64_t c = r0 / CHUNK_ROWS; c * CHUNK_ROWS < r1; c++) {
      int64_t lo = std::max(r0, c * CHUNK_ROWS), hi = std::min(r1, (c + 1) * CHUNK_ROWS);
      uint32_t desc = (uint32_t)(lo - c * CHUNK_ROWS) | ((uint32_t)(hi - c * CHUNK_ROWS) << 8);
      if (lo > r0) desc |= 1u << 16;  // continues a segment opened in an earlier chunk
      if (hi < r1) desc |= 1u << 17;  // stays open into the next chunk
      per_chunk[(size_t)c].push_back({desc, (int32_t)s});
    }
    row = r1;
  }
  chunk_seg_begin.push_back(0);
  for (auto &v : per_chunk) {
    for (auto &p : v) {
      seg_desc.push_back(p.first);
      seg_out.push_back(p.second);
    }
    chunk_seg_begin.push_back((int32_t)seg_desc.size());
  }
  pack_rows(g, rows, g->mix, nullptr);
  g->mix.chunk_seg_begin.upload(chunk_seg_begin.data(), chunk_seg_begin.size());
  g->mix.seg_desc.upload(seg_desc.data(), seg_desc.size());
  g->mix.seg_out.upload(seg_out.data(), seg_out.size());
}

void gmm_build_pool(aasr_gmm *g) {
  if (g->pool_built) return;
  std::vector<RowSpec> rows((size_t)g->G);
  for (int64_t i = 0; i < g->G; i++) rows[(size_t)i] = {i, 0.0};
  pack_rows(g, rows, g->pool, nullptr);
  g->pool_built = true;
}

}  // namespace aasr
