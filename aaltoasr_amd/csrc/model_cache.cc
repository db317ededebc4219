// model_cache.cc -- binary cache of a parsed acoustic model.
//
// The reference parses .gk/.mc/.ph text on every start (aku/HmmSet.cc:351-357,
// aku/Distributions.cc:2811-2910); for 50 000 x 78 numbers that is seconds per
// process, and one process per GPU each pays it.  The cache holds exactly what
// the parser produced (double precision, mixture weights already normalised
// like Mixture::read does), so a model built from it scores bit-identically.
//
// A cache also records which text files it was written from -- (size, FNV-1a content hash) of
// the .gk, .mc and .ph -- and read_model_cache_checked refuses a cache whose record does not
// match the files the caller names: after retraining, or with another -b next to the same cache
// path, the text files are parsed again instead of scoring with a stale model.
//
// Layout (little endian): "AASRGMM1", u32 version (3), i32 dim, i64 G, i64 S, i64 K,
// u32 flags (bit 0: covariances present, bit 1: source fingerprint present, bit 2: per-Gaussian
// constant offsets present), i64 number of HMMs,
// u64 fingerprint[6], then mean[G*dim],
// var[G*dim] (f64), [cov[G*dim*dim] f64, is_full[G] u8], [gauss_bias[G] f64], mix_off[S+1] i32,
// mix_idx[K] i32, mix_w[K] f64, per HMM {u32 label length, label, u32 states,
// i32 pdf[states]}, and a 64-bit FNV-1a checksum of everything before it.
#include <cstdio>
#include <cstring>
#include <fstream>

#include "gmm.h"

namespace aasr {

namespace {

const char kMagic[8] = {'A', 'A', 'S', 'R', 'G', 'M', 'M', '1'};
const uint32_t kVersion = 3;

uint64_t fnv1a(const char *b, size_t n, uint64_t h = 1469598103934665603ull) {
  for (size_t i = 0; i < n; i++) {
    h ^= (uint8_t)b[i];
    h *= 1099511628211ull;
  }
  return h;
}
uint64_t fnv1a(const std::vector<char> &b, size_t n) { return fnv1a(b.data(), n); }

// (size, content hash) of one file; a missing file gives (0, 0)
void file_fingerprint(const char *path, uint64_t *size, uint64_t *hash) {
  *size = 0;
  *hash = 0;
  if (!path || !*path) return;
  FILE *fp = fopen(path, "rb");
  if (!fp) return;
  std::vector<char> buf(1 << 20);
  uint64_t h = 1469598103934665603ull, n = 0;
  size_t got;
  while ((got = fread(buf.data(), 1, buf.size(), fp)) > 0) {
    h = fnv1a(buf.data(), got, h);
    n += got;
  }
  fclose(fp);
  *size = n;
  *hash = h;
}

struct Writer {
  std::vector<char> b;
  void raw(const void *p, size_t n) {
    const char *c = (const char *)p;
    b.insert(b.end(), c, c + n);
  }
  template <class T>
  void put(T v) { raw(&v, sizeof v); }
  template <class T>
  void arr(const std::vector<T> &v) { if (!v.empty()) raw(v.data(), v.size() * sizeof(T)); }
};

struct Reader {
  const std::vector<char> &b;
  size_t pos = 0, end;
  const char *path;
  void need(size_t n) {
    if (pos + n > end) raise(AASR_ERR_INVALID, "%s: truncated model cache", path);
  }
  template <class T>
  T get() {
    T v;
    need(sizeof v);
    memcpy(&v, b.data() + pos, sizeof v);
    pos += sizeof v;
    return v;
  }
  template <class T>
  void arr(std::vector<T> &v, size_t n) {
    if (n > (end - pos) / sizeof(T)) raise(AASR_ERR_INVALID, "%s: truncated model cache", path);
    v.resize(n);
    if (n) memcpy(v.data(), b.data() + pos, n * sizeof(T));
    pos += n * sizeof(T);
  }
};

}  // namespace

void model_files_fingerprint(const char *gk, const char *mc, const char *ph, uint64_t fp[6]) {
  file_fingerprint(gk, &fp[0], &fp[1]);
  file_fingerprint(mc, &fp[2], &fp[3]);
  file_fingerprint(ph, &fp[4], &fp[5]);
}

void write_model_cache(const HostModel &m, const char *path) {
  if (m.n_transforms > 0) raise(AASR_ERR_INVALID, "an adapted model is not cached (remove the transform first)");
  if (!m.weights_normalized) raise(AASR_ERR_INVALID, "model cache: the model has not been built yet");
  Writer w;
  w.raw(kMagic, 8);
  w.put<uint32_t>(kVersion);
  w.put<int32_t>(m.dim);
  w.put<int64_t>(m.G);
  w.put<int64_t>(m.S);
  w.put<int64_t>((int64_t)m.mix_idx.size());
  w.put<uint32_t>((m.any_full() ? 1u : 0u) | (m.has_src_fp ? 2u : 0u) | (m.gauss_bias.empty() ? 0u : 4u));
  w.put<int64_t>((int64_t)m.hmm_label.size());
  for (int i = 0; i < 6; i++) w.put<uint64_t>(m.has_src_fp ? m.src_fp[i] : 0ull);
  w.arr(m.mean);
  w.arr(m.var);
  if (m.any_full()) {
    w.arr(m.cov);
    w.arr(m.is_full);
  }
  if (!m.gauss_bias.empty()) w.arr(m.gauss_bias);
  w.arr(m.mix_off);
  w.arr(m.mix_idx);
  w.arr(m.mix_w);
  for (size_t h = 0; h < m.hmm_label.size(); h++) {
    w.put<uint32_t>((uint32_t)m.hmm_label[h].size());
    w.raw(m.hmm_label[h].data(), m.hmm_label[h].size());
    w.put<uint32_t>((uint32_t)m.hmm_states[h].size());
    w.arr(m.hmm_states[h]);
  }
  w.put<uint64_t>(fnv1a(w.b, w.b.size()));
  // temporary name + rename: concurrent ranks never see a half-written cache
  const std::string tmp = std::string(path) + ".tmp";
  {
    std::ofstream out(tmp, std::ios::binary);
    if (!out) raise(AASR_ERR_IO, "could not open %s for writing", tmp.c_str());
    out.write(w.b.data(), (std::streamsize)w.b.size());
    if (!out) raise(AASR_ERR_IO, "Write error: %s", tmp.c_str());
  }
  if (rename(tmp.c_str(), path) != 0) {
    remove(tmp.c_str());
    raise(AASR_ERR_IO, "could not rename %s to %s", tmp.c_str(), path);
  }
}

HostModel read_model_cache(const char *path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) raise(AASR_ERR_IO, "could not open model cache %s", path);
  std::vector<char> b((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (b.size() < 8 + 4 + 4 + 8 * 4 + 4 + 6 * 8 + 8 || memcmp(b.data(), kMagic, 8) != 0)
    raise(AASR_ERR_INVALID, "%s is not a model cache", path);
  uint64_t want;
  memcpy(&want, b.data() + b.size() - 8, 8);
  if (fnv1a(b, b.size() - 8) != want) raise(AASR_ERR_INVALID, "%s: model cache checksum mismatch", path);
  Reader r{b, 8, b.size() - 8, path};
  if (r.get<uint32_t>() != kVersion)
    raise(AASR_ERR_INVALID, "%s: unknown model cache version (written by another build; delete it)", path);
  HostModel m;
  m.dim = r.get<int32_t>();
  m.G = r.get<int64_t>();
  m.S = r.get<int64_t>();
  const int64_t K = r.get<int64_t>();
  const uint32_t flags = r.get<uint32_t>();
  const int64_t n_hmm = r.get<int64_t>();
  if (m.dim <= 0 || m.G <= 0 || m.S <= 0 || K < 0 || n_hmm < 0)
    raise(AASR_ERR_INVALID, "%s: implausible model cache header", path);
  m.has_src_fp = (flags & 2u) != 0;
  for (int i = 0; i < 6; i++) m.src_fp[i] = r.get<uint64_t>();
  const size_t gd = (size_t)m.G * m.dim;
  r.arr(m.mean, gd);
  r.arr(m.var, gd);
  if (flags & 1u) {
    r.arr(m.cov, gd * m.dim);
    r.arr(m.is_full, (size_t)m.G);
  }
  if (flags & 4u) r.arr(m.gauss_bias, (size_t)m.G);
  r.arr(m.mix_off, (size_t)m.S + 1);
  r.arr(m.mix_idx, (size_t)K);
  r.arr(m.mix_w, (size_t)K);
  for (int64_t h = 0; h < n_hmm; h++) {
    const uint32_t len = r.get<uint32_t>();
    r.need(len);
    m.hmm_label.emplace_back(b.data() + r.pos, len);
    r.pos += len;
    const uint32_t ns = r.get<uint32_t>();
    m.hmm_states.emplace_back();
    r.arr(m.hmm_states.back(), ns);
  }
  if (r.pos != r.end) raise(AASR_ERR_INVALID, "%s: trailing bytes in model cache", path);
  // the checksum only catches accidental damage: the index arrays of a foreign or crafted file
  // must not send the builder out of range
  bool ok = m.mix_off[0] == 0 && (int64_t)m.mix_off[(size_t)m.S] == K;
  for (int64_t s = 0; ok && s < m.S; s++) ok = m.mix_off[(size_t)s] <= m.mix_off[(size_t)s + 1];
  for (int64_t k = 0; ok && k < K; k++) ok = m.mix_idx[(size_t)k] >= 0 && m.mix_idx[(size_t)k] < m.G;
  for (size_t h = 0; ok && h < m.hmm_states.size(); h++)
    for (int32_t pdf : m.hmm_states[h]) ok = ok && pdf >= 0 && pdf < m.S;
  if (!ok) raise(AASR_ERR_INVALID, "%s: inconsistent mixture tables in model cache", path);
  m.weights_normalized = true;
  return m;
}

HostModel read_model_cache_checked(const char *path, const char *gk, const char *mc, const char *ph) {
  HostModel m = read_model_cache(path);
  uint64_t fp[6];
  model_files_fingerprint(gk, mc, ph, fp);
  if (!m.has_src_fp || memcmp(fp, m.src_fp, sizeof fp) != 0)
    raise(AASR_ERR_INVALID, "%s: stale model cache (it was not written from %s / %s / %s as they are now)",
          path, gk ? gk : "-", mc ? mc : "-", ph ? ph : "-");
  return m;
}

}  // namespace aasr
