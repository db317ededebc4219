// feat_graph.cc -- parses an aku feature configuration and builds the device
// tables of each module.
//
// Reference behaviour restated here (paths relative to the AaltoASR tree):
//   ModuleConfig::read / get           aku/ModuleConfig.cc:15-202
//   FeatureGenerator::load_configuration  aku/FeatureGenerator.cc:96-219
//   *Module::set_module_config         aku/FeatureModules.cc (cited per module)
// Tables that the reference derives with libm float functions (Hamming window,
// mel edges, DCT cosines, KissFFT twiddles) are computed HERE on the host with
// the same expressions, so the device never evaluates cosf/log10f/pow itself.
#include <climits>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "feat.h"

namespace aasr {

// ------------------------------------------------------------ ModuleConfig --

static std::string clean(const std::string &s, const char *chars) {
  size_t a = s.find_first_not_of(chars);
  if (a == std::string::npos) return "";
  size_t b = s.find_last_not_of(chars);
  return s.substr(a, b - a + 1);
}

static std::vector<std::string> split_ws(const std::string &s) {
  std::vector<std::string> out;
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) i++;
    size_t j = i;
    while (j < s.size() && s[j] != ' ' && s[j] != '\t') j++;
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j;
  }
  return out;
}

static bool next_line(const std::string &text, size_t *pos, std::string *line) {
  if (*pos >= text.size()) return false;
  size_t e = text.find('\n', *pos);
  if (e == std::string::npos) e = text.size();
  *line = text.substr(*pos, e - *pos);
  if (!line->empty() && line->back() == '\r') line->pop_back();
  *pos = e + 1;
  return true;
}

// str::str2float (aku/str.cc:260-282): strtod narrowed to float
static float str2float(const std::string &s, bool *ok) {
  char *end;
  float v = (float)strtod(s.c_str(), &end);
  if (s.empty() || *end != '\0') *ok = false;
  return v;
}
static long str2long(const std::string &s, bool *ok) {
  char *end;
  long v = strtol(s.c_str(), &end, 10);
  if (s.empty() || *end != '\0') *ok = false;
  return v;
}

void ModuleConfig::read(const std::string &text, size_t *pos) {
  bool first_line = true;
  std::string line;
  while (true) {
    if (!next_line(text, pos, &line))
      raise(AASR_ERR_INVALID, "unexpected end of module config file");
    line = clean(line, " \t");
    if (line.empty()) continue;
    if (first_line) {
      if (line != "{")
        raise(AASR_ERR_INVALID, "'{' expected in module config file: %s", line.c_str());
      first_line = false;
      continue;
    }
    if (line == "}") break;
    size_t sp = line.find_first_of(" \t");
    if (sp == std::string::npos)
      raise(AASR_ERR_INVALID, "value missing for option: %s", line.c_str());
    std::string key = line.substr(0, sp);
    std::string val = clean(line.substr(sp), " \t");
    if (val.empty()) raise(AASR_ERR_INVALID, "value missing for option: %s", line.c_str());
    if (index.count(key)) raise(AASR_ERR_INVALID, "value redefined: %s", line.c_str());
    index[key] = (int)values.size();
    names.push_back(key);
    values.push_back(val);
  }
}

void ModuleConfig::insert(const std::string &name, const std::string &value) {
  auto it = index.find(name);
  if (it == index.end()) {
    index[name] = (int)values.size();
    names.push_back(name);
    values.push_back(value);
  } else {
    values[it->second] = value;
  }
}

static std::string fmt_g(float v) {
  char buf[64];
  snprintf(buf, sizeof buf, "%g", v);
  return buf;
}

void ModuleConfig::set(const std::string &name, float value) { insert(name, fmt_g(value)); }

void ModuleConfig::set(const std::string &name, const std::vector<float> &vec) {
  std::string v;
  for (size_t i = 0; i < vec.size(); i++) {
    if (i) v += " ";
    v += fmt_g(vec[i]);
  }
  insert(name, v);
}

bool ModuleConfig::get(const std::string &k, std::string &v) const {
  auto it = index.find(k);
  if (it == index.end()) return false;
  v = values[it->second];
  return true;
}
bool ModuleConfig::get(const std::string &k, int &v) const {
  auto it = index.find(k);
  if (it == index.end()) return false;
  bool ok = true;
  v = (int)str2long(values[it->second], &ok);
  if (!ok) raise(AASR_ERR_INVALID, "invalid integer value: %s", values[it->second].c_str());
  return true;
}
bool ModuleConfig::get(const std::string &k, float &v) const {
  auto it = index.find(k);
  if (it == index.end()) return false;
  bool ok = true;
  v = str2float(values[it->second], &ok);
  if (!ok) raise(AASR_ERR_INVALID, "invalid float value: %s", values[it->second].c_str());
  return true;
}
bool ModuleConfig::get(const std::string &k, std::vector<float> &v) const {
  auto it = index.find(k);
  if (it == index.end()) return false;
  std::vector<std::string> f = split_ws(values[it->second]);
  v.resize(f.size());
  bool ok = true;
  for (size_t i = 0; i < f.size(); i++) {
    v[i] = str2float(f[i], &ok);
    if (!ok)
      raise(AASR_ERR_INVALID, "invalid value '%s' in float vector: %s", f[i].c_str(),
            values[it->second].substr(0, 200).c_str());
  }
  return true;
}
bool ModuleConfig::get(const std::string &k, std::vector<std::string> &v) const {
  auto it = index.find(k);
  if (it == index.end()) return false;
  v = split_ws(values[it->second]);
  return true;
}

// ------------------------------------------------------------- FFT tables --

// Radix schedule of KissFFT's kf_factor (vendor/kiss_fft/kiss_fft.c:309-331):
// 4s first, then 2s, then odd primes.  Radices 2, 3, 4 and 5 have kernels.
static void build_fft_plan(FftPlan &p, int window) {
  if (window & 1) raise(AASR_ERR_INVALID, "Real FFT optimization must be even.");
  int nc = window / 2;
  if (nc < 2) raise(AASR_ERR_UNSUPPORTED, "FFT window %d too short", window);
  p.nc = nc;
  int n = nc, r = 4, ns = 0;
  double root = floor(sqrt((double)n));
  do {
    while (n % r) {
      if (r == 4) r = 2;
      else if (r == 2) r = 3;
      else r += 2;
      if (r > root) r = n;
    }
    n /= r;
    if (r > 64)  // kFftMaxRadix (feat_kernels.hip): the generic butterfly's private scratch
      raise(AASR_ERR_UNSUPPORTED, "FFT window %d has the prime factor %d; radices up to 64 are built", window, r);
    if (ns >= 16) raise(AASR_ERR_UNSUPPORTED, "FFT window %d too long", window);
    p.radix[ns] = r;
    p.sublen[ns] = n;
    ns++;
  } while (n > 1);
  p.ns = ns;
  if (nc > 2048) raise(AASR_ERR_UNSUPPORTED, "FFT window %d > 4096 samples is not built", window);

  // FFTModule::set_module_config (aku/FeatureModules.cc:488-490)
  std::vector<float> ham((size_t)window);
  for (int i = 0; i < window; i++)
    ham[i] = .54 - .46 * cosf(2 * M_PI * i / (window - 1.0));
  p.hamming.upload(ham.data(), ham.size());
  // kiss_fft_alloc twiddles (vendor/kiss_fft/kiss_fft.c:355-363)
  std::vector<float> tw((size_t)nc * 2);
  const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
  for (int i = 0; i < nc; i++) {
    double phase = -2 * pi * i / nc;
    tw[2 * i] = (float)cos(phase);
    tw[2 * i + 1] = (float)sin(phase);
  }
  p.twiddle.upload(tw.data(), tw.size());
  // kiss_fftr_alloc super twiddles (vendor/kiss_fft/kiss_fftr.c:57-63)
  std::vector<float> st((size_t)(nc / 2 + 1) * 2, 0.0f);
  for (int i = 0; i < nc / 2; i++) {
    double phase = -3.14159265358979323846264338327 * ((double)(i + 1) / nc + .5);
    st[2 * i] = (float)cos(phase);
    st[2 * i + 1] = (float)sin(phase);
  }
  p.stwiddle.upload(st.data(), st.size());
  // mixed-radix digit reversal implied by kf_work's recursion (:240-302)
  std::vector<int32_t> perm((size_t)nc);
  for (int o = 0; o < nc; o++) {
    int rem = o, src = 0, stride = 1;
    for (int s = 0; s < ns; s++) {
      int k = rem / p.sublen[s];
      rem -= k * p.sublen[s];
      src += k * stride;
      stride *= p.radix[s];
    }
    perm[o] = src;
  }
  p.perm.upload(perm.data(), perm.size());
}

// ---------------------------------------------------- vtln / sr_norm tables --

// util::sinc (aku/util.hh:151-159): float argument, double sine, float result
static float aku_sinc(float x) {
  const double PI = 3.14159265358979323846;
  if (fabs(x) < 1e-8) return 1;
  double y = PI * x;
  return sin(y) / y;
}

// VtlnModule::set_warp_factor / set_slapt_warp (aku/FeatureModules.cc:1600-1623):
// warped bin positions, then the interpolation weights of every output bin.
static void build_vtln_tables(FeatModule &m) {
  const int dim = m.dim;
  std::vector<float> &bins = m.vtln_bins;
  bins.assign((size_t)dim, 0.0f);
  int t;
  if (m.use_slapt) {
    // create_slapt_bins (:1669-1686)
    for (t = 0; t < dim - 1; t++) {
      double nf = M_PI * (double)t / (dim - 1);
      bins[t] = t;
      for (int i = 0; i < (int)m.slapt_params.size(); i++)
        bins[t] += m.slapt_params[i] * sin((i + 1) * nf) * (dim - 1);
    }
    bins[t] = dim - 1;
  } else if (m.use_pwlin) {
    // create_pwlin_bins (:1625-1651), float arithmetic throughout
    float border, slope = 0, point = 0;
    bool limit = false;
    border = m.pwlin_turn * (float)(dim - 1);
    for (t = 0; t < dim - 1; t++) {
      if (!limit) bins[t] = m.warp_factor * (float)t;
      else bins[t] = slope * (float)t + point;
      if (!limit && (t >= border || bins[t] >= border)) {
        slope = ((float)dim - 1 - bins[t]) / ((float)dim - 1 - t);
        point = (1 - slope) * (float)(dim - 1);
        limit = true;
      }
    }
    bins[t] = (float)(dim - 1);
  } else {
    // create_blin_bins (:1653-1667)
    for (t = 0; t < dim - 1; t++) {
      double nf = M_PI * (double)t / (dim - 1);
      bins[t] = t + 2 * atan2((m.warp_factor - 1) * sin(nf), 1 + (1 - m.warp_factor) * cos(nf)) / M_PI * (dim - 1);
    }
    bins[t] = dim - 1;
  }
  m.d_vtln_bins.upload(bins.data(), bins.size());
  std::vector<int32_t> start((size_t)dim, 0), len((size_t)dim, 0);
  std::vector<float> coef;
  if (m.all_pass) {
    // create_all_pass_blin_transform / create_all_pass_slapt_transform + set_all_pass_transform
    // (:1716-1756, 1758-1868, 1870-1905): final = IDCT * (warp series * DCT); the reference multiplies
    // with BLAS dgemm (summation order unpinned), plain k-ascending loops here.
    const size_t n = (size_t)dim;
    std::vector<double> tr(n * n, 0.0), dct(n * n), tmp(n * n);
    double temp;
    if (m.use_slapt) {
      // exp(f1) as a Taylor sum of convolution powers of f1 (terms 0..10) -> two-sided sequence q;
      // row i of the transform from the i-th convolution power of q folded about its centre
      auto conv_at = [](const std::vector<double> &a, const std::vector<double> &b, int j) {
        int high1 = j, low1 = 0, low2 = j;
        if (high1 >= (int)a.size()) high1 = (int)a.size() - 1;
        if (low2 >= (int)b.size()) {
          low1 = j - (int)b.size() + 1;
          low2 = (int)b.size() - 1;
        }
        double t = 0;
        for (int k = 0; k < high1 - low1 + 1; k++) t += a[(size_t)(low1 + k)] * b[(size_t)(low2 - k)];
        return t;
      };
      const int order = (int)m.slapt_params.size();
      std::vector<double> f1((size_t)(2 * order + 1), 0.0), q((size_t)(2 * dim + 1), 0.0), cur(1, 1.0), fn;
      for (int i = 0; i < order; i++) {
        f1[(size_t)i] = -m.slapt_params[(size_t)(order - i - 1)] * M_PI / 2;
        f1[(size_t)(i + order + 1)] = m.slapt_params[(size_t)i] * M_PI / 2;
      }
      int cur_center = 0;
      double cur_m = 1;
      for (int i = 0; i <= 10; i++) {
        if (i > 0) cur_m = cur_m / (double)i;
        const int low1 = std::max(0, dim - cur_center), high1 = std::min(2 * dim + 1, dim + cur_center + 1);
        for (int j = low1; j < high1; j++) q[(size_t)j] = q[(size_t)j] + cur_m * cur[(size_t)(j - (dim + 1) + cur_center + 1)];
        fn.assign(f1.size() + cur.size() - 1, 0.0);
        for (int j = 0; j < (int)fn.size(); j++) fn[(size_t)j] = conv_at(cur, f1, j);
        cur = fn;
        cur_center = ((int)cur.size() - 1) / 2;
      }
      q.pop_back();  // "make the initial sequence symmetric"
      q.pop_back();
      const std::vector<double> q1 = q;
      std::vector<double> qn(q.size(), 0.0);
      tr[0] = 1;
      for (int i = 1; i < dim; i++) {
        tr[(size_t)i] = 2 * q[(size_t)(dim - 1)];
        for (int j = 1; j < dim; j++) tr[(size_t)j * n + i] = q[(size_t)(dim + j - 1)] + q[(size_t)(dim - j - 1)];
        for (int j = dim - 1; j < 3 * dim - 2; j++) qn[(size_t)(j - dim + 1)] = conv_at(q, q1, j);
        q = qn;
      }
    } else {
      std::vector<double> q1(n, 0.0), q(n, 0.0), qn(n, 0.0);
      double alpha = m.warp_factor - 1;
      q1[0] = -alpha;
      temp = 1 - alpha * alpha;
      for (int i = 1; i < dim; i++) {
        q1[i] = temp;
        temp *= alpha;
      }
      q[0] = 1;
      tr[0] = 1;
      for (int i = 1; i < dim; i++) {
        for (int j = 0; j < dim; j++) {
          temp = 0;
          for (int k = 0; k <= j; k++) temp += q[k] * q1[j - k];
          qn[j] = temp;
        }
        q = qn;
        tr[i] = 2 * q[0];
        for (int j = 1; j < dim; j++) tr[j * n + i] = q[j];
      }
    }
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) dct[i * n + j] = cos(i * (j + 0.5) * M_PI / dim);
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) {
        double a = 0;
        for (int k = 0; k < dim; k++) a += tr[i * n + k] * dct[k * n + j];
        tmp[i * n + j] = a;
      }
    for (int i = 0; i < dim; i++) {
      dct[i * n] = 1.0 / dim;
      for (int j = 1; j < dim; j++) dct[i * n + j] = cos((i + 0.5) * j * M_PI / dim) * 2 / dim;
    }
    coef.assign(n * n, 0.0f);
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) {
        double a = 0;
        for (int k = 0; k < dim; k++) a += dct[i * n + k] * tmp[k * n + j];
        coef[i * n + j] = a;
      }
    std::fill(len.begin(), len.end(), dim);
    m.sp_stride = dim;
  } else if (m.sinc_rad > 0) {
    // create_sinc_coef_table (:1688-1714)
    const int rad = m.sinc_rad;
    m.sp_stride = 2 * rad + 1;
    coef.assign((size_t)dim * m.sp_stride, 0.0f);
    for (int b = 0; b < dim; b++) {
      int cent = (int)(bins[b] + 0.5);
      int min_i = std::max(cent - rad, 0);
      int max_i = std::min(cent + rad + 1, dim);
      start[b] = min_i;
      len[b] = std::max(0, max_i - min_i);
      for (int i = min_i; i < max_i; i++) {
        float w = aku_sinc(i - bins[b]);
        if (m.lanczos) {
          if (fabs(i - bins[b]) < rad) w *= aku_sinc((i - bins[b]) / (float)rad);
          else w = 0;
        }
        coef[(size_t)b * m.sp_stride + (i - min_i)] = w;
      }
    }
  } else {
    m.sp_stride = 0;  // linear interpolation on the bins themselves
  }
  m.sp_start.upload(start.data(), start.size());
  m.sp_len.upload(len.data(), len.size());
  m.sp_coef.upload(coef.data(), coef.size());
}

// SRNormModule::set_speech_rate (aku/FeatureModules.cc:2003-2034)
static void build_srnorm_table(FeatModule &m) {
  float in_cent = (float)(m.in_frames - 1) / 2;
  float out_cent = (float)(m.out_frames - 1) / 2;
  m.sp_stride = 2 * m.lanczos_order + 1;
  std::vector<int32_t> start((size_t)m.out_frames, 0), len((size_t)m.out_frames, 0);
  std::vector<float> coef((size_t)m.out_frames * m.sp_stride, 0.0f);
  for (int i = 0; i < m.out_frames; i++) {
    float target_pos = (i - out_cent) / m.speech_rate + in_cent;
    int cent = (int)roundf(target_pos);
    int a = std::max(cent - m.lanczos_order, 0);
    int b = std::min(cent + m.lanczos_order + 1, m.in_frames);
    start[i] = a;
    len[i] = std::max(0, b - a);
    for (int j = a; j < b; j++) {
      float w = aku_sinc(j - target_pos);
      if (fabs(j - target_pos) < m.lanczos_order) w *= aku_sinc((j - target_pos) / (float)m.lanczos_order);
      else w = 0;
      coef[(size_t)i * m.sp_stride + (j - a)] = w;
    }
  }
  m.sp_start.upload(start.data(), start.size());
  m.sp_len.upload(len.data(), len.size());
  m.sp_coef.upload(coef.data(), coef.size());
}

// ------------------------------------------------------- module configure --

static void configure(aasr_feat *h, FeatModule &m, const ModuleConfig &c) {
  auto src = [&](int i) -> const FeatModule & { return h->mods[m.sources[i]]; };
  switch (m.type) {
    case MOD_AUDIOFILE: {
      // AudioFileModule::set_module_config (aku/FeatureModules.cc:327-360)
      if (!c.get("sample_rate", m.sample_rate))
        raise(AASR_ERR_INVALID, "AudioFileModule: Must set sample rate");
      m.emph = 0.97;
      c.get("pre_emph_coef", m.emph);
      m.frame_rate = 125;
      c.get("frame_rate", m.frame_rate);
      m.advance = m.sample_rate / m.frame_rate;
      m.width = (int)(2 * m.sample_rate / m.frame_rate);
      c.get("window_width", m.width);
      m.dim = m.width;
      m.copy_borders = 1;
      c.get("copy_borders", m.copy_borders);
      {
        // :345-356 -- headerless input and its byte order
        std::string endian;
        c.get("endian", endian);
        m.endian = endian == "little" ? 1 : endian == "big" ? 2 : 0;
        int raw = 0;
        c.get("raw", raw);
        m.raw_audio = raw ? 1 : 0;
      }
      if (m.width <= 0 || !(m.advance > 0))
        raise(AASR_ERR_INVALID, "AudioFileModule: invalid window (%d samples, advance %g)", m.width, m.advance);
      break;
    }
    case MOD_PRE: {
      // PreModule::set_module_config (aku/FeatureModules.cc:672-690)
      m.frame_rate = 125;
      m.sample_rate = 16000;
      m.legacy_file = 0;
      c.get("sample_rate", m.sample_rate);
      c.get("frame_rate", m.frame_rate);
      c.get("legacy_file", m.legacy_file);
      if (!c.get("dim", m.dim)) raise(AASR_ERR_INVALID, "PreModule: Must set dimension");
      if (m.legacy_file && m.dim > 127)
        raise(AASR_ERR_INVALID, "PreModule: legacy files store the dimension in one signed byte");
      m.width = 0;
      m.advance = m.sample_rate / m.frame_rate;
      break;
    }
    case MOD_FFT: {
      // FFTModule::set_module_config (aku/FeatureModules.cc:475-518)
      m.magnitude = 1;
      c.get("magnitude", m.magnitude);
      m.take_log = 0;
      c.get("log", m.take_log);
      int source_dim = src(0).dim;
      m.dim = source_dim / 2 + 1;
      build_fft_plan(m.fft, source_dim);
      break;
    }
    case MOD_MEL: {
      // MelModule::set_module_config / create_mel_bins (aku/FeatureModules.cc:775-803)
      m.root = 0;
      c.get("root", m.root);
      int sr = h->mods[0].sample_rate;
      m.dim = (int)((21 + 2) * log10f(1 + sr / 1400.0) / log10f(1 + 16000 / 1400.0) - 2);
      if (m.dim < 1) raise(AASR_ERR_INVALID, "MelModule: sample rate %d gives no mel bins", sr);
      int edges = m.dim + 2;
      float rate = sr;
      float mel_step = 2595 * log10f(1.0 + rate / 1400.0) / edges;
      std::vector<float> edge((size_t)edges);
      int sdim = src(0).dim;
      for (int i = 0; i < edges; i++)
        edge[i] = 1400.0 * (pow(10, (i + 1) * mel_step / 2595) - 1) * (sdim - 1) / rate;
      // MelModule::generate's triangular ramps (:805-835), evaluated once: the
      // float scale of every (bin, t) term in accumulation order and the float
      // running sum per bin.
      std::vector<int32_t> off(1, 0), tt;
      std::vector<float> sc, sums;
      for (int b = 0; b < m.dim; b++) {
        float sum = 0, scale;
        float beg = edge[b] - 1;
        float end = edge[b + 1];
        int t = (int)std::max(ceilf(beg), 0.0f);
        while (t < end) {
          scale = (t - beg) / (end - beg);
          if (t >= sdim) raise(AASR_ERR_INVALID, "MelModule: bin edge beyond the spectrum");
          tt.push_back(t);
          sc.push_back(scale);
          sum += scale;
          t++;
        }
        beg = end;
        end = edge[b + 2];
        while (t < end) {
          scale = (end - t) / (end - beg);
          if (t >= sdim) raise(AASR_ERR_INVALID, "MelModule: bin edge beyond the spectrum");
          tt.push_back(t);
          sc.push_back(scale);
          sum += scale;
          t++;
        }
        sums.push_back(sum);
        off.push_back((int32_t)tt.size());
      }
      m.mel_off.upload(off.data(), off.size());
      m.mel_t.upload(tt.data(), tt.size());
      m.mel_scale.upload(sc.data(), sc.size());
      m.mel_sum.upload(sums.data(), sums.size());
      {
        // k_spectral_fused gives a frame's bins to its 16 lanes in rounds; a round lasts as long as its longest bin, so
        // the bins go out by falling term count (the 16 longest together, the short ones in the last round)
        std::vector<int32_t> order((size_t)m.dim);
        for (int b = 0; b < m.dim; b++) order[(size_t)b] = b;
        std::stable_sort(order.begin(), order.end(),
                         [&](int32_t a, int32_t b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
        m.mel_order.upload(order.data(), order.size());
      }
      break;
    }
    case MOD_POWER:
      m.dim = 1;  // PowerModule::set_module_config (aku/FeatureModules.cc:866-872)
      break;
    case MOD_DCT: {
      // DCTModule::set_module_config / generate (aku/FeatureModules.cc:937-979)
      m.dim = 12;
      m.zeroth = 0;
      c.get("dim", m.dim);
      if (m.dim < 1) raise(AASR_ERR_INVALID, "DCTModule: Dimension must be > 0");
      c.get("zeroth", m.zeroth);
      int sdim = src(0).dim;
      int bias = m.zeroth ? 1 : 0;
      std::vector<float> cs((size_t)std::max(1, m.dim - bias) * sdim);
      for (int i = 0; i < m.dim - bias; i++)
        for (int b = 0; b < sdim; b++)
          cs[(size_t)i * sdim + b] = cosf((i + 1) * (b + 0.5) * M_PI / sdim);
      m.dct_cos.upload(cs.data(), cs.size());
      break;
    }
    case MOD_DELTA: {
      // DeltaModule::set_module_config (aku/FeatureModules.cc:998-1016)
      m.dim = src(0).dim;
      m.delta_width = 2;
      c.get("width", m.delta_width);
      m.delta_norm = 2 * m.delta_width * (m.delta_width + 1) * (2 * m.delta_width + 1) / 6;
      c.get("normalization", m.delta_norm);
      if (m.delta_width < 1) raise(AASR_ERR_INVALID, "DeltaModule: Delta width must be > 0");
      m.own_left = m.own_right = m.delta_width;
      break;
    }
    case MOD_NORMALIZATION: {
      // NormalizationModule::set_module_config (aku/FeatureModules.cc:1056-1086)
      m.dim = src(0).dim;
      m.mean.assign(m.dim, 0.0f);
      m.scale.assign(m.dim, 1.0f);
      c.get("mean", m.mean);
      if ((int)m.mean.size() != m.dim)
        raise(AASR_ERR_INVALID, "NormalizationModule: Invalid mean dimension");
      if (c.exists("var") && c.exists("scale"))
        raise(AASR_ERR_INVALID, "NormalizationModule: Both scale and var can not be defined simultaneously");
      if (c.get("var", m.scale)) {
        if ((int)m.scale.size() != m.dim)
          raise(AASR_ERR_INVALID, "Normalization module: Invalid variance dimension");
        for (int i = 0; i < m.dim; i++) m.scale[i] = 1 / sqrtf(m.scale[i]);
      } else if (c.get("scale", m.scale)) {
        if ((int)m.scale.size() != m.dim)
          raise(AASR_ERR_INVALID, "NormalizationModule: Invalid scale dimension");
      }
      m.d_mean.upload(m.mean.data(), m.mean.size());
      m.d_scale.upload(m.scale.data(), m.scale.size());
      break;
    }
    case MOD_LIN_TRANSFORM: {
      // LinTransformModule::set_module_config / check_transform_parameters
      // (aku/FeatureModules.cc:1167-1241)
      m.src_dim = src(0).dim;
      m.dim = m.src_dim;
      m.matrix.clear();
      m.bias.clear();
      c.get("matrix", m.matrix);
      c.get("bias", m.bias);
      c.get("dim", m.dim);
      if (m.dim < 1) raise(AASR_ERR_INVALID, "LinTransformModule: Dimension must be > 0");
      m.matrix_defined = !m.matrix.empty();
      m.bias_defined = !m.bias.empty();
      m.orig_matrix = m.matrix;
      m.orig_bias = m.bias;
      if (m.matrix_defined && (int)m.matrix.size() != m.dim * m.src_dim)
        raise(AASR_ERR_INVALID, "LinTransformModule: Invalid matrix dimension");
      if (m.bias_defined && (int)m.bias.size() != m.dim)
        raise(AASR_ERR_INVALID, "LinTransformModule: Invalid bias dimension");
      if (!m.matrix_defined && m.dim > m.src_dim)
        raise(AASR_ERR_INVALID, "LinTransformModule: identity transform needs dim <= source dim");
      m.d_matrix.upload(m.matrix.data(), m.matrix.size());
      m.d_bias.upload(m.bias.data(), m.bias.size());
      break;
    }
    case MOD_MERGE: {
      // MergerModule::set_module_config (aku/FeatureModules.cc:1341-1349)
      m.dim = 0;
      std::vector<int32_t> sc;
      for (size_t i = 0; i < m.sources.size(); i++) {
        for (int j = 0; j < src((int)i).dim; j++) {
          sc.push_back((int32_t)i);
          sc.push_back(j);
        }
        m.dim += src((int)i).dim;
      }
      m.merge_src_col.upload(sc.data(), sc.size());
      break;
    }
    case MOD_MEAN_SUBTRACTOR: {
      // MeanSubtractorModule::set_module_config (aku/FeatureModules.cc:1384-1404)
      m.dim = src(0).dim;
      m.cms_left = 75;
      c.get("left", m.cms_left);
      m.cms_right = 75;
      c.get("right", m.cms_right);
      if (m.cms_left + 1 < 1 || m.cms_right + 1 < 1)
        raise(AASR_ERR_INVALID, "MeanSubtractorModule: context widths must be >= 0");
      m.own_left = m.cms_left;
      m.own_right = m.cms_right;
      m.lead_left = kCmsLead;
      break;
    }
    case MOD_CONCAT: {
      // ConcatModule::set_module_config (aku/FeatureModules.cc:1472-1486)
      m.own_left = m.own_right = 0;
      c.get("left", m.own_left);
      c.get("right", m.own_right);
      if (m.own_left < 0 || m.own_right < 0)
        raise(AASR_ERR_INVALID, "ConcatModule: context spans must be >= 0");
      m.src_dim = src(0).dim;
      m.dim = m.src_dim * (1 + m.own_left + m.own_right);
      break;
    }
    case MOD_MEL_POWER:
      m.dim = 1;  // MelPowerModule::set_module_config (aku/FeatureModules.cc:904-910)
      break;
    case MOD_VTLN: {
      // VtlnModule::set_module_config (aku/FeatureModules.cc:1529-1573)
      m.dim = src(0).dim;
      m.use_pwlin = 0;
      m.pwlin_turn = 0.8;
      c.get("pwlin_vtln", m.use_pwlin);
      c.get("pwlin_turnpoint", m.pwlin_turn);
      m.use_slapt = 0;
      c.get("slapt", m.use_slapt);
      if (m.use_pwlin && m.use_slapt)
        raise(AASR_ERR_INVALID, "VtlnModule: Can not use both pwlin_vtln and slapt!");
      m.sinc_rad = 8;
      c.get("sinc_interpolation_rad", m.sinc_rad);
      m.all_pass = 0;
      c.get("all-pass", m.all_pass);
      if (m.use_pwlin && m.all_pass)
        raise(AASR_ERR_INVALID, "VtlnModule: Can not use both pwlin_vtln and all-pass!");
      int lanczos = m.all_pass ? 0 : 1;
      c.get("lanczos_window", lanczos);
      m.lanczos = lanczos > 0;
      if (m.lanczos && m.all_pass)
        raise(AASR_ERR_INVALID, "VtlnModule: Can not use both lanczos_window and all-pass!");
      m.warp_factor = 1.0;
      m.slapt_params.assign(1, 0.0f);
      build_vtln_tables(m);
      break;
    }
    case MOD_SR_NORM: {
      // SRNormModule::set_module_config (aku/FeatureModules.cc:1953-1987)
      m.in_frames = m.out_frames = 0;
      c.get("in_frames", m.in_frames);
      c.get("out_frames", m.out_frames);
      if (m.in_frames == 0 || m.out_frames == 0)
        raise(AASR_ERR_INVALID, "SRNormModule: Must set both in_frames and out_frames.");
      if (m.in_frames < 0 || m.out_frames < 0)
        raise(AASR_ERR_INVALID, "SRNormModule: frame counts must be positive");
      m.frame_dim = src(0).dim / m.in_frames;
      if (src(0).dim % m.in_frames != 0)
        raise(AASR_ERR_INVALID, "SRNormModule: in_frames does not match with the input dimension");
      m.dim = m.out_frames * m.frame_dim;
      m.lanczos_order = 4;
      c.get("lanczos_order", m.lanczos_order);
      if (m.lanczos_order < 1) raise(AASR_ERR_INVALID, "SRNormModule: lanczos_order must be positive.");
      m.speech_rate = 1.0;
      c.get("speech_rate", m.speech_rate);
      build_srnorm_table(m);
      break;
    }
    case MOD_HOST: {
      // the user's set_module_config: dimension and look-around come from the callback
      const HostModuleType &ht = host_module_types()[(size_t)m.host_type];
      m.opt_names = c.names;
      m.opt_values = c.values;
      std::string block = "{\n";
      for (size_t k = 0; k < c.names.size(); k++) block += "  " + c.names[k] + " " + c.values[k] + "\n";
      block += "}\n";
      std::vector<int32_t> sdims;
      for (int sidx : m.sources) sdims.push_back(h->mods[sidx].dim);
      int32_t dim = 0, left = 0, right = 0;
      char err[512] = {0};
      if (!ht.vtbl.configure || !ht.vtbl.generate)
        raise(AASR_ERR_INVALID, "module type %s was registered without callbacks", ht.name.c_str());
      if (ht.vtbl.configure(ht.user, m.name.c_str(), block.c_str(), (int32_t)sdims.size(), sdims.data(), &dim, &left,
                            &right, &m.host_instance, err, (int32_t)sizeof err - 1) != 0)
        raise(AASR_ERR_INVALID, "%s", err[0] ? err : "user module configuration failed");
      if (left < 0 || right < 0) raise(AASR_ERR_INVALID, "module %s: negative look-around", m.name.c_str());
      m.dim = dim;
      m.own_left = left;
      m.own_right = right;
      break;
    }
    case MOD_QUANTEQ:
      // QuantEqModule::set_module_config (aku/FeatureModules.cc:2078-2083); the
      // channel parameters only arrive through set_parameters
      m.dim = src(0).dim;
      m.quant_train.clear();
      c.get("quant_train", m.quant_train);
      break;
  }
  if (m.dim <= 0) raise(AASR_ERR_INVALID, "module %s has no output dimension", m.name.c_str());
}

static std::vector<HostModuleType> &host_registry() {
  static std::vector<HostModuleType> r;
  return r;
}
const std::vector<HostModuleType> &host_module_types() { return host_registry(); }

int register_host_module_type(const char *name, const aasr_host_module *vtbl, void *user) {
  static const char *builtin[] = {"audiofile", "fft", "mel", "power", "dct", "delta", "normalization",
                                  "lin_transform", "merge", "mean_subtractor", "concat", "vtln", "sr_norm",
                                  "mel_power", "quanteq", "pre"};
  if (!name || !*name || !vtbl) raise(AASR_ERR_INVALID, "aasr_feat_register_module_type: null argument");
  for (const char *b : builtin)
    if (std::string(b) == name) raise(AASR_ERR_INVALID, "module type '%s' is built in", name);
  std::vector<HostModuleType> &r = host_registry();
  for (size_t k = 0; k < r.size(); k++)
    if (r[k].name == name) {  // re-registration replaces the callbacks (handles created earlier keep index k)
      r[k].vtbl = *vtbl;
      r[k].user = user;
      return (int)k;
    }
  r.push_back(HostModuleType{name, *vtbl, user});
  return (int)r.size() - 1;
}

aasr_feat *feat_create(const std::string &text) {
  require_device();
  aasr_feat *h = new aasr_feat();
  try {
    AASR_HIP(hipGetDevice(&h->device));
    size_t pos = 0;
    std::string line;
    int lineno = 0;
    while (next_line(text, &pos, &line)) {
      lineno++;
      line = clean(line, " \t");
      if (line.empty()) continue;
      if (line != "module")
        raise(AASR_ERR_INVALID, "expected keyword 'module' on line %d: %s", lineno, line.c_str());
      ModuleConfig cfg;
      size_t before = pos;
      try {
        cfg.read(text, &pos);
      } catch (Error &e) {
        raise(e.code, "failed reading feature module around line %d: %s", lineno, e.msg.c_str());
      }
      for (size_t i = before; i < pos && i < text.size(); i++)
        if (text[i] == '\n') lineno++;
      std::string type, name;
      if (!cfg.get("type", type))
        raise(AASR_ERR_INVALID, "type not defined for module ending on line %d", lineno);
      if (!cfg.get("name", name))
        raise(AASR_ERR_INVALID, "name not defined for module ending on line %d", lineno);
      if (name.find_first_of(" \t\n") != std::string::npos)
        raise(AASR_ERR_INVALID, "module name may not contain whitespaces");
      h->mods.emplace_back();  // DevBuf members are non-copyable: build in place
      FeatModule &m = h->mods.back();
      m.name = name;
      m.type_str = type;
      static const struct { const char *s; ModType t; } kinds[] = {
          {"audiofile", MOD_AUDIOFILE}, {"fft", MOD_FFT}, {"mel", MOD_MEL},
          {"power", MOD_POWER}, {"dct", MOD_DCT}, {"delta", MOD_DELTA},
          {"normalization", MOD_NORMALIZATION}, {"lin_transform", MOD_LIN_TRANSFORM},
          {"merge", MOD_MERGE}, {"mean_subtractor", MOD_MEAN_SUBTRACTOR},
          {"concat", MOD_CONCAT}, {"vtln", MOD_VTLN}, {"sr_norm", MOD_SR_NORM},
          {"mel_power", MOD_MEL_POWER}, {"quanteq", MOD_QUANTEQ}, {"pre", MOD_PRE}};
      bool found = false;
      for (auto &k : kinds)
        if (type == k.s) {
          m.type = k.t;
          found = true;
        }
      if (!found) {
        const std::vector<HostModuleType> &reg = host_module_types();
        for (size_t k = 0; k < reg.size(); k++)
          if (reg[k].name == type) {
            m.type = MOD_HOST;
            m.host_type = (int)k;
            found = true;
          }
      }
      if (!found) raise(AASR_ERR_INVALID, "Unknown module type '%s'", type.c_str());
      const bool is_first = h->mods.size() == 1;
      const bool is_base = m.type == MOD_AUDIOFILE || m.type == MOD_PRE;
      if (is_first && !is_base)
        raise(AASR_ERR_INVALID, "first module should be a base module");
      if (h->by_name.count(name))
        raise(AASR_ERR_INVALID, "multiple definitions of module name: %s", name.c_str());
      bool has_sources = cfg.exists("sources");
      if (is_first && has_sources)
        raise(AASR_ERR_INVALID, "can not define sources for the first module");
      if (!is_first && !has_sources)
        raise(AASR_ERR_INVALID, "sources not defined for module: %s", name.c_str());
      if (!is_first && is_base)
        raise(AASR_ERR_INVALID, "base module FFT can not have sources");
      if (has_sources) {
        std::vector<std::string> srcs;
        cfg.get("sources", srcs);
        for (auto &s : srcs) {
          auto it = h->by_name.find(s);
          if (it == h->by_name.end())
            raise(AASR_ERR_INVALID, "unknown source module: %s", s.c_str());
          if (!m.sources.empty() && m.type != MOD_MERGE && m.type != MOD_HOST)
            raise(AASR_ERR_INVALID, "Multiple sources are not allowed for module %s", type.c_str());
          m.sources.push_back(it->second);
        }
      }
      h->by_name[name] = (int)h->mods.size() - 1;
      configure(h, m, cfg);
    }
    if (h->mods.empty()) raise(AASR_ERR_INVALID, "no feature modules defined");
    h->bufs.resize(h->mods.size());
  } catch (...) {
    delete h;
    throw;
  }
  return h;
}

// AudioFileModule::last_frame (aku/FeatureModules.cc:305-308): int/float, truncated
int feat_last_frame(const aasr_feat *h, int64_t n_samples) {
  const FeatModule &a = h->mods[0];
  // PreModule::last_frame (aku/FeatureModules.cc:649-660): whole frames in the file - 1;
  // n_samples counts int16 units, two per float
  if (a.type == MOD_PRE) return (int)(n_samples / 2 / a.dim) - 1;
  return (int)(((int)n_samples - a.width - 1) / a.advance);
}

// The frame at which a sequential reader meets the end of the input: the first frame whose window
// [ws, ws + width + 1), ws = (int)(frame * advance) in float, crosses it (AudioFileModule::generate,
// aku/FeatureModules.cc:399-413 sets m_eof_frame = frame there; phone_probs stops at that frame and
// the border copy repeats frame m_eof_frame - 1).  Equal to last_frame() + 1 for an integral
// advance below 2^24 samples; the float formula of last_frame() can be one off beyond that and
// with a fractional advance.
int feat_eof_frame(const aasr_feat *h, int64_t n_samples) {
  const FeatModule &a = h->mods[0];
  if (a.type == MOD_PRE) return feat_last_frame(h, n_samples) + 1;
  if (n_samples < a.width + 1) return 0;
  int g = std::max(1, feat_last_frame(h, n_samples) + 1);
  while (g > 1 && (int64_t)(int)((float)(g - 1) * a.advance) + a.width + 1 > n_samples) g--;
  while ((int64_t)(int)((float)g * a.advance) + a.width + 1 <= n_samples) g++;
  return g;
}

void feat_halo(const aasr_feat *h, int target, int *left, int *right) {
  // longest accumulated look-around from `target` down to the base module
  std::vector<int> L(h->mods.size(), -1), R(h->mods.size(), -1);
  L[target] = R[target] = 0;
  for (int i = target; i >= 0; i--) {
    if (L[i] < 0) continue;
    const FeatModule &m = h->mods[i];
    for (int s : m.sources) {
      L[s] = std::max(L[s], L[i] + m.own_left + m.lead_left);
      R[s] = std::max(R[s], R[i] + m.own_right);
    }
  }
  *left = std::max(0, L[0]);
  *right = std::max(0, R[0]);
}

void feat_set_parameters(aasr_feat *h, const std::string &module, const std::string &block) {
  ModuleConfig c;
  size_t pos = 0;
  c.read(block, &pos);
  feat_set_parameters(h, module, c);
}

// FeatureGenerator::write_configuration (aku/FeatureGenerator.cc:222-243) over each module's
// get_config / get_module_config (aku/FeatureModules.cc:173-179 and per type): "module", the
// option block in the order the reference sets the keys ("%d" / "%g" values), "sources" last.
std::string feat_write_configuration(const aasr_feat *h) {
  std::string out;
  auto I = [](int v) { return std::to_string(v); };
  for (const FeatModule &m : h->mods) {
    ModuleConfig c;
    c.insert("name", m.name);
    c.insert("type", m.type_str);
    switch (m.type) {
      case MOD_AUDIOFILE:  // :311-325
        c.set("pre_emph_coef", m.emph);
        c.insert("sample_rate", I(m.sample_rate));
        c.set("frame_rate", m.frame_rate);
        c.insert("window_width", I(m.width));
        c.insert("copy_borders", I(m.copy_borders));
        if (m.endian == 1) c.insert("endian", "little");
        else if (m.endian == 2) c.insert("endian", "big");
        if (m.raw_audio) c.insert("raw", "1");
        break;
      case MOD_PRE:  // :661-669
        c.insert("sample_rate", I(m.sample_rate));
        c.set("frame_rate", m.frame_rate);
        c.insert("dim", I(m.dim));
        if (m.legacy_file) c.insert("legacy_file", I(m.legacy_file));
        break;
      case MOD_FFT:  // :468-473
        c.insert("magnitude", I(m.magnitude));
        if (m.take_log) c.insert("log", I(m.take_log));
        break;
      case MOD_MEL:  // :769-773
        if (m.root) c.insert("root", I(m.root));
        break;
      case MOD_DCT:  // :934-939
        c.insert("dim", I(m.dim));
        c.insert("zeroth", I(m.zeroth));
        break;
      case MOD_DELTA:  // :992-996
        c.insert("width", I(m.delta_width));
        c.set("normalization", m.delta_norm);
        break;
      case MOD_NORMALIZATION:  // :1050-1054
        c.set("mean", m.mean);
        c.set("scale", m.scale);
        break;
      case MOD_LIN_TRANSFORM:  // :1155-1164: the transformation as configured
        c.insert("dim", I(m.dim));
        if (!m.orig_matrix.empty()) c.set("matrix", m.orig_matrix);
        if (!m.orig_bias.empty()) c.set("bias", m.orig_bias);
        break;
      case MOD_MEAN_SUBTRACTOR:  // :1378-1382
        c.insert("left", I(m.cms_left));
        c.insert("right", I(m.cms_right));
        break;
      case MOD_CONCAT:  // :1467-1471
        c.insert("left", I(m.own_left));
        c.insert("right", I(m.own_right));
        break;
      case MOD_VTLN:  // :1513-1527
        if (m.use_pwlin) {
          c.insert("pwlin_vtln", I(m.use_pwlin));
          c.set("pwlin_turnpoint", m.pwlin_turn);
        }
        if (m.use_slapt) c.insert("slapt", "1");
        if (m.lanczos) c.insert("lanczos_window", "1");
        c.insert("sinc_interpolation_rad", I(m.sinc_rad));
        if (m.all_pass > 0) c.insert("all-pass", I(m.all_pass));
        break;
      case MOD_SR_NORM:  // :1947-1952
        c.insert("in_frames", I(m.in_frames));
        c.insert("out_frames", I(m.out_frames));
        c.insert("lanczos_order", I(m.lanczos_order));
        break;
      case MOD_QUANTEQ:  // :2071-2074
        c.set("quant_train", m.quant_train);
        break;
      case MOD_HOST:  // a user type: its options as they were read
        for (size_t k = 0; k < m.opt_names.size(); k++)
          if (m.opt_names[k] != "name" && m.opt_names[k] != "type" && m.opt_names[k] != "sources")
            c.insert(m.opt_names[k], m.opt_values[k]);
        break;
      default:  // power, mel_power, merge: no options
        break;
    }
    if (!m.sources.empty()) {
      std::string v;
      for (size_t i = 0; i < m.sources.size(); i++) v += (i ? " " : "") + h->mods[(size_t)m.sources[i]].name;
      c.insert("sources", v);
    }
    // ModuleConfig::write with indent 0 (aku/ModuleConfig.cc:204-222)
    out += "module\n{\n";
    for (size_t i = 0; i < c.names.size(); i++) out += "  " + c.names[i] + " " + c.values[i] + "\n";
    out += "}\n\n";
  }
  return out;
}

void feat_get_parameters(const aasr_feat *h, const std::string &module, ModuleConfig &c) {
  auto it = h->by_name.find(module);
  if (it == h->by_name.end())
    raise(AASR_ERR_INVALID, "unknown module requested: %s", module.c_str());
  const FeatModule &m = h->mods[it->second];
  switch (m.type) {
    case MOD_NORMALIZATION:  // aku/FeatureModules.cc:1114-1119
      c.set("mean", m.mean);
      c.set("scale", m.scale);
      break;
    case MOD_LIN_TRANSFORM: {  // :1197-1202; undefined parts read back as identity / zeros
      std::vector<float> mat = m.matrix, bias = m.bias;
      if (!m.matrix_defined) {
        mat.assign((size_t)m.dim * m.src_dim, 0.0f);
        for (int r = 0; r < m.dim && r < m.src_dim; r++) mat[(size_t)r * m.src_dim + r] = 1.0f;
      }
      if (!m.bias_defined) bias.assign((size_t)m.dim, 0.0f);
      c.set("matrix", mat);
      c.set("bias", bias);
      break;
    }
    case MOD_VTLN:  // :1594-1600
      if (m.use_slapt) c.set("slapt_coef", m.slapt_params);
      else c.set("warp_factor", m.warp_factor);
      break;
    case MOD_SR_NORM:  // :1998-2001
      c.set("speech_rate", m.speech_rate);
      break;
    case MOD_QUANTEQ:  // :2096-2102
      c.set("alpha", m.q_alpha);
      c.set("gamma", m.q_gamma);
      c.set("quant_max", m.q_max);
      break;
    default:  // FeatureModule::get_parameters is a no-op (aku/FeatureModule.hh:108)
      break;
  }
}

void feat_set_parameters(aasr_feat *h, const std::string &module, const ModuleConfig &c) {
  auto it = h->by_name.find(module);
  if (it == h->by_name.end())
    raise(AASR_ERR_INVALID, "unknown module requested: %s", module.c_str());
  FeatModule &m = h->mods[it->second];
  if (m.type == MOD_NORMALIZATION) {
    // NormalizationModule::set_parameters (aku/FeatureModules.cc:1089-1112)
    c.get("mean", m.mean);
    if ((int)m.mean.size() != m.dim)
      raise(AASR_ERR_INVALID, "NormalizationModule: Invalid mean dimension");
    if (c.exists("var") && c.exists("scale"))
      raise(AASR_ERR_INVALID, "NormalizationModule: Both scale and var can not be defined simultaneously");
    if (c.get("var", m.scale)) {
      if ((int)m.scale.size() != m.dim)
        raise(AASR_ERR_INVALID, "Normalization module: Invalid variance dimension");
      for (int i = 0; i < m.dim; i++) m.scale[i] = 1 / sqrtf(m.scale[i]);
    } else if (c.get("scale", m.scale)) {
      if ((int)m.scale.size() != m.dim)
        raise(AASR_ERR_INVALID, "NormalizationModule: Invalid scale dimension");
    }
    m.d_mean.upload(m.mean.data(), m.mean.size());
    m.d_scale.upload(m.scale.data(), m.scale.size());
  } else if (m.type == MOD_LIN_TRANSFORM) {
    // LinTransformModule::set_parameters (aku/FeatureModules.cc:1188-1196)
    m.matrix.clear();
    m.bias.clear();
    c.get("matrix", m.matrix);
    c.get("bias", m.bias);
    m.matrix_defined = !m.matrix.empty();
    m.bias_defined = !m.bias.empty();
    if (m.matrix_defined && (int)m.matrix.size() != m.dim * m.src_dim)
      raise(AASR_ERR_INVALID, "LinTransformModule: Invalid matrix dimension");
    if (m.bias_defined && (int)m.bias.size() != m.dim)
      raise(AASR_ERR_INVALID, "LinTransformModule: Invalid bias dimension");
    m.d_matrix.upload(m.matrix.data(), m.matrix.size());
    m.d_bias.upload(m.bias.data(), m.bias.size());
  } else if (m.type == MOD_VTLN) {
    // VtlnModule::set_parameters (aku/FeatureModules.cc:1575-1592)
    if (m.use_slapt) {
      m.slapt_params.assign(1, 0.0f);
      c.get("slapt_coef", m.slapt_params);
    } else {
      m.warp_factor = 1.0;
      c.get("warp_factor", m.warp_factor);
    }
    build_vtln_tables(m);
  } else if (m.type == MOD_SR_NORM) {
    // SRNormModule::set_parameters (aku/FeatureModules.cc:1990-1996)
    m.speech_rate = 1.0;
    c.get("speech_rate", m.speech_rate);
    build_srnorm_table(m);
  } else if (m.type == MOD_QUANTEQ) {
    // QuantEqModule::set_parameters (aku/FeatureModules.cc:2085-2094)
    if (!c.get("alpha", m.q_alpha)) m.q_alpha.clear();
    if (!c.get("gamma", m.q_gamma)) m.q_gamma.clear();
    if (!c.get("quant_max", m.q_max)) m.q_max.clear();
    const bool any = !m.q_alpha.empty() && !m.q_gamma.empty() && !m.q_max.empty();
    if (any && ((int)m.q_alpha.size() < m.dim || (int)m.q_gamma.size() < m.dim || (int)m.q_max.size() < m.dim))
      raise(AASR_ERR_INVALID, "QuantEqModule: alpha, gamma and quant_max need %d values each", m.dim);
    m.d_q_alpha.upload(m.q_alpha.data(), m.q_alpha.size());
    m.d_q_gamma.upload(m.q_gamma.data(), m.q_gamma.size());
    m.d_q_max.upload(m.q_max.data(), m.q_max.size());
  } else {
    // FeatureModule::set_parameters default is a no-op (aku/FeatureModule.hh:107)
  }
}

}  // namespace aasr
