// feat_kernels.hip -- device evaluation of the aku feature graph for a batch
// of utterances (gfx950).
//
// Each kernel restates one FeatureModule::generate with the reference's exact
// arithmetic type per operation (float32 islands inside a double pipeline) so
// that the features agree with the CPU path to the last few ulps; the build
// disables floating-point contraction for this reason.  References (relative
// to the AaltoASR tree, aku/FeatureModules.cc unless noted):
//   AudioFileModule::generate :370-440     FFTModule::generate :520-566
//   kiss_fftr / kf_bfly2 / kf_bfly4  vendor/kiss_fft/kiss_fftr.c:67-121,
//                                     vendor/kiss_fft/kiss_fft.c:21-90
//   MelModule::generate :805-849           PowerModule::generate :874-885
//   DCTModule::generate :955-979           DeltaModule::generate :1018-1037
//   NormalizationModule::generate :1135-1142
//   LinTransformModule::generate :1243-1269   MergerModule::generate :1351-1364
//   MeanSubtractorModule::generate :1413-1454 (full-window branch)
//
// Layout: module m keeps a double buffer [rows_m x dim_m]; utterance u owns
// rows key_m(u) .. key_m(u+1)-1 with key_m(u) = frame_off[u] + u*(L_m+R_m),
// covering frames first[u]-L_m .. first[u]+n_u-1+R_m (L_m/R_m = look-around
// the module's consumers need).  These kernels are HBM/latency bound and tiny
// next to scoring (412 algorithmic bytes per frame).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <climits>

#include "feat.h"

namespace aasr {

struct DevBatch {
  int n_utts;
  const int64_t *frame_off;  // [n+1]
  const int64_t *pcm_off;    // [n+1]
  const int32_t *first;      // [n]
  const int32_t *eof_frame;  // [n] feat_eof_frame(): first frame whose window crosses the end
};

// largest u with key(u) = frame_off[u] + u*span <= r
__device__ __forceinline__ int find_utt(const DevBatch &b, int64_t r, int span) {
  int lo = 0, hi = b.n_utts - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    int64_t key = b.frame_off[mid] + (int64_t)mid * span;
    if (key <= r) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// The same for a workgroup's first row (r uniform: the loads are scalar): the bisection is nine DEPENDENT loads at 360
// utterances, 16 us of k_temporal_fused's 165.  First guess = utterances of equal length (the usual shape of a
// batch), corrected by a walk of at most three; the bisection finishes on what the walk left when lengths differ a lot.
__device__ __forceinline__ int find_utt_near(const DevBatch &b, int64_t r, int span, int64_t rows) {
  const int n = b.n_utts;
  int u = (int)((float)r * (float)n / (float)rows);
  u = u < 0 ? 0 : (u > n - 1 ? n - 1 : u);
  int lo = 0, hi = n - 1;
  for (int step = 0; step < 3; step++) {
    const int64_t k0 = b.frame_off[u] + (int64_t)u * span;
    const int64_t k1 = b.frame_off[u + 1] + (int64_t)(u + 1) * span;
    if (k0 > r && u > 0) {
      hi = u - 1;
      u--;
    } else if (k1 <= r && u < n - 1) {
      lo = u + 1;
      u++;
    } else {
      return u;
    }
  }
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    int64_t key = b.frame_off[mid] + (int64_t)mid * span;
    if (key <= r) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// translation of a module row to its source's row (same frame)
struct SrcMap {
  int span_diff;  // (Ls+Rs) - (L+R)
  int shift;      // Ls - L
};
__device__ __forceinline__ int64_t src_row(const DevBatch &b, int64_t r, int span, SrcMap sm) {
  if (sm.span_diff == 0) return r + sm.shift;
  int u = find_utt(b, r, span);
  return r + (int64_t)u * sm.span_diff + sm.shift;
}

struct AudioPrm {
  int width;
  float advance, emph;
  int copy_borders;
};

__device__ __forceinline__ float preemph_sample(const int16_t *pcm, int64_t nsamp, int64_t i,
                                                float emph) {
  // y = x[i+1] - emph*x[i] in FLOAT (short - float*short); zero outside the file
  float a = (i + 1 >= 0 && i + 1 < nsamp) ? (float)pcm[i + 1] : 0.0f;
  float c = (i >= 0 && i < nsamp) ? (float)pcm[i] : 0.0f;
  float prod = emph * c;
  return a - prod;
}

__device__ __forceinline__ int64_t window_start(const DevBatch &b, int u, int frame, AudioPrm ap) {
  int src = frame;
  if (ap.copy_borders) {
    int eof = b.eof_frame[u];
    if (src < 0) src = 0;
    if (src >= eof) src = eof - 1;
  }
  return (int64_t)(int)((float)src * ap.advance);
}

__global__ void k_audio_frames(DevBatch b, const int16_t *__restrict__ pcm, AudioPrm ap, int L,
                               int R, int64_t rows, double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * ap.width) return;
  int64_t r = idx / ap.width;
  int j = (int)(idx - r * ap.width);
  int u = find_utt(b, r, L + R);
  int frame = b.first[u] - L + (int)(r - (b.frame_off[u] + (int64_t)u * (L + R)));
  int64_t ws = window_start(b, u, frame, ap);
  const int16_t *p = pcm + b.pcm_off[u];
  int64_t ns = b.pcm_off[u + 1] - b.pcm_off[u];
  dst[idx] = (double)preemph_sample(p, ns, ws + j, ap.emph);
}

// PreModule::generate (aku/FeatureModules.cc:705-755): frame t of the feature file;
// frames before 0 repeat frame 0, frames from the end of the file on repeat the last
// one.  `in` is the float view of the input buffer (pcm_off counts int16 units).
__global__ void k_pre_frames(DevBatch b, const float *__restrict__ in, int dim, int L, int R,
                             int64_t rows, double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int j = (int)(idx - r * dim);
  int u = find_utt(b, r, L + R);
  int frame = b.first[u] - L + (int)(r - (b.frame_off[u] + (int64_t)u * (L + R)));
  const int eof = b.eof_frame[u];
  if (frame < 0) frame = 0;
  if (frame >= eof) frame = eof - 1;
  dst[idx] = (double)in[b.pcm_off[u] / 2 + (int64_t)frame * dim + j];
}

// floor(e / d) for 0 <= e < 65536 and 1 <= d < 65536 without the ~25-instruction integer division:
// magic = floor(2^32 / d) + 1 (fast_magic), one v_mul_hi_u32; d = 1 has no 32-bit magic and is passed through
__host__ __device__ __forceinline__ unsigned fast_magic(int d) { return d > 1 ? 0xFFFFFFFFu / (unsigned)d + 1u : 0u; }
__device__ __forceinline__ int fast_div(int e, unsigned magic) {
  return magic ? (int)__umulhi((unsigned)e, magic) : e;
}

struct FftPrm {
  int nc, ns;
  int radix[16], sublen[16];
  unsigned sub_magic[16];  // fast_magic(sublen[s])
  int fstride[16];         // nc / (radix[s] * sublen[s])
  const float *hamming, *twiddle, *stwiddle;
  const int32_t *perm;
  int magnitude, take_log;
};

struct cpx { float r, i; };
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {
  cpx m;
  m.r = a.r * b.r - a.i * b.i;
  m.i = a.r * b.i + a.i * b.r;
  return m;
}

// kf_bfly2 / kf_bfly4 on values (vendor/kiss_fft/kiss_fft.c:21-90): shared by the LDS stages below and by the two
// stages k_spectral_fused runs in registers, so both perform the same operations in the same order
__device__ __forceinline__ void bfly2_values(cpx &f0, cpx &f1, cpx w) {
  const cpx t = cmul(f1, w);
  cpx o1;
  o1.r = f0.r - t.r;
  o1.i = f0.i - t.i;
  f0.r += t.r;
  f0.i += t.i;
  f1 = o1;
}
__device__ __forceinline__ void bfly4_values(cpx &f0, cpx &f1, cpx &f2, cpx &f3, cpx w1, cpx w2, cpx w3) {
  const cpx s0 = cmul(f1, w1);
  const cpx s1 = cmul(f2, w2);
  const cpx s2 = cmul(f3, w3);
  cpx s3, s4, s5, o1, o2, o3;
  s5.r = f0.r - s1.r;  s5.i = f0.i - s1.i;
  f0.r += s1.r;        f0.i += s1.i;
  s3.r = s0.r + s2.r;  s3.i = s0.i + s2.i;
  s4.r = s0.r - s2.r;  s4.i = s0.i - s2.i;
  o2.r = f0.r - s3.r;  o2.i = f0.i - s3.i;
  f0.r += s3.r;        f0.i += s3.i;
  o1.r = s5.r + s4.i;  o1.i = s5.i - s4.r;
  o3.r = s5.r - s4.i;  o3.i = s5.i + s4.r;
  f1 = o1;
  f2 = o2;
  f3 = o3;
}

// One butterfly of KissFFT's decimation-in-time stage (vendor/kiss_fft/kiss_fft.c:21-198, radices 2,
// 3, 4, 5): butterfly index bi of the stage with radix pr and sub-length m, in place in buf.
__device__ __forceinline__ void kiss_butterfly(cpx *buf, int bi, int pr, int m, unsigned m_magic, int fstride,
                                               const cpx *__restrict__ tw) {
    int g = fast_div(bi, m_magic), j = bi - g * m;
    cpx *F = buf + g * pr * m;
    if (pr == 2) {
      cpx f0 = F[j], f1 = F[m + j];
      bfly2_values(f0, f1, tw[j * fstride]);
      F[m + j] = f1;
      F[j] = f0;
    } else if (pr == 3) {
      // kf_bfly3 (vendor/kiss_fft/kiss_fft.c:92-135); HALF_OF(x) = x*.5 in double
      const cpx epi3 = tw[fstride * m];
      cpx s1 = cmul(F[m + j], tw[j * fstride]);
      cpx s2 = cmul(F[2 * m + j], tw[2 * j * fstride]);
      cpx f0 = F[j], s3, s0, o1, o2;
      s3.r = s1.r + s2.r;  s3.i = s1.i + s2.i;
      s0.r = s1.r - s2.r;  s0.i = s1.i - s2.i;
      o1.r = (float)((double)f0.r - (double)s3.r * .5);
      o1.i = (float)((double)f0.i - (double)s3.i * .5);
      s0.r *= epi3.i;
      s0.i *= epi3.i;
      f0.r += s3.r;
      f0.i += s3.i;
      o2.r = o1.r + s0.i;
      o2.i = o1.i - s0.r;
      o1.r -= s0.i;
      o1.i += s0.r;
      F[j] = f0;
      F[m + j] = o1;
      F[2 * m + j] = o2;
    } else if (pr == 5) {
      // kf_bfly5 (vendor/kiss_fft/kiss_fft.c:137-198)
      const cpx ya = tw[fstride * m], yb = tw[fstride * 2 * m];
      cpx s0 = F[j];
      cpx s1 = cmul(F[m + j], tw[j * fstride]);
      cpx s2 = cmul(F[2 * m + j], tw[2 * j * fstride]);
      cpx s3 = cmul(F[3 * m + j], tw[3 * j * fstride]);
      cpx s4 = cmul(F[4 * m + j], tw[4 * j * fstride]);
      cpx s5, s6, s7, s8, s9, s10, s11, s12, f0 = s0, o1, o2, o3, o4;
      s7.r = s1.r + s4.r;   s7.i = s1.i + s4.i;
      s10.r = s1.r - s4.r;  s10.i = s1.i - s4.i;
      s8.r = s2.r + s3.r;   s8.i = s2.i + s3.i;
      s9.r = s2.r - s3.r;   s9.i = s2.i - s3.i;
      f0.r += s7.r + s8.r;
      f0.i += s7.i + s8.i;
      s5.r = s0.r + s7.r * ya.r + s8.r * yb.r;
      s5.i = s0.i + s7.i * ya.r + s8.i * yb.r;
      s6.r = s10.i * ya.i + s9.i * yb.i;
      s6.i = -(s10.r * ya.i) - s9.r * yb.i;
      o1.r = s5.r - s6.r;  o1.i = s5.i - s6.i;
      o4.r = s5.r + s6.r;  o4.i = s5.i + s6.i;
      s11.r = s0.r + s7.r * yb.r + s8.r * ya.r;
      s11.i = s0.i + s7.i * yb.r + s8.i * ya.r;
      s12.r = -(s10.i * yb.i) + s9.i * ya.i;
      s12.i = s10.r * yb.i - s9.r * ya.i;
      o2.r = s11.r + s12.r;  o2.i = s11.i + s12.i;
      o3.r = s11.r - s12.r;  o3.i = s11.i - s12.i;
      F[j] = f0;
      F[m + j] = o1;
      F[2 * m + j] = o2;
      F[3 * m + j] = o3;
      F[4 * m + j] = o4;
    } else {
      cpx f0 = F[j], f1 = F[m + j], f2 = F[2 * m + j], f3 = F[3 * m + j];
      bfly4_values(f0, f1, f2, f3, tw[j * fstride], tw[2 * j * fstride], tw[3 * j * fstride]);
      F[j] = f0;
      F[m + j] = f1;
      F[2 * m + j] = f2;
      F[3 * m + j] = f3;
    }
}

// kf_bfly_generic (vendor/kiss_fft/kiss_fft.c:198-235) for the odd prime radices above 5: a plain
// DFT of the p points u, u + m, ..., the twiddle index advanced by fstride * k per term and wrapped
// once.  One thread per u; the p inputs sit in private memory, which is why only the unfused FFT
// kernel's <GEN = true> instance carries this path (feat_graph.cc limits p to kFftMaxRadix).
constexpr int kFftMaxRadix = 64;
__device__ __noinline__ void kiss_butterfly_generic(cpx *buf, int bi, int pr, int m, int fstride,
                                                    const cpx *__restrict__ tw) {
  const int g = bi / m, u = bi - g * m;
  const int n = fstride * pr * m;
  cpx *F = buf + g * pr * m;
  cpx scratch[kFftMaxRadix];
  for (int q1 = 0, k = u; q1 < pr; q1++, k += m) scratch[q1] = F[k];
  for (int q1 = 0, k = u; q1 < pr; q1++, k += m) {
    int twidx = 0;
    cpx acc = scratch[0];
    for (int q = 1; q < pr; q++) {
      twidx += fstride * k;
      if (twidx >= n) twidx -= n;
      const cpx t = cmul(scratch[q], tw[twidx]);
      acc.r += t.r;
      acc.i += t.i;
    }
    F[k] = acc;
  }
}

// logf as glibc computes it (sysdeps/ieee754/flt-32/e_logf.c and logf_data.c of glibc 2.27 and later,
// from ARM's optimized-routines; the reference's MelModule / FFTModule call logf, aku/FeatureModules.cc:
// 564, 846): x = 2^k z with z in [0x3f330000, 2 x that), a 16-entry table of c near the centre of z's
// sub-interval (invc ~ 1/c, logc ~ ln c), r = z * invc - 1, ln x = k ln2 + logc + r + r^2 (A2 + A1 r +
// A0 r^2), everything in double, rounded to float at the end.  The result is within 0.82 ulp but not
// always the correctly rounded one -- which a (float) log((double) x) would be -- and that last bit
// used to show as ~1e-6 in a few frames' features.  Non-finite, non-positive and subnormal arguments
// (which the feature chain does not produce: it takes log(val / sum + 1)) go the double way.
__device__ __forceinline__ float glibc_logf(float x) {
  const double T[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
      {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
      {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
      {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
      {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
      {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
      {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
  const double Ln2 = 0x1.62e42fefa39efp-1;
  const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
  const uint32_t ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.0f;
  if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return (float)log((double)x);
  const uint32_t tmp = ix - 0x3f330000u;
  const int i = (int)((tmp >> 19) & 15u);
  const int k = (int32_t)tmp >> 23;
  const uint32_t iz = ix - (tmp & 0xff800000u);
  const double z = (double)__uint_as_float(iz);
  const double r = z * T[i][0] - 1.0;
  const double y0 = T[i][1] + (double)k * Ln2;
  const double r2 = r * r;
  double y = A1 * r + A2;
  y = A0 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

// |X|^2 of one bin in float, then sqrtf / logf as FFTModule::generate applies them
// (aku/FeatureModules.cc:533-565); sqrt and log evaluated in double and rounded once
__device__ __forceinline__ float spec_value(float re, float im, int magnitude, int take_log) {
  float a = re * re;
  float c = im * im;
  float v = a + c;
  if (magnitude) v = (float)sqrt((double)v);  // == correctly rounded sqrtf
  if (take_log) v = glibc_logf(v);
  return v;
}

// kiss_fftr's split of the packed half-length transform into bins k and nc-k (kiss_fftr.c:86-120)
__device__ __forceinline__ void real_split(const cpx *buf, int nc, int k, const cpx *__restrict__ stw,
                                           float &re_k, float &im_k, float &re_n, float &im_n) {
  cpx fpk = buf[k], fpnk, f1k, f2k, t;
  fpnk.r = buf[nc - k].r;
  fpnk.i = -buf[nc - k].i;
  f1k.r = fpk.r + fpnk.r;  f1k.i = fpk.i + fpnk.i;
  f2k.r = fpk.r - fpnk.r;  f2k.i = fpk.i - fpnk.i;
  t = cmul(f2k, stw[k - 1]);
  float ar = f1k.r + t.r, ai = f1k.i + t.i, br = f1k.r - t.r, bi = t.i - f1k.i;
  // HALF_OF(x) = x * .5 promotes to double in the reference; halving is exact, so the float product is the same value
  // (one rounding of the exact result, subnormal results included) at a third of the instructions
  re_k = ar * .5f;
  im_k = ai * .5f;
  re_n = br * .5f;
  im_n = bi * .5f;
}

// One wave per frame: Hamming -> packed half-length complex FFT in LDS with
// KissFFT's butterfly order -> real split -> |X|^2 (/ sqrt / log).
template <bool FROM_PCM, bool GEN>
__global__ __launch_bounds__(256) void k_fft(DevBatch b, const int16_t *__restrict__ pcm,
                                             AudioPrm ap, const double *__restrict__ src,
                                             SrcMap sm, int L, int R, int64_t rows, FftPrm fp,
                                             double *__restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= rows) return;  // whole wave exits together
  cpx *buf = (cpx *)smem_raw + (size_t)wave * fp.nc;
  const int nc = fp.nc, width = 2 * nc;

  const int16_t *p = nullptr;
  int64_t nsamp = 0, ws = 0;
  const double *xrow = nullptr;
  if (FROM_PCM) {
    int u = find_utt(b, r, L + R);
    int frame = b.first[u] - L + (int)(r - (b.frame_off[u] + (int64_t)u * (L + R)));
    ws = window_start(b, u, frame, ap);
    p = pcm + b.pcm_off[u];
    nsamp = b.pcm_off[u + 1] - b.pcm_off[u];
  } else {
    xrow = src + src_row(b, r, L + R, sm) * width;
  }
  // datain[t] = (float)(hamming[t] * x[t]); packed (even, odd) -> complex, loaded
  // through KissFFT's digit-reversed order
  for (int o = lane; o < nc; o += 64) {
    int s = fp.perm[o];
    double x0, x1;
    if (FROM_PCM) {
      x0 = (double)preemph_sample(p, nsamp, ws + 2 * s, ap.emph);
      x1 = (double)preemph_sample(p, nsamp, ws + 2 * s + 1, ap.emph);
    } else {
      x0 = xrow[2 * s];
      x1 = xrow[2 * s + 1];
    }
    cpx v;
    v.r = (float)((double)fp.hamming[2 * s] * x0);
    v.i = (float)((double)fp.hamming[2 * s + 1] * x1);
    buf[o] = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const cpx *tw = (const cpx *)fp.twiddle;
  for (int s = fp.ns - 1; s >= 0; s--) {
    const int pr = fp.radix[s], m = fp.sublen[s];
    const int fstride = fp.fstride[s];
    const int nb = fstride * m;  // nc / pr
    if (GEN && pr > 5) {
      for (int bi = lane; bi < nb; bi += 64) kiss_butterfly_generic(buf, bi, pr, m, fstride, tw);
    } else {
      for (int bi = lane; bi < nb; bi += 64) kiss_butterfly(buf, bi, pr, m, fp.sub_magic[s], fstride, tw);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // real split (kiss_fftr.c:86-120) + power spectrum (FeatureModules.cc:533-565)
  double *out = dst + r * (int64_t)(nc + 1);
  const cpx *stw = (const cpx *)fp.stwiddle;
  if (lane == 0) {
    cpx t0 = buf[0];
    out[0] = (double)spec_value(t0.r + t0.i, 0.0f, fp.magnitude, fp.take_log);
    out[nc] = (double)spec_value(t0.r - t0.i, 0.0f, fp.magnitude, fp.take_log);
  }
  for (int k = 1 + lane; k <= nc / 2; k += 64) {
    float re_k, im_k, re_n, im_n;
    real_split(buf, nc, k, stw, re_k, im_k, re_n, im_n);
    if (k != nc - k) out[k] = (double)spec_value(re_k, im_k, fp.magnitude, fp.take_log);
    out[nc - k] = (double)spec_value(re_n, im_n, fp.magnitude, fp.take_log);  // for k == nc-k the later assignment wins
  }
}


// ---------------------------------------------------------------------------
// Fused spectral front end: audiofile -> fft -> {mel -> dct, power} -> merge in ONE kernel.
// The unfused chain writes the (nc+1)-bin spectrum as doubles (0.5 GB per hour of audio) and reads
// it back twice; here a workgroup takes SPEC_FRAMES frames through every stage in LDS and only the
// merged cepstra + log power (13 doubles per frame) leave the chip.  All 256 threads stay busy:
// the FFT stages spread frames x butterflies over the threads, and the serial float
// accumulations of the reference (PowerModule's left-to-right sum of 129 bins, MelModule's per-bin
// ramps) run as one task per (frame, bin) side by side instead of one frame per wave.
// Every operation is the one the single-module kernels perform (same device functions, same
// order), so the result is bit-identical to the unfused path (tests/test_feat_gpu.py).
// ---------------------------------------------------------------------------
// phase ablations exist only in AASR_BUILD_ABLATION=1 builds (tools/ablate_spectral.sh, tools/ablate_temporal.sh):
//   k_spectral_fused, AASR_SPEC_DBG bits: 1 no utterance search, 8 / 32 no sample fetch + window, 2 no FFT stages,
//     64 no real split, 4 no mel, 16 no power sum, 128 no DCT / stores
//   k_temporal_fused, AASR_TEMP_DBG: 1 no transform products, 2 no second difference / normalisation, 4 no first
//     difference, 8 no matrix staging, 16 no row loads, 32 no stores, 64 no utterance search
//   k_mean_subtract_tiled, AASR_CMS_DBG: 1 no window sums, 2 no block sums, 4 no staging
#ifndef AASR_ABLATION
#define AASR_ABLATION 0
#endif
#if AASR_ABLATION
#define AASR_FDBG(bits) (dbg & (bits))
#else
#define AASR_FDBG(bits) false
#endif
constexpr int SPEC_FRAMES = 16;   // frames per pass of a workgroup
constexpr int SPEC_TPF = 16;      // threads per frame (a wave = 4 frames)

struct SpectralPrm {
  FftPrm fp;
  int mel_dim, mel_root, mel_terms;
  const int32_t *mel_off, *mel_t, *mel_order;
  const float *mel_scale, *mel_sum;
  int dct_dim, zeroth;
  const float *dct_cos;
};

// the float tables stay floats in LDS: staged as doubles (no conversion per use) the kernel ran 6 % SLOWER (measured:
// 510 -> 540 us; the extra 4 KB per workgroup cost more than ~80 conversions per pass saved)
typedef float spec_tbl_t;
// LDS plan shared by the host (size) and the kernel (offsets), in bytes
struct SpectralLds {
  size_t bufs, frame_u, melv, powv, ham, perm, tw, stw, moff, mt, msc, msum, mord, dct, prm, total;
  size_t frame_u_stride;  // per frame: int16 samples first, the float spectrum later
  __host__ __device__ SpectralLds(int nc, int mel_dim, int mel_terms, int dct_rows) {
    size_t o = 0;
    auto take = [&](size_t n) {
      size_t at = o;
      o += (n + 15) & ~(size_t)15;
      return at;
    };
    bufs = take((size_t)SPEC_FRAMES * nc * 8);
    const size_t pcm_b = (size_t)(2 * nc + 2) * 2, spec_b = (size_t)((nc + 1) | 1) * 4;
    frame_u_stride = ((pcm_b > spec_b ? pcm_b : spec_b) + 7) & ~(size_t)7;
    frame_u_stride |= 4;  // not a multiple of 8 bytes: rows of different frames start on different banks
    frame_u = take((size_t)SPEC_FRAMES * frame_u_stride);
    melv = take((size_t)SPEC_FRAMES * mel_dim * 8);
    powv = 0;
    ham = take((size_t)2 * nc * sizeof(spec_tbl_t));
    perm = take((size_t)nc * 4);
    tw = take((size_t)nc * 8);
    stw = take((size_t)(nc / 2 + 1) * 8);
    moff = take((size_t)(mel_dim + 1) * 4);
    mt = take((size_t)mel_terms * 4);
    msc = take((size_t)mel_terms * sizeof(spec_tbl_t));
    msum = take((size_t)mel_dim * 4);
    mord = take((size_t)mel_dim * 4);
    dct = take((size_t)dct_rows * mel_dim * sizeof(spec_tbl_t));
    prm = take((size_t)SPEC_FRAMES * 3 * 8);
    total = o;
  }
};

__global__ __launch_bounds__(256) void k_spectral_fused(DevBatch b, const int16_t *__restrict__ pcm,
                                                        AudioPrm ap, int L, int R, int64_t rows,
                                                        SpectralPrm sp, int passes, int dbg,
                                                        double *__restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int FB = SPEC_FRAMES, TPF = SPEC_TPF;
  const int nc = sp.fp.nc, nbins = nc + 1;
  const int dct_rows = sp.dct_dim - (sp.zeroth ? 1 : 0);
  const SpectralLds lds(nc, sp.mel_dim, sp.mel_terms, dct_rows);
  cpx *bufs = (cpx *)(smem_raw + lds.bufs);
  double *melv = (double *)(smem_raw + lds.melv);
  spec_tbl_t *t_ham = (spec_tbl_t *)(smem_raw + lds.ham);
  int32_t *t_perm = (int32_t *)(smem_raw + lds.perm);
  cpx *t_tw = (cpx *)(smem_raw + lds.tw);
  cpx *t_stw = (cpx *)(smem_raw + lds.stw);
  int32_t *t_moff = (int32_t *)(smem_raw + lds.moff);
  int32_t *t_mt = (int32_t *)(smem_raw + lds.mt);
  spec_tbl_t *t_msc = (spec_tbl_t *)(smem_raw + lds.msc);
  float *t_msum = (float *)(smem_raw + lds.msum);
  int32_t *t_mord = (int32_t *)(smem_raw + lds.mord);
  spec_tbl_t *t_dct = (spec_tbl_t *)(smem_raw + lds.dct);
  int64_t *s_prm = (int64_t *)(smem_raw + lds.prm);  // [FB][3]: window start, utterance offset, samples
  const int tid = threadIdx.x;
  const int f = tid / TPF, l = tid - f * TPF;

  // tables: once per workgroup (every later access is an LDS read; the loads of the FFT, mel and
  // DCT stages used to wait on global memory one after another)
  for (int i = tid; i < 2 * nc; i += 256) t_ham[i] = (spec_tbl_t)sp.fp.hamming[i];
  for (int i = tid; i < nc; i += 256) {
    t_perm[i] = sp.fp.perm[i];
    t_tw[i] = ((const cpx *)sp.fp.twiddle)[i];
  }
  for (int i = tid; i < nc / 2; i += 256) t_stw[i] = ((const cpx *)sp.fp.stwiddle)[i];
  for (int i = tid; i <= sp.mel_dim; i += 256) t_moff[i] = sp.mel_off[i];
  for (int i = tid; i < sp.mel_terms; i += 256) {
    t_mt[i] = sp.mel_t[i];
    t_msc[i] = (spec_tbl_t)sp.mel_scale[i];
  }
  for (int i = tid; i < sp.mel_dim; i += 256) {
    t_msum[i] = sp.mel_sum[i];
    t_mord[i] = sp.mel_order[i];
  }
  for (int i = tid; i < dct_rows * sp.mel_dim; i += 256) t_dct[i] = (spec_tbl_t)sp.dct_cos[i];

  cpx *buf = bufs + (size_t)f * nc;
  char *fu = smem_raw + lds.frame_u + (size_t)f * lds.frame_u_stride;
  float *spec = (float *)fu;      // later: its nc + 1 spectrum values
  const int out_dim = sp.dct_dim + 1;
  const int64_t nblk = (rows + FB - 1) / FB;
  int u_cur = -1;  // tid < FB: utterance of this thread's row in the previous pass (rows only move forward)
  for (int pass = 0; pass < passes; pass++) {
    const int64_t blk = (int64_t)blockIdx.x * passes + pass;
    if (blk >= nblk) break;  // workgroup-uniform
    const int64_t r0 = blk * FB;
    __syncthreads();  // tables staged / previous pass done with s_prm
    if (tid < FB) {
      int64_t r = r0 + tid;
      if (r > rows - 1) r = rows - 1;  // spare frames of the last block repeat the last row (never stored)
      int u;
      if (AASR_FDBG(1)) {
        u = 0;
      } else if (u_cur < 0) {
        u = find_utt_near(b, r, L + R, rows);  // only in the first pass
      } else {
        u = u_cur;
        while (u + 1 < b.n_utts && b.frame_off[u + 1] + (int64_t)(u + 1) * (L + R) <= r) u++;
      }
      u_cur = u;
      const int frame = b.first[u] - L + (int)(r - (b.frame_off[u] + (int64_t)u * (L + R)));
      s_prm[3 * tid] = window_start(b, u, frame, ap);
      s_prm[3 * tid + 1] = b.pcm_off[u];
      s_prm[3 * tid + 2] = b.pcm_off[u + 1] - b.pcm_off[u];
    }
    __syncthreads();
    // From here on a frame belongs to 16 threads of one wave: wave-level ordering is enough.
    // pre-emphasis in float (AudioFileModule::generate), Hamming, packed (even, odd) -> complex in KissFFT's
    // digit-reversed order.  A lane fetches the three samples of each of its points straight from global memory (a frame's
    // 514 bytes stay in the L1 cache, neighbouring frames share half of them): staging the window in LDS first cost 17
    // loads with a 64-bit range test each + as many LDS writes, 72 us of the kernel's 634.  Only a frame whose window
    // crosses an end of its file (zero outside, AudioReader) takes the range tests.
    {
      const int64_t ws = s_prm[3 * f], ns = s_prm[3 * f + 2];
      const int16_t *p = pcm + s_prm[3 * f + 1];
      const bool inside = ws >= 0 && ws + 2 * nc < ns;   // uniform over the frame's 16 lanes
      auto point = [&](int o) {
        const int s = t_perm[o];
        float c0, c1, c2;
        if (inside) {
          const int16_t *q = p + ws + 2 * s;
          c0 = (float)q[0];
          c1 = (float)q[1];
          c2 = (float)q[2];
        } else {
          const int64_t i = ws + 2 * s;
          c0 = (i >= 0 && i < ns) ? (float)p[i] : 0.0f;
          c1 = (i + 1 >= 0 && i + 1 < ns) ? (float)p[i + 1] : 0.0f;
          c2 = (i + 2 >= 0 && i + 2 < ns) ? (float)p[i + 2] : 0.0f;
        }
        const float p0 = ap.emph * c0, p1 = ap.emph * c1;
        // (float)((double) window * (double) sample) as the reference writes it: the product of two floats is exact in
        // double (48 bits), so its rounding to float IS the float product -- one instruction instead of four
        cpx v;
        v.r = t_ham[2 * s] * (c1 - p0);
        v.i = t_ham[2 * s + 1] * (c2 - p1);
        return v;
      };
      for (int o = l; o < ((AASR_FDBG(32) || AASR_FDBG(8)) ? 0 : nc); o += TPF) buf[o] = point(o);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int st = AASR_FDBG(2) ? -1 : sp.fp.ns - 1; st >= 0; st--) {
      const int pr = sp.fp.radix[st], m = sp.fp.sublen[st];
      const int fstride = sp.fp.fstride[st];
      const int nb = fstride * m;  // nc / pr
      for (int bi = l; bi < nb; bi += TPF) kiss_butterfly(buf, bi, pr, m, sp.fp.sub_magic[st], fstride, t_tw);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // real split + power spectrum into spec[0..nc] (overwrites the staged samples)
    // (k = 0 needs no split; lane 0 takes k = nc / 2 in its place, so nc / 2 + 1 values are four rounds of 16, not five)
    if (l == 0 && !AASR_FDBG(64)) {
      const cpx t0 = buf[0];
      spec[0] = spec_value(t0.r + t0.i, 0.0f, sp.fp.magnitude, sp.fp.take_log);
      spec[nc] = spec_value(t0.r - t0.i, 0.0f, sp.fp.magnitude, sp.fp.take_log);
    }
    for (int k0 = l; k0 < (AASR_FDBG(64) ? 0 : nc / 2); k0 += TPF) {
      const int k = k0 == 0 ? nc / 2 : k0;
      float re_k, im_k, re_n, im_n;
      real_split(buf, nc, k, t_stw, re_k, im_k, re_n, im_n);
      if (k != nc - k) spec[k] = spec_value(re_k, im_k, sp.fp.magnitude, sp.fp.take_log);
      spec[nc - k] = spec_value(re_n, im_n, sp.fp.magnitude, sp.fp.take_log);
    }
    // PowerModule's left-to-right float sum (aku/FeatureModules.cc:874-885) is a chain of 129 dependent additions per
    // frame.  One wave forms it for all 16 frames of the pass, a frame per lane (the waves take turns): carried by the
    // last lane of every frame it cost each of the four waves the same ~260 instructions with 4 lanes active -- 13 % of
    // this VALU-bound kernel (rocprofv3: vector unit busy 71 % of the cycles, 65 % of the lanes on average).  The
    // barrier makes every frame's spectrum visible; the next pass's first barrier keeps it until the sums are done.
    __syncthreads();
    if ((tid >> 6) == (pass & 3) && (tid & 63) < FB && !AASR_FDBG(16)) {
      const int ff = tid & 63;
      const float *sf = (const float *)(smem_raw + lds.frame_u + (size_t)ff * lds.frame_u_stride);
      float power = 0;
      int i = 0;
      for (; i + 8 <= nbins; i += 8) {  // eight reads in flight, the additions in order
        const float v0 = sf[i], v1 = sf[i + 1], v2 = sf[i + 2], v3 = sf[i + 3];
        const float v4 = sf[i + 4], v5 = sf[i + 5], v6 = sf[i + 6], v7 = sf[i + 7];
        power = power + v0;
        power = power + v1;
        power = power + v2;
        power = power + v3;
        power = power + v4;
        power = power + v5;
        power = power + v6;
        power = power + v7;
      }
      for (; i < nbins; i++) power = power + sf[i];
      if (r0 + ff < rows && !AASR_FDBG(128)) dst[(r0 + ff) * out_dim + sp.dct_dim] = log((double)power + 1e-10);
    }
    // MelModule::generate (aku/FeatureModules.cc:805-849): one bin per thread and round
    for (int slot = l; slot < (AASR_FDBG(4) ? 0 : sp.mel_dim); slot += TPF) {
      const int bin = t_mord[slot];   // by falling term count: a round lasts as long as its longest bin
      float val = 0;
      const int e_end = t_moff[bin + 1];
      int e = t_moff[bin];
      // the table and spectrum reads of four terms are issued together; the float accumulation
      // itself stays term by term, in the reference's order
      for (; e + 4 <= e_end; e += 4) {
        const double p0 = (double)t_msc[e] * (double)spec[t_mt[e]];
        const double p1 = (double)t_msc[e + 1] * (double)spec[t_mt[e + 1]];
        const double p2 = (double)t_msc[e + 2] * (double)spec[t_mt[e + 2]];
        const double p3 = (double)t_msc[e + 3] * (double)spec[t_mt[e + 3]];
        val = (float)((double)val + p0);
        val = (float)((double)val + p1);
        val = (float)((double)val + p2);
        val = (float)((double)val + p3);
      }
      for (; e < e_end; e++) val = (float)((double)val + (double)t_msc[e] * (double)spec[t_mt[e]]);
      const float q = (float)((double)val / (double)t_msum[bin]);
      double o;
      if (sp.mel_root) {
        o = pow((double)q, 0.1);
      } else {
        const float a = q + 1.0f;
        o = (double)glibc_logf(a);
      }
      melv[(size_t)f * sp.mel_dim + bin] = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // DCTModule::generate (:955-979) into the merged row [cepstra..., log power]
    if (r0 + f < rows && !AASR_FDBG(128)) {
      const double *data = melv + (size_t)f * sp.mel_dim;
      for (int i = l; i < sp.dct_dim; i += TPF) {   // column dct_dim (the log power) is written by the summing wave
        double acc = 0.0;
        if (sp.zeroth && i == 0) {
          for (int k = 0; k < sp.mel_dim; k++) acc += data[k];
        } else {
          const spec_tbl_t *c = t_dct + (size_t)(i - (sp.zeroth ? 1 : 0)) * sp.mel_dim;
#pragma unroll 4
          for (int k = 0; k < sp.mel_dim; k++) acc += data[k] * (double)c[k];
        }
        dst[(r0 + f) * out_dim + i] = acc;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// Fused temporal stage: delta -> delta-delta -> merge -> normalization -> lin_transform in one
// kernel.  A workgroup stages the source rows of ROWS frames plus the (w1 + w2)-frame look-around
// in LDS, forms both difference streams there, normalises the merged 3*dx values and multiplies
// by the transform; only the transformed rows (the mean subtractor's input, with its own
// look-around) are written.  Expressions and summation orders are those of k_delta,
// k_normalization and k_lin_transform_tiled: bit-identical to the unfused chain.
// ---------------------------------------------------------------------------
struct TemporalPrm {
  int dx, w1, w2;
  float norm1, norm2;
  const float *mean, *scale;
  const float *matrix, *bias;  // null: identity / no bias
  int dim;
};

// frame rows per workgroup of k_temporal_fused
#ifdef AASR_TR
constexpr int kTemporalRows = AASR_TR;
#else
constexpr int kTemporalRows = 64;   // x 512 threads: 156 -> 146 us against 32 x 256 (64 x 256 was slower than both)
#endif

// LDS bytes of k_temporal_fused: source rows, first differences, normalised rows, the transposed transform, vectors
static inline size_t temporal_smem(int tile_rows, int dx, int H, int w2, int dim) {
  return (size_t)(tile_rows + 2 * H) * dx * 8 + (size_t)(tile_rows + 2 * w2) * dx * 8 + (size_t)tile_rows * 3 * dx * 8 + 16 +
         (size_t)(3 * dx) * ((dim + 2) / 3) * 16 + (size_t)(6 * dx + dim) * 4;
}

template <int ROWS, int NT>
__global__ __launch_bounds__(NT) void k_temporal_fused(DevBatch b, const double *__restrict__ src,
                                                        SrcMap sm, int span, int64_t rows,
                                                        TemporalPrm tp, int dbg, double *__restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int dx = tp.dx, H = tp.w1 + tp.w2, md = 3 * tp.dx;
  const unsigned dx_magic = fast_magic(dx), dim_magic = fast_magic(tp.dim);
  double *xs = (double *)smem_raw;                              // [ROWS + 2H][dx]
  double *d1 = xs + (size_t)(ROWS + 2 * H) * dx;                // [ROWS + 2 w2][dx]
  double *nrm = d1 + (size_t)(ROWS + 2 * tp.w2) * dx;           // [ROWS][md]
  // the transform, transposed and in groups of three output rows: mt[j][g] = (m[3g][j], m[3g+1][j], m[3g+2][j], 0), so
  // one 16-byte LDS read serves three products.  The product loop was bound by the LDS pipe (a b64 and a b32 read per
  // product, 115 us of the kernel's 227): a thread now forms a 3 x 2 block of outputs (three transform rows, two
  // frames) from one b128 and two b64 reads per step, every output's products and additions in the same order.
  float4 *mt = (float4 *)(smem_raw + ((((size_t)(2 * ROWS + 2 * H + 2 * tp.w2) * dx + (size_t)ROWS * md) * 8 + 15) & ~(size_t)15));  // [md][ngrp]
  const int ngrp = (tp.dim + 2) / 3;
  const unsigned ngrp_magic = fast_magic(ngrp);
  if (tp.matrix && !AASR_FDBG(8))
    for (int e = threadIdx.x; e < md * ngrp; e += NT) {
      const int j = fast_div(e, ngrp_magic), g = e - j * ngrp;
      float4 v;
      v.x = tp.matrix[(size_t)(3 * g) * md + j];
      v.y = 3 * g + 1 < tp.dim ? tp.matrix[(size_t)(3 * g + 1) * md + j] : 0.0f;
      v.z = 3 * g + 2 < tp.dim ? tp.matrix[(size_t)(3 * g + 2) * md + j] : 0.0f;
      v.w = 0.0f;
      mt[e] = v;
    }
  // normalisation vectors and the bias: LDS copies (a global load per value and round stood in the dependent chain)
  float *nmean = (float *)(mt + (size_t)md * ngrp), *nscale = nmean + md, *nbias = nscale + md;
  for (int e = threadIdx.x; e < md; e += NT) {
    nmean[e] = tp.mean[e];
    nscale[e] = tp.scale[e];
  }
  if (tp.bias)
    for (int e = threadIdx.x; e < tp.dim; e += NT) nbias[e] = tp.bias[e];
  const int64_t tile0 = (int64_t)blockIdx.x * ROWS;
  const int64_t tile1 = tile0 + ROWS < rows ? tile0 + ROWS : rows;
  int64_t r0 = tile0;
  while (r0 < tile1) {  // one utterance segment at a time (see k_mean_subtract_tiled)
    const int u = AASR_FDBG(64) ? 0 : find_utt_near(b, r0, span, rows);
    const int64_t u_end = AASR_FDBG(64) ? rows : b.frame_off[u + 1] + (int64_t)(u + 1) * span;
    const int64_t r_end = u_end < tile1 ? u_end : tile1;
    const int64_t s0 = r0 + (int64_t)u * sm.span_diff + sm.shift;  // source row of r0
    const int n_seg = (int)(r_end - r0);
    const int n_x = n_seg + 2 * H, n_d1 = n_seg + 2 * tp.w2;
    for (int e = threadIdx.x; e < (AASR_FDBG(16) ? 0 : n_x * dx); e += NT) xs[e] = src[(s0 - H) * dx + e];
    __syncthreads();
    // DeltaModule::generate (aku/FeatureModules.cc:1018-1037) on the source rows
    for (int e = threadIdx.x; e < (AASR_FDBG(4) ? 0 : n_d1 * dx); e += NT) {
      const int j = fast_div(e, dx_magic), i = e - j * dx;
      const int c = j + tp.w1;
      double acc = 0;
      for (int k = 1; k <= tp.w1; k++) {
        const double left = xs[(size_t)(c - k) * dx + i];
        const double right = xs[(size_t)(c + k) * dx + i];
        acc += k * (right - left);
      }
      d1[e] = acc / (double)tp.norm1;
    }
    __syncthreads();
    // second difference, merge (source, delta, delta-delta) and NormalizationModule::generate (:1135-1142)
    // (one thread forms a frame's three values of one source column: no divergence between the three parts)
    for (int e = threadIdx.x; e < (AASR_FDBG(2) ? 0 : n_seg * dx); e += NT) {
      const int lr = fast_div(e, dx_magic), i = e - lr * dx;
      const double v0 = xs[(size_t)(lr + H) * dx + i];
      const double v1 = d1[(size_t)(lr + tp.w2) * dx + i];
      const int c = lr + tp.w2;
      double acc = 0;
      for (int k = 1; k <= tp.w2; k++) {
        const double left = d1[(size_t)(c - k) * dx + i];
        const double right = d1[(size_t)(c + k) * dx + i];
        acc += k * (right - left);
      }
      const double v2 = acc / (double)tp.norm2;
      double *o = nrm + (size_t)lr * md + i;
      o[0] = (v0 - (double)nmean[i]) * (double)nscale[i];
      o[dx] = (v1 - (double)nmean[dx + i]) * (double)nscale[dx + i];
      o[2 * dx] = (v2 - (double)nmean[2 * dx + i]) * (double)nscale[2 * dx + i];
    }
    __syncthreads();
    // LinTransformModule::generate (:1243-1269)
    if (tp.matrix) {
      const int npair = (n_seg + 1) / 2;
      for (int it = threadIdx.x; it < ngrp * npair; it += NT) {
        const int rp = fast_div(it, ngrp_magic), g = it - rp * ngrp;
        const int la = 2 * rp, lb = 2 * rp + 1 < n_seg ? 2 * rp + 1 : la;
        const double *xa = nrm + (size_t)la * md, *xb = nrm + (size_t)lb * md;
        const float4 *mg = mt + g;
        double a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
        if (!AASR_FDBG(1)) {
#pragma unroll 3
          for (int j = 0; j < md; j++) {
            const float4 m = mg[j * ngrp];
            const double va = xa[j], vb = xb[j];
            const double m0 = (double)m.x, m1 = (double)m.y, m2 = (double)m.z;
            a0 += m0 * va;
            a1 += m1 * va;
            a2 += m2 * va;
            b0 += m0 * vb;
            b1 += m1 * vb;
            b2 += m2 * vb;
          }
        }
        const int i0 = 3 * g;
        if (tp.bias) {
          const double c0 = (double)nbias[i0];
          const double c1 = i0 + 1 < tp.dim ? (double)nbias[i0 + 1] : 0.0;
          const double c2 = i0 + 2 < tp.dim ? (double)nbias[i0 + 2] : 0.0;
          a0 += c0;
          b0 += c0;
          a1 += c1;
          b1 += c1;
          a2 += c2;
          b2 += c2;
        }
        if (AASR_FDBG(32) && a0 != 12345.0) continue;
        double *da = dst + (r0 + la) * tp.dim + i0;
        da[0] = a0;
        if (i0 + 1 < tp.dim) da[1] = a1;
        if (i0 + 2 < tp.dim) da[2] = a2;
        if (lb != la) {
          double *db = dst + (r0 + lb) * tp.dim + i0;
          db[0] = b0;
          if (i0 + 1 < tp.dim) db[1] = b1;
          if (i0 + 2 < tp.dim) db[2] = b2;
        }
      }
    } else {
      for (int e = threadIdx.x; e < n_seg * tp.dim; e += NT) {
        const int lr = fast_div(e, dim_magic), i = e - lr * tp.dim;
        double acc = nrm[(size_t)lr * md + i];
        if (tp.bias) acc += (double)nbias[i];
        dst[(r0 + lr) * tp.dim + i] = acc;
      }
    }
    __syncthreads();
    r0 = r_end;
  }
}

__global__ void k_mel(DevBatch b, const double *__restrict__ src, SrcMap sm, int span, int64_t rows,
                      int src_dim, int dim, const int32_t *__restrict__ off,
                      const int32_t *__restrict__ tt, const float *__restrict__ sc,
                      const float *__restrict__ sums, int root, double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int bin = (int)(idx - r * dim);
  const double *data = src + src_row(b, r, span, sm) * src_dim;
  float val = 0;
  for (int e = off[bin]; e < off[bin + 1]; e++)
    val = (float)((double)val + (double)sc[e] * data[tt[e]]);
  // float division, evaluated in double and rounded once (identical result)
  float q = (float)((double)val / (double)sums[bin]);
  double o;
  if (root) {
    o = pow((double)q, 0.1);
  } else {
    float a = q + 1.0f;
    o = (double)glibc_logf(a);
  }
  dst[idx] = o;
}

__global__ void k_power(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                        int64_t rows, int src_dim, double *__restrict__ dst) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const double *data = src + src_row(b, r, span, sm) * src_dim;
  float power = 0;
  for (int i = 0; i < src_dim; i++) power = (float)((double)power + data[i]);
  dst[r] = log((double)power + 1e-10);
}

// PowerModule when the source is an FFT module: every input is a float stored in
// a double (FFTModule keeps float results), so (float)((double)power + x) is the
// plain float sum power + (float)x.  One wave per frame: coalesced row load, then
// the reference's left-to-right float accumulation with lane broadcasts.
__global__ __launch_bounds__(256) void k_power_f32src(DevBatch b, const double *__restrict__ src,
                                                      SrcMap sm, int span, int64_t rows,
                                                      int src_dim, double *__restrict__ dst) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const double *data = src + src_row(b, r, span, sm) * src_dim;
  float power = 0;
  for (int base = 0; base < src_dim; base += 64) {
    const float v = (base + lane < src_dim) ? (float)data[base + lane] : 0.0f;
    const int cnt = src_dim - base < 64 ? src_dim - base : 64;
    if (cnt == 64) {
#pragma unroll
      for (int i = 0; i < 64; i++)
        power = power + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
    } else {
      for (int i = 0; i < cnt; i++)
        power = power + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
    }
  }
  if (lane == 0) dst[r] = log((double)power + 1e-10);
}

__global__ void k_dct(DevBatch b, const double *__restrict__ src, SrcMap sm, int span, int64_t rows,
                      int src_dim, int dim, int zeroth, const float *__restrict__ cs,
                      double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int i = (int)(idx - r * dim);
  const double *data = src + src_row(b, r, span, sm) * src_dim;
  double acc = 0.0;
  if (zeroth && i == 0) {
    for (int k = 0; k < src_dim; k++) acc += data[k];
  } else {
    const float *c = cs + (size_t)(i - (zeroth ? 1 : 0)) * src_dim;
    for (int k = 0; k < src_dim; k++) acc += data[k] * (double)c[k];
  }
  dst[idx] = acc;
}

__global__ void k_delta(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                        int64_t rows, int dim, int width, float norm, double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int i = (int)(idx - r * dim);
  int64_t sr = src_row(b, r, span, sm);
  double acc = 0;
  for (int k = 1; k <= width; k++) {
    double left = src[(sr - k) * dim + i];
    double right = src[(sr + k) * dim + i];
    acc += k * (right - left);
  }
  dst[idx] = acc / (double)norm;
}

__global__ void k_normalization(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                                int64_t rows, int dim, const float *__restrict__ mean,
                                const float *__restrict__ scale, double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int i = (int)(idx - r * dim);
  double v = src[src_row(b, r, span, sm) * dim + i];
  dst[idx] = (v - (double)mean[i]) * (double)scale[i];
}

__global__ void k_lin_transform(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                                int64_t rows, int src_dim, int dim,
                                const float *__restrict__ matrix, const float *__restrict__ bias,
                                double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int i = (int)(idx - r * dim);
  const double *data = src + src_row(b, r, span, sm) * src_dim;
  double acc;
  if (matrix) {
    acc = 0;
    const float *mr = matrix + (size_t)i * src_dim;
    for (int j = 0; j < src_dim; j++) acc += (double)mr[j] * data[j];
  } else {
    acc = data[i];
  }
  if (bias) acc += (double)bias[i];
  dst[idx] = acc;
}

// LinTransformModule, tiled: a block stages ROWS source rows and the matrix in
// LDS; thread (row, i) accumulates sum_j matrix[i][j]*src[j] in j order (same
// order and types as the reference loop).
template <int ROWS>
__global__ __launch_bounds__(256) void k_lin_transform_tiled(
    DevBatch b, const double *__restrict__ src, SrcMap sm, int span, int64_t rows, int src_dim,
    int dim, const float *__restrict__ matrix, const float *__restrict__ bias,
    double *__restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *xs = (double *)smem_raw;                  // [ROWS][src_dim]
  float *ms = (float *)(xs + (size_t)ROWS * src_dim);  // [dim][src_dim + 1]
  const int64_t r0 = (int64_t)blockIdx.x * ROWS;
  const int mstride = src_dim | 1;  // odd row stride: threads of a wave read 64 different banks (40 = 8-way conflicts)
  for (int e = threadIdx.x; e < dim * src_dim; e += 256)
    ms[(e / src_dim) * mstride + (e % src_dim)] = matrix[e];
  for (int e = threadIdx.x; e < ROWS * src_dim; e += 256) {
    int64_t r = r0 + e / src_dim;
    xs[e] = r < rows ? src[src_row(b, r, span, sm) * src_dim + (e % src_dim)] : 0.0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < ROWS * dim; e += 256) {
    const int lr = e / dim, i = e - lr * dim;
    const int64_t r = r0 + lr;
    if (r >= rows) continue;
    const double *x = xs + (size_t)lr * src_dim;
    const float *mr = ms + (size_t)i * mstride;
    double acc = 0;
    for (int j = 0; j < src_dim; j++) acc += (double)mr[j] * x[j];
    if (bias) acc += (double)bias[i];
    dst[r * dim + i] = acc;
  }
}

// MeanSubtractorModule, tiled: the block's rows plus the window look-around are staged in LDS once.
// OUT = float: the module is the graph's output and writes the caller's float rows itself (no
// separate narrowing pass).
// A window sum is formed in full only at frames whose ABSOLUTE number (within the utterance) is a multiple of 8 -- from
// single rows at its two ends and sums of 8 consecutive rows (blocks on absolute multiples of 8 too) in between -- and
// slid from there: S(t+1) = (S(t) + x[t+1+right]) - x[t-left].  Every value is a function of the frame's absolute
// number alone, not of how the frame range was tiled or batched; the additions are grouped differently from the
// reference's frame-order loop (1e-15 relative, in double, below the float output's rounding).  The source provides
// kCmsLead rows more of left look-around so that the anchor of an utterance chunk's first frames has its window.
// (One LDS read per row of the window stood behind every output value: 151 reads, then ~21 with the block sums --
// 87 us of the kernel's 195; anchor + slide is 42 reads per 8 values.)
template <int ROWS, class OUT, int NT = 256>
__global__ __launch_bounds__(NT) void k_mean_subtract_tiled(
    DevBatch b, const double *__restrict__ src, SrcMap sm, int span, int halo_left, int64_t rows, int dim,
    int left, int right, int dbg, OUT *__restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int LEAD = kCmsLead;
  double *xs = (double *)smem_raw;  // [ROWS + LEAD + left + right][dim]
  double *bs = xs + (size_t)(ROWS + LEAD + left + right) * dim;  // [(ROWS + LEAD + left + right) / 8 + 1][dim]
  const unsigned dim_magic = fast_magic(dim);              // element indices stay below 65536 (LDS-sized tiles)
  const int W = left + right + 1;
  const int64_t tile0 = (int64_t)blockIdx.x * ROWS;
  const int64_t tile1 = tile0 + ROWS < rows ? tile0 + ROWS : rows;
  // A tile may straddle utterances: source rows of consecutive module rows are
  // consecutive only inside one utterance, so the tile is walked one utterance
  // segment at a time (rows of utterance u: key(u) .. key(u+1)-1 with
  // key(u) = frame_off[u] + u*span).
  int64_t r0 = tile0;
  while (r0 < tile1) {
    const int u = find_utt_near(b, r0, span, rows);
    const int64_t u_end = b.frame_off[u + 1] + (int64_t)(u + 1) * span;
    const int64_t r_end = u_end < tile1 ? u_end : tile1;
    const int64_t s0 = r0 + (int64_t)u * sm.span_diff + sm.shift;  // source row of r0
    const int n_seg = (int)(r_end - r0);
    const int n_src = n_seg + LEAD + left + right;
    for (int e = threadIdx.x; e < (AASR_FDBG(4) ? 0 : n_src * dim); e += NT) xs[e] = src[(s0 - left - LEAD) * dim + e];
    __syncthreads();
    // absolute frame number of output row 0 of the segment; staged row 0 is frame absf0 - left - LEAD
    const int64_t key_u = b.frame_off[u] + (int64_t)u * span;
    const int64_t absf0 = (int64_t)b.first[u] - halo_left + (r0 - key_u);
    const int i0 = (int)(((-(absf0 - left - LEAD)) % 8 + 8) % 8);   // first staged row that starts a block
    const int n_blk = n_src > i0 ? (n_src - i0) / 8 : 0;
    for (int e = threadIdx.x; e < (AASR_FDBG(2) ? 0 : n_blk * dim); e += NT) {
      const int k = fast_div(e, dim_magic), d = e - k * dim;
      const double *c = xs + (size_t)(i0 + 8 * k) * dim + d;
      double t = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) t += c[(size_t)i * dim];
      bs[e] = t;
    }
    __syncthreads();
    // chunks of 8 output rows on absolute multiples of 8: chunk k holds segment rows 8k - off .. 8k - off + 7
    const int off = (int)((absf0 % 8 + 8) % 8);
    const int n_chunk = (n_seg + off + 7) / 8;
    for (int e = threadIdx.x; e < n_chunk * dim; e += NT) {
      const int k = fast_div(e, dim_magic), d = e - k * dim;
      const int la = 8 * k - off;                 // the anchor's segment row (>= -7)
      // staged rows [a, b_end) are the anchor's window
      const int a = la + LEAD, b_end = a + W;
      int ka = a <= i0 ? 0 : (a - i0 + 7) / 8;      // whole blocks [ka, kb): staged rows i0 + 8k ...
      int kb = b_end <= i0 ? 0 : (b_end - i0) / 8;
      if (kb > n_blk) kb = n_blk;
      double sum = 0;
      if (AASR_FDBG(1)) {
      } else if (ka >= kb) {
        for (int i = a; i < b_end; i++) sum += xs[(size_t)i * dim + d];
      } else {
        for (int i = a; i < i0 + 8 * ka; i++) sum += xs[(size_t)i * dim + d];
        for (int kk = ka; kk < kb; kk++) sum += bs[(size_t)kk * dim + d];
        for (int i = i0 + 8 * kb; i < b_end; i++) sum += xs[(size_t)i * dim + d];
      }
      const int l_end = la + 8 < n_seg ? la + 8 : n_seg;
      for (int lr = la; lr < l_end; lr++) {
        if (lr >= 0) {
          const double mean = sum / W;
          dst[(r0 + lr) * dim + d] = (OUT)(xs[(size_t)(lr + LEAD + left) * dim + d] - mean);
        }
        if (lr + 1 < l_end && !AASR_FDBG(1))
          sum = (sum + xs[(size_t)(lr + LEAD + W) * dim + d]) - xs[(size_t)(lr + LEAD) * dim + d];
      }
    }
    __syncthreads();
    r0 = r_end;
  }
}

struct MergeSrc {
  const double *ptr[8];
  SrcMap sm[8];
  int dim[8];
};

__global__ void k_merge(DevBatch b, MergeSrc ms, int span, int64_t rows, int dim,
                        const int32_t *__restrict__ src_col, double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int i = (int)(idx - r * dim);
  int s = src_col[2 * i], c = src_col[2 * i + 1];
  dst[idx] = ms.ptr[s][src_row(b, r, span, ms.sm[s]) * ms.dim[s] + c];
}

__global__ void k_mean_subtract(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                                int64_t rows, int dim, int left, int right,
                                double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int d = (int)(idx - r * dim);
  int64_t sr = src_row(b, r, span, sm);
  double mean = 0;
  for (int i = -left; i <= right; i++) mean += src[(sr + i) * dim + d];
  mean /= (left + right + 1);
  dst[idx] = src[sr * dim + d] - mean;
}

// ConcatModule::generate (aku/FeatureModules.cc:1488-1501): frames t-left..t+right
// of the source side by side
__global__ void k_concat(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                         int64_t rows, int src_dim, int dim, int left, double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int i = (int)(idx - r * dim);
  int k = i / src_dim, j = i - k * src_dim;
  dst[idx] = src[(src_row(b, r, span, sm) + (k - left)) * src_dim + j];
}

// MelPowerModule::generate (aku/FeatureModules.cc:912-923): float running sum of
// double exponentials, natural log
__global__ void k_mel_power(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                            int64_t rows, int src_dim, double *__restrict__ dst) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const double *data = src + src_row(b, r, span, sm) * src_dim;
  float power = 0;
  for (int i = 0; i < src_dim; i++) power = (float)((double)power + exp(data[i]));
  dst[r] = log((double)power + 1e-10);
}

// Windowed interpolation of VtlnModule::generate (sinc / all-pass weights,
// aku/FeatureModules.cc:1913-1925) and SRNormModule::generate (:2045-2057):
// output element (i, d) = max((float) sum_j w[i][j] * src[(start[i]+j)*fd + d], 0).
__global__ void k_window_interp(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                                int64_t rows, int src_dim, int dim, int frame_dim,
                                const int32_t *__restrict__ start, const int32_t *__restrict__ len,
                                const float *__restrict__ coef, int stride,
                                double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int e = (int)(idx - r * dim);
  int i = e / frame_dim, d = e - i * frame_dim;
  const double *data = src + src_row(b, r, span, sm) * src_dim;
  const float *w = coef + (size_t)i * stride;
  double t = 0;
  for (int j = 0, fi = start[i]; j < len[i]; j++, fi++) t += data[fi * frame_dim + d] * (double)w[j];
  const float v = (float)t;
  dst[idx] = (v < 0.0f) ? 0.0f : v;  // std::max((float)t, 0.0f)
}

// VtlnModule::generate without sinc interpolation (:1927-1935)
__global__ void k_vtln_linear(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                              int64_t rows, int dim, const float *__restrict__ bins,
                              double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int bi = (int)(idx - r * dim);
  const double *data = src + src_row(b, r, span, sm) * dim;
  const float pos = bins[bi];
  const float p = ceilf(pos) - pos;
  const float q = 1 - p;
  const double lo = (double)p * data[(int)floorf(pos)];
  const double hi = (double)q * data[(int)ceilf(pos)];
  dst[idx] = lo + hi;
}

// QuantEqModule::generate (aku/FeatureModules.cc:2122-2141); the exponent is
// gamma + (1-alpha)*(x/qmax), the reference's parenthesisation
__global__ void k_quanteq(DevBatch b, const double *__restrict__ src, SrcMap sm, int span,
                          int64_t rows, int dim, const float *__restrict__ alpha,
                          const float *__restrict__ gamma, const float *__restrict__ qmax,
                          double *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * dim) return;
  int64_t r = idx / dim;
  int k = (int)(idx - r * dim);
  const double x = src[src_row(b, r, span, sm) * dim + k];
  if (!alpha) {
    dst[idx] = x;
    return;
  }
  const double ratio = x / (double)qmax[k];
  const float one_minus = 1 - alpha[k];
  const double ex = (double)gamma[k] + (double)one_minus * ratio;
  dst[idx] = (double)qmax[k] * ((double)alpha[k] * pow(ratio, ex));
}

template <class T>
__global__ void k_emit(const double *__restrict__ src, int64_t n, T *__restrict__ dst) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) dst[idx] = (T)src[idx];
}

// ---------------------------------------------------------------- driver ---

static inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }

// AASR_FEAT_FUSE=0 (or aasr_debug_feat_fusion(0)) evaluates every module with its own kernel;
// the fused kernels must reproduce that path bit for bit.
static int g_feat_fusion = AASR_EXPERIMENT_ENV("AASR_FEAT_FUSE") ? atoi(AASR_EXPERIMENT_ENV("AASR_FEAT_FUSE")) : 1;

void feat_run_batch(aasr_feat *h, const int16_t *d_pcm, const UttBatch &ub, int target,
                    float *out_f32, double *out_f64, hipStream_t stream) {
  const int nm = (int)h->mods.size();
  const int n = ub.n_utts;
  if (n <= 0) return;
  const int64_t total = ub.frame_off[n];
  if (total <= 0) return;
  const FeatModule &base = h->mods[0];

  // look-around every needed module must cover
  std::vector<int> L(nm, -1), R(nm, -1), consumers(nm, 0);
  L[target] = R[target] = 0;
  for (int i = target; i >= 0; i--) {
    if (L[i] < 0) continue;
    const FeatModule &m = h->mods[i];
    for (int s : m.sources) {
      L[s] = std::max(L[s], L[i] + m.own_left + m.lead_left);
      R[s] = std::max(R[s], R[i] + m.own_right);
      consumers[s]++;
    }
  }

  // per-utterance descriptors
  std::vector<int32_t> eof(n);
  for (int u = 0; u < n; u++) {
    int64_t ns = ub.pcm_off[u + 1] - ub.pcm_off[u];
    if (base.type == MOD_PRE) {
      if ((ub.pcm_off[u] & 1) || ns / 2 / base.dim < 1)
        raise(AASR_ERR_SHORT_AUDIO, "PreModule: Could not read the file");
      eof[u] = (int32_t)(ns / 2 / base.dim);
      continue;
    }
    if (ns < base.width + 1)
      raise(AASR_ERR_SHORT_AUDIO, "audio shorter than frame");
    if (ns > INT32_MAX)
      raise(AASR_ERR_UNSUPPORTED, "utterance longer than 2^31 samples");
    eof[u] = feat_eof_frame(h, ns);
    if (eof[u] < 1) raise(AASR_ERR_SHORT_AUDIO, "audio shorter than frame");
  }
  // descriptors travel through a pinned staging block owned by the handle; an
  // event guards its reuse so calls can be enqueued back to back
  const size_t stage_bytes = (size_t)(n + 1) * 16 + (size_t)n * 8;
  if (h->stage_event && h->stage_busy) {
    AASR_HIP(hipEventSynchronize(h->stage_event));
    h->stage_busy = false;
  }
  if (stage_bytes > h->stage_cap) {
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    h->stage_host = nullptr;
    AASR_HIP(hipHostMalloc(&h->stage_host, stage_bytes * 2, hipHostMallocDefault));
    h->stage_cap = stage_bytes * 2;
  }
  if (!h->stage_event) AASR_HIP(hipEventCreateWithFlags(&h->stage_event, hipEventDisableTiming));
  {
    char *sp = (char *)h->stage_host;
    int64_t *s_fo = (int64_t *)sp;
    int64_t *s_po = s_fo + (n + 1);
    int32_t *s_first = (int32_t *)(s_po + (n + 1));
    int32_t *s_eof = s_first + n;
    memcpy(s_fo, ub.frame_off.data(), (n + 1) * sizeof(int64_t));
    memcpy(s_po, ub.pcm_off.data(), (n + 1) * sizeof(int64_t));
    memcpy(s_first, ub.first.data(), n * sizeof(int32_t));
    memcpy(s_eof, eof.data(), n * sizeof(int32_t));
    h->d_desc.ensure(stage_bytes);
    AASR_HIP(hipMemcpyAsync(h->d_desc.p, sp, stage_bytes, hipMemcpyHostToDevice, stream));
    AASR_HIP(hipEventRecord(h->stage_event, stream));
    h->stage_busy = true;
  }
  const int64_t *d_fo = (const int64_t *)h->d_desc.p;
  const int32_t *d_first = (const int32_t *)(d_fo + 2 * (size_t)(n + 1));
  DevBatch db{n, d_fo, d_fo + (n + 1), d_first, d_first + n};
  AudioPrm ap{base.width, base.advance, base.emph, base.copy_borders};

  // ---- fusion plan -------------------------------------------------------------------------
  // spectral group: audiofile -> F fft -> {M mel -> D dct, P power} -> G merge [D, P], nothing else
  // reading F, M, D, P and none of them the requested output
  std::vector<char> skip(nm, 0);
  int sgF = -1, sgM = -1, sgD = -1, sgG = -1;
  int tgX = -1, tgA = -1, tgB = -1, tgN = -1, tgT = -1;
  if (g_feat_fusion && base.type == MOD_AUDIOFILE && target != 0 && consumers[0] == 1) {
    for (int gi = 1; gi <= target && sgG < 0; gi++) {
      const FeatModule &G = h->mods[gi];
      if (L[gi] < 0 || G.type != MOD_MERGE || G.sources.size() != 2) continue;
      const int D = G.sources[0], P = G.sources[1];
      if (h->mods[D].type != MOD_DCT || h->mods[P].type != MOD_POWER) continue;
      if (consumers[D] != 1 || consumers[P] != 1 || D == target || P == target) continue;
      const int M = h->mods[D].sources[0], F = h->mods[P].sources[0];
      if (h->mods[M].type != MOD_MEL || h->mods[F].type != MOD_FFT || h->mods[M].sources[0] != F) continue;
      if (consumers[M] != 1 || consumers[F] != 2 || M == target || F == target) continue;
      if (h->mods[F].sources[0] != 0 || G.dim != h->mods[D].dim + 1) continue;
      bool generic_radix = false;  // the fused kernel carries the radix 2/3/4/5 butterflies only
      for (int st = 0; st < h->mods[F].fft.ns; st++) generic_radix = generic_radix || h->mods[F].fft.radix[st] > 5;
      if (generic_radix) continue;
      const SpectralLds lds(h->mods[F].fft.nc, h->mods[M].dim, (int)h->mods[M].mel_t.n,
                            h->mods[D].dim - (h->mods[D].zeroth ? 1 : 0));
      if (lds.total > 64 * 1024) continue;
      sgF = F; sgM = M; sgD = D; sgG = gi;
      skip[F] = skip[M] = skip[P] = skip[D] = 1;
    }
  }
  // temporal group: X -> A delta -> B delta, C merge [X, A, B] -> N normalization -> T lin_transform
  if (g_feat_fusion) {
    for (int ti = 1; ti <= target && tgT < 0; ti++) {
      const FeatModule &T = h->mods[ti];
      if (L[ti] < 0 || T.type != MOD_LIN_TRANSFORM) continue;
      const int N = T.sources[0];
      if (h->mods[N].type != MOD_NORMALIZATION || consumers[N] != 1 || N == target) continue;
      const int C = h->mods[N].sources[0];
      const FeatModule &Cm = h->mods[C];
      if (Cm.type != MOD_MERGE || Cm.sources.size() != 3 || consumers[C] != 1 || C == target) continue;
      const int X = Cm.sources[0], A = Cm.sources[1], B = Cm.sources[2];
      if (h->mods[A].type != MOD_DELTA || h->mods[B].type != MOD_DELTA) continue;
      if (h->mods[A].sources[0] != X || h->mods[B].sources[0] != A) continue;
      if (consumers[A] != 2 || consumers[B] != 1 || consumers[X] != 2 || A == target || B == target) continue;
      const int dx = h->mods[X].dim;
      if (h->mods[A].dim != dx || h->mods[B].dim != dx || Cm.dim != 3 * dx || T.src_dim != 3 * dx) continue;
      if ((int)h->mods[N].mean.size() != 3 * dx || (int)h->mods[N].scale.size() != 3 * dx) continue;
      const int H = h->mods[A].delta_width + h->mods[B].delta_width;
      constexpr int TR = kTemporalRows;
      if (temporal_smem(TR, dx, H, h->mods[B].delta_width, T.dim) > 64 * 1024) continue;
      tgX = X; tgA = A; tgB = B; tgN = N; tgT = ti;
      skip[A] = skip[B] = skip[C] = skip[N] = 1;
    }
  }

  auto rows_of = [&](int i) { return total + (int64_t)n * (L[i] + R[i]); };
  auto map_of = [&](int i, int s) {
    return SrcMap{(L[s] + R[s]) - (L[i] + R[i]), L[s] - L[i]};
  };

  bool emitted = false;  // the output module wrote the caller's rows itself
  for (int i = 0; i <= target; i++) {
    if (L[i] < 0 || skip[i]) continue;
    FeatModule &m = h->mods[i];
    const int64_t rows = rows_of(i);
    const int span = L[i] + R[i];
    if (i == sgG) {
      const FeatModule &F = h->mods[sgF], &M = h->mods[sgM], &D = h->mods[sgD];
      SpectralPrm sp;
      sp.fp.nc = F.fft.nc;
      sp.fp.ns = F.fft.ns;
      for (int k = 0; k < 16; k++) {
        sp.fp.radix[k] = F.fft.radix[k];
        sp.fp.sublen[k] = F.fft.sublen[k];
        sp.fp.sub_magic[k] = fast_magic(F.fft.sublen[k]);
        sp.fp.fstride[k] = k < F.fft.ns ? F.fft.nc / (F.fft.radix[k] * F.fft.sublen[k]) : 0;
      }
      sp.fp.hamming = F.fft.hamming.p;
      sp.fp.twiddle = F.fft.twiddle.p;
      sp.fp.stwiddle = F.fft.stwiddle.p;
      sp.fp.perm = F.fft.perm.p;
      sp.fp.magnitude = F.magnitude;
      sp.fp.take_log = F.take_log;
      sp.mel_dim = M.dim;
      sp.mel_root = M.root;
      sp.mel_off = M.mel_off.p;
      sp.mel_t = M.mel_t.p;
      sp.mel_scale = M.mel_scale.p;
      sp.mel_sum = M.mel_sum.p;
      sp.mel_order = M.mel_order.p;
      sp.dct_dim = D.dim;
      sp.zeroth = D.zeroth;
      sp.dct_cos = D.dct_cos.p;
      sp.mel_terms = (int)M.mel_t.n;
      const SpectralLds lds(F.fft.nc, M.dim, sp.mel_terms, D.dim - (D.zeroth ? 1 : 0));
      const int64_t nblk = (rows + SPEC_FRAMES - 1) / SPEC_FRAMES;
      // a workgroup stages the tables once and walks `passes` blocks of 16 frames; enough
      // workgroups remain to fill the chip several times over
      const int passes = (int)std::max<int64_t>(1, std::min<int64_t>(8, nblk / 4096));
      h->bufs[i].ensure((size_t)rows * m.dim);
      hipLaunchKernelGGL(k_spectral_fused, dim3((unsigned)((nblk + passes - 1) / passes)), dim3(256), lds.total,
                         stream, db, d_pcm, ap, L[i], R[i], rows, sp, passes,
                         AASR_EXPERIMENT_ENV("AASR_SPEC_DBG") ? atoi(AASR_EXPERIMENT_ENV("AASR_SPEC_DBG")) : 0, h->bufs[i].p);
      AASR_HIP(hipGetLastError());
      continue;
    }
    if (i == tgT) {
      const FeatModule &A = h->mods[tgA], &B = h->mods[tgB], &N = h->mods[tgN];
      TemporalPrm tp;
      tp.dx = h->mods[tgX].dim;
      tp.w1 = A.delta_width;
      tp.w2 = B.delta_width;
      tp.norm1 = A.delta_norm;
      tp.norm2 = B.delta_norm;
      tp.mean = N.d_mean.p;
      tp.scale = N.d_scale.p;
      tp.matrix = m.matrix_defined ? m.d_matrix.p : nullptr;
      tp.bias = m.bias_defined ? m.d_bias.p : nullptr;
      tp.dim = m.dim;
      constexpr int TR = kTemporalRows;
      const int H = tp.w1 + tp.w2;
      const size_t smem = temporal_smem(TR, tp.dx, H, tp.w2, m.dim);
      h->bufs[i].ensure((size_t)rows * m.dim);
      constexpr int TNT = TR >= 128 ? 1024 : TR >= 64 ? 512 : 256;
      if (smem > 64 * 1024)
        AASR_HIP(hipFuncSetAttribute((const void *)k_temporal_fused<TR, TNT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL((k_temporal_fused<TR, TNT>), dim3((unsigned)((rows + TR - 1) / TR)), dim3(TNT), smem, stream, db,
                         (const double *)h->bufs[tgX].p, map_of(i, tgX), span, rows, tp,
                         AASR_EXPERIMENT_ENV("AASR_TEMP_DBG") ? atoi(AASR_EXPERIMENT_ENV("AASR_TEMP_DBG")) : 0, h->bufs[i].p);
      AASR_HIP(hipGetLastError());
      continue;
    }
    // audio frames are produced inside the FFT kernel when nothing else reads them
    if (m.type == MOD_AUDIOFILE && i != target && consumers[0] == 1) {
      bool only_fft = false;
      for (int j = 1; j <= target; j++)
        if (L[j] >= 0 && h->mods[j].type == MOD_FFT && h->mods[j].sources[0] == 0) only_fft = true;
      if (only_fft) continue;
    }
    h->bufs[i].ensure((size_t)rows * m.dim);
    double *dst = h->bufs[i].p;
    const int s0 = m.sources.empty() ? -1 : m.sources[0];
    const double *src = s0 >= 0 ? h->bufs[s0].p : nullptr;
    SrcMap sm = s0 >= 0 ? map_of(i, s0) : SrcMap{0, 0};
    const int64_t nelem = rows * m.dim;
    switch (m.type) {
      case MOD_PRE:
        hipLaunchKernelGGL(k_pre_frames, dim3(grid_for(nelem)), dim3(256), 0, stream, db,
                           (const float *)d_pcm, m.dim, L[i], R[i], rows, dst);
        break;
      case MOD_AUDIOFILE:
        hipLaunchKernelGGL(k_audio_frames, dim3(grid_for(nelem)), dim3(256), 0, stream, db, d_pcm,
                           ap, L[i], R[i], rows, dst);
        break;
      case MOD_FFT: {
        FftPrm fp;
        fp.nc = m.fft.nc;
        fp.ns = m.fft.ns;
        for (int k = 0; k < 16; k++) {
          fp.radix[k] = m.fft.radix[k];
          fp.sublen[k] = m.fft.sublen[k];
          fp.sub_magic[k] = fast_magic(m.fft.sublen[k]);
          fp.fstride[k] = k < m.fft.ns ? m.fft.nc / (m.fft.radix[k] * m.fft.sublen[k]) : 0;
        }
        fp.hamming = m.fft.hamming.p;
        fp.twiddle = m.fft.twiddle.p;
        fp.stwiddle = m.fft.stwiddle.p;
        fp.perm = m.fft.perm.p;
        fp.magnitude = m.magnitude;
        fp.take_log = m.take_log;
        const size_t smem = (size_t)4 * fp.nc * sizeof(float) * 2;
        const unsigned blocks = (unsigned)((rows + 3) / 4);
        bool gen = false;
        for (int st = 0; st < fp.ns; st++) gen = gen || fp.radix[st] > 5;
        const bool from_pcm = s0 == 0 && consumers[0] == 1 && target != 0;
        const double *fsrc = from_pcm ? (const double *)nullptr : src;
        if (from_pcm && gen)
          hipLaunchKernelGGL((k_fft<true, true>), dim3(blocks), dim3(256), smem, stream, db, d_pcm, ap, fsrc, sm, L[i],
                             R[i], rows, fp, dst);
        else if (from_pcm)
          hipLaunchKernelGGL((k_fft<true, false>), dim3(blocks), dim3(256), smem, stream, db, d_pcm, ap, fsrc, sm, L[i],
                             R[i], rows, fp, dst);
        else if (gen)
          hipLaunchKernelGGL((k_fft<false, true>), dim3(blocks), dim3(256), smem, stream, db, d_pcm, ap, fsrc, sm, L[i],
                             R[i], rows, fp, dst);
        else
          hipLaunchKernelGGL((k_fft<false, false>), dim3(blocks), dim3(256), smem, stream, db, d_pcm, ap, fsrc, sm, L[i],
                             R[i], rows, fp, dst);
        break;
      }
      case MOD_MEL:
        hipLaunchKernelGGL(k_mel, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src, sm, span,
                           rows, h->mods[s0].dim, m.dim, m.mel_off.p, m.mel_t.p, m.mel_scale.p,
                           m.mel_sum.p, m.root, dst);
        break;
      case MOD_POWER:
        if (h->mods[s0].type == MOD_FFT)
          hipLaunchKernelGGL(k_power_f32src, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream,
                             db, src, sm, span, rows, h->mods[s0].dim, dst);
        else
          hipLaunchKernelGGL(k_power, dim3(grid_for(rows)), dim3(256), 0, stream, db, src, sm, span,
                             rows, h->mods[s0].dim, dst);
        break;
      case MOD_DCT:
        hipLaunchKernelGGL(k_dct, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src, sm, span,
                           rows, h->mods[s0].dim, m.dim, m.zeroth, m.dct_cos.p, dst);
        break;
      case MOD_DELTA:
        hipLaunchKernelGGL(k_delta, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src, sm, span,
                           rows, m.dim, m.delta_width, m.delta_norm, dst);
        break;
      case MOD_NORMALIZATION:
        hipLaunchKernelGGL(k_normalization, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src,
                           sm, span, rows, m.dim, m.d_mean.p, m.d_scale.p, dst);
        break;
      case MOD_LIN_TRANSFORM: {
        constexpr int LT_ROWS = 64;
        const size_t lt_smem = (size_t)LT_ROWS * m.src_dim * 8 + (size_t)m.dim * (m.src_dim + 1) * 4;
        if (m.matrix_defined && lt_smem <= 60 * 1024)
          hipLaunchKernelGGL(k_lin_transform_tiled<LT_ROWS>, dim3((unsigned)((rows + LT_ROWS - 1) / LT_ROWS)),
                             dim3(256), lt_smem, stream, db, src, sm, span, rows, m.src_dim, m.dim,
                             m.d_matrix.p, m.bias_defined ? m.d_bias.p : (const float *)nullptr, dst);
        else
          hipLaunchKernelGGL(k_lin_transform, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src,
                             sm, span, rows, m.src_dim, m.dim,
                             m.matrix_defined ? m.d_matrix.p : (const float *)nullptr,
                             m.bias_defined ? m.d_bias.p : (const float *)nullptr, dst);
        break;
      }
      case MOD_MERGE: {
        if (m.sources.size() > 8)
          raise(AASR_ERR_UNSUPPORTED, "merge module with more than 8 sources");
        MergeSrc ms;
        for (size_t k = 0; k < 8; k++) {
          ms.ptr[k] = nullptr;
          ms.sm[k] = SrcMap{0, 0};
          ms.dim[k] = 0;
        }
        for (size_t k = 0; k < m.sources.size(); k++) {
          ms.ptr[k] = h->bufs[m.sources[k]].p;
          ms.sm[k] = map_of(i, m.sources[k]);
          ms.dim[k] = h->mods[m.sources[k]].dim;
        }
        hipLaunchKernelGGL(k_merge, dim3(grid_for(nelem)), dim3(256), 0, stream, db, ms, span, rows,
                           m.dim, m.merge_src_col.p, dst);
        break;
      }
      case MOD_CONCAT:
        hipLaunchKernelGGL(k_concat, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src, sm, span,
                           rows, m.src_dim, m.dim, m.own_left, dst);
        break;
      case MOD_MEL_POWER:
        hipLaunchKernelGGL(k_mel_power, dim3(grid_for(rows)), dim3(256), 0, stream, db, src, sm,
                           span, rows, h->mods[s0].dim, dst);
        break;
      case MOD_VTLN:
        if (m.sp_stride > 0)
          hipLaunchKernelGGL(k_window_interp, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src,
                             sm, span, rows, m.dim, m.dim, 1, m.sp_start.p, m.sp_len.p, m.sp_coef.p,
                             m.sp_stride, dst);
        else
          hipLaunchKernelGGL(k_vtln_linear, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src, sm,
                             span, rows, m.dim, m.d_vtln_bins.p, dst);
        break;
      case MOD_SR_NORM:
        hipLaunchKernelGGL(k_window_interp, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src, sm,
                           span, rows, h->mods[s0].dim, m.dim, m.frame_dim, m.sp_start.p, m.sp_len.p,
                           m.sp_coef.p, m.sp_stride, dst);
        break;
      case MOD_QUANTEQ: {
        const bool full = !m.q_alpha.empty() && !m.q_gamma.empty() && !m.q_max.empty();
        hipLaunchKernelGGL(k_quanteq, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src, sm, span,
                           rows, m.dim, full ? m.d_q_alpha.p : (const float *)nullptr, m.d_q_gamma.p,
                           m.d_q_max.p, dst);
        break;
      }
      case MOD_HOST: {
        // A user-registered type (aasr_feat_register_module_type): its frames are computed by the
        // user's callback on the host.  Sources come down, results go up; everything enqueued so far
        // has to finish first.  Rows of utterance u: key(u) .. key(u+1)-1 with key(u) = frame_off[u] +
        // u*span; row r is frame first[u] - L + (r - key(u)), its source row r + u*span_diff + shift.
        const HostModuleType &ht = host_module_types()[(size_t)m.host_type];
        AASR_HIP(hipStreamSynchronize(stream));
        const size_t ns = m.sources.size();
        std::vector<std::vector<double>> hsrc(ns);
        std::vector<SrcMap> smap(ns);
        std::vector<int> sdim(ns);
        for (size_t k = 0; k < ns; k++) {
          const int sidx = m.sources[k];
          sdim[k] = h->mods[sidx].dim;
          smap[k] = map_of(i, sidx);
          hsrc[k].resize((size_t)rows_of(sidx) * sdim[k]);
          AASR_HIP(hipMemcpy(hsrc[k].data(), h->bufs[sidx].p, hsrc[k].size() * sizeof(double), hipMemcpyDeviceToHost));
        }
        std::vector<double> hout((size_t)rows * m.dim);
        std::vector<const double *> ptr(ns);
        char err[512] = {0};
        for (int u = 0; u < n; u++) {
          const int64_t key = ub.frame_off[u] + (int64_t)u * span;
          const int64_t n_rows = (ub.frame_off[u + 1] - ub.frame_off[u]) + span;
          for (int64_t lr = 0; lr < n_rows; lr++) {
            const int64_t r = key + lr;
            const int frame = ub.first[u] - L[i] + (int)lr;
            for (size_t k = 0; k < ns; k++) {
              const int64_t sr = r + (int64_t)u * smap[k].span_diff + smap[k].shift;
              ptr[k] = hsrc[k].data() + (size_t)(sr - m.own_left) * sdim[k];
            }
            if (ht.vtbl.generate(m.host_instance, frame, ptr.data(), hout.data() + (size_t)r * m.dim, err,
                                 (int32_t)sizeof err - 1) != 0)
              raise(AASR_ERR_INVALID, "module %s: %s", m.name.c_str(), err[0] ? err : "generate() failed");
          }
        }
        AASR_HIP(hipMemcpy(dst, hout.data(), hout.size() * sizeof(double), hipMemcpyHostToDevice));
        break;
      }
      case MOD_MEAN_SUBTRACTOR: {
        const int ms_dbg = AASR_EXPERIMENT_ENV("AASR_CMS_DBG") ? atoi(AASR_EXPERIMENT_ENV("AASR_CMS_DBG")) : 0;
        // tile: 128 rows x 512 threads where two such workgroups fit a CU's LDS (the look-around is re-read once per
        // tile: 2.3x the rows at 64, 1.6x at 128 -- 134 -> 90 us on the production graph), 64 x 256 where three of
        // those fit, 128 x 512 alone on a CU for wide windows (the reference's default 75 + 75 at 39 columns: 100 KB);
        // the untiled kernel beyond that
        auto smem_of = [&](int tile_rows) {
          const size_t n = (size_t)(tile_rows + kCmsLead + m.cms_left + m.cms_right);
          return (n + n / 8 + 1) * m.dim * 8;
        };
        int tile_rows = 0;
        if (smem_of(128) <= 78 * 1024) tile_rows = 128;
        else if (smem_of(64) <= 52 * 1024) tile_rows = 64;
        else if (smem_of(128) <= 156 * 1024) tile_rows = 128;
        else if (smem_of(64) <= 156 * 1024) tile_rows = 64;
        const bool emit_f32 = i == target && g_feat_fusion && out_f32 && !out_f64;
        auto launch = [&](auto kern, int rows_per, int nt, auto *out) {
          const size_t smem = smem_of(rows_per);
          if (smem > 64 * 1024)
            AASR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
          hipLaunchKernelGGL(kern, dim3((unsigned)((rows + rows_per - 1) / rows_per)), dim3(nt), smem, stream, db, src, sm,
                             span, L[i], rows, m.dim, m.cms_left, m.cms_right, ms_dbg, out);
        };
        if (tile_rows >= 128) {
          // the output module: its rows are the caller's rows (no look-around of its own)
          if (emit_f32) launch(k_mean_subtract_tiled<128, float, 512>, 128, 512, out_f32);
          else launch(k_mean_subtract_tiled<128, double, 512>, 128, 512, dst);
          emitted = emit_f32;
        } else if (tile_rows > 0) {
          if (emit_f32) launch(k_mean_subtract_tiled<64, float, 256>, 64, 256, out_f32);
          else launch(k_mean_subtract_tiled<64, double, 256>, 64, 256, dst);
          emitted = emit_f32;
        } else {
          hipLaunchKernelGGL(k_mean_subtract, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src,
                             sm, span, rows, m.dim, m.cms_left, m.cms_right, dst);
        }
        break;
      }
    }
    AASR_HIP(hipGetLastError());
  }
  const int64_t nout = total * h->mods[target].dim;
  if (emitted) return;
  if (out_f32)
    hipLaunchKernelGGL(k_emit<float>, dim3(grid_for(nout)), dim3(256), 0, stream,
                       (const double *)h->bufs[target].p, nout, out_f32);
  if (out_f64)
    hipLaunchKernelGGL(k_emit<double>, dim3(grid_for(nout)), dim3(256), 0, stream,
                       (const double *)h->bufs[target].p, nout, out_f64);
  AASR_HIP(hipGetLastError());
}

}  // namespace aasr

// Diagnostic: 0 = every feature module runs its own kernel, 1 = fused kernels where the graph
// has the production shape (default).  The two paths give identical bits.
extern "C" void aasr_debug_feat_fusion(int on) { aasr::g_feat_fusion = on; }
