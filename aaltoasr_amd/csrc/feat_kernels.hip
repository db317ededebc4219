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
  const int32_t *eof_frame;  // [n] last_frame()+1
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

struct FftPrm {
  int nc, ns;
  int radix[16], sublen[16];
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

// One wave per frame: Hamming -> packed half-length complex FFT in LDS with
// KissFFT's butterfly order -> real split -> |X|^2 (/ sqrt / log).
template <bool FROM_PCM>
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
    const int fstride = nc / (pr * m);
    const int nb = nc / pr;
    for (int bi = lane; bi < nb; bi += 64) {
      int g = bi / m, j = bi - g * m;
      cpx *F = buf + g * pr * m;
      if (pr == 2) {
        cpx t = cmul(F[m + j], tw[j * fstride]);
        cpx f0 = F[j];
        cpx o1;
        o1.r = f0.r - t.r;
        o1.i = f0.i - t.i;
        f0.r += t.r;
        f0.i += t.i;
        F[m + j] = o1;
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
        cpx s0 = cmul(F[m + j], tw[j * fstride]);
        cpx s1 = cmul(F[2 * m + j], tw[2 * j * fstride]);
        cpx s2 = cmul(F[3 * m + j], tw[3 * j * fstride]);
        cpx f0 = F[j], s3, s4, s5, o1, o2, o3;
        s5.r = f0.r - s1.r;  s5.i = f0.i - s1.i;
        f0.r += s1.r;        f0.i += s1.i;
        s3.r = s0.r + s2.r;  s3.i = s0.i + s2.i;
        s4.r = s0.r - s2.r;  s4.i = s0.i - s2.i;
        o2.r = f0.r - s3.r;  o2.i = f0.i - s3.i;
        f0.r += s3.r;        f0.i += s3.i;
        o1.r = s5.r + s4.i;  o1.i = s5.i - s4.r;
        o3.r = s5.r - s4.i;  o3.i = s5.i + s4.r;
        F[j] = f0;
        F[m + j] = o1;
        F[2 * m + j] = o2;
        F[3 * m + j] = o3;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // real split (kiss_fftr.c:86-120) + power spectrum (FeatureModules.cc:533-565)
  double *out = dst + r * (int64_t)(nc + 1);
  auto emit = [&](int k, float re, float im) {
    float a = re * re;
    float c = im * im;
    float v = a + c;
    if (fp.magnitude) v = (float)sqrt((double)v);  // == correctly rounded sqrtf
    if (fp.take_log) v = (float)log((double)v);
    out[k] = (double)v;
  };
  const cpx *stw = (const cpx *)fp.stwiddle;
  if (lane == 0) {
    cpx t0 = buf[0];
    emit(0, t0.r + t0.i, 0.0f);
    emit(nc, t0.r - t0.i, 0.0f);
  }
  for (int k = 1 + lane; k <= nc / 2; k += 64) {
    cpx fpk = buf[k], fpnk, f1k, f2k, t;
    fpnk.r = buf[nc - k].r;
    fpnk.i = -buf[nc - k].i;
    f1k.r = fpk.r + fpnk.r;  f1k.i = fpk.i + fpnk.i;
    f2k.r = fpk.r - fpnk.r;  f2k.i = fpk.i - fpnk.i;
    t = cmul(f2k, stw[k - 1]);
    float ar = f1k.r + t.r, ai = f1k.i + t.i, br = f1k.r - t.r, bi = t.i - f1k.i;
    float re_k = (float)((double)ar * .5), im_k = (float)((double)ai * .5);
    float re_n = (float)((double)br * .5), im_n = (float)((double)bi * .5);
    if (k != nc - k) emit(k, re_k, im_k);
    emit(nc - k, re_n, im_n);  // for k == nc-k the later assignment wins
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
    o = (double)(float)log((double)a);
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

// MeanSubtractorModule, tiled: the block's rows plus the window look-around are
// staged in LDS once; each thread sums its window in frame order.
template <int ROWS>
__global__ __launch_bounds__(256) void k_mean_subtract_tiled(
    DevBatch b, const double *__restrict__ src, SrcMap sm, int span, int halo_left, int64_t rows, int dim,
    int left, int right, double *__restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *xs = (double *)smem_raw;  // [ROWS + left + right][dim]
  // sums of 8 consecutive frames: a window of left+right+1 rows is then a few singles at its ends
  // plus whole blocks (151 LDS reads per value became ~30; the kernel was LDS-bandwidth bound).
  // The blocks sit on ABSOLUTE frame numbers (multiples of 8 within the utterance), so a frame's
  // value does not depend on how the frame range was tiled or batched; the additions are grouped
  // differently from the frame-order loop (1e-16 relative).
  double *bs = xs + (size_t)(ROWS + left + right) * dim;  // [(ROWS + left + right) / 8 + 1][dim]
  const int64_t tile0 = (int64_t)blockIdx.x * ROWS;
  const int64_t tile1 = tile0 + ROWS < rows ? tile0 + ROWS : rows;
  // A tile may straddle utterances: source rows of consecutive module rows are
  // consecutive only inside one utterance, so the tile is walked one utterance
  // segment at a time (rows of utterance u: key(u) .. key(u+1)-1 with
  // key(u) = frame_off[u] + u*span).
  int64_t r0 = tile0;
  while (r0 < tile1) {
    const int u = find_utt(b, r0, span);
    const int64_t u_end = b.frame_off[u + 1] + (int64_t)(u + 1) * span;
    const int64_t r_end = u_end < tile1 ? u_end : tile1;
    const int64_t s0 = r0 + (int64_t)u * sm.span_diff + sm.shift;  // source row of r0
    const int n_seg = (int)(r_end - r0);
    const int n_src = n_seg + left + right;
    for (int e = threadIdx.x; e < n_src * dim; e += 256) xs[e] = src[(s0 - left) * dim + e];
    __syncthreads();
    // frame number of staged row 0, and the first staged row that starts a block
    const int64_t key_u = b.frame_off[u] + (int64_t)u * span;
    const int64_t abs0 = (int64_t)b.first[u] - halo_left + (r0 - key_u) - left;
    const int i0 = (int)(((-abs0) % 8 + 8) % 8);
    const int n_blk = n_src > i0 ? (n_src - i0) / 8 : 0;
    for (int e = threadIdx.x; e < n_blk * dim; e += 256) {
      const int k = e / dim, d = e - k * dim;
      const double *c = xs + (size_t)(i0 + 8 * k) * dim + d;
      double t = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) t += c[(size_t)i * dim];
      bs[e] = t;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_seg * dim; e += 256) {
      const int lr = e / dim, d = e - lr * dim;
      // staged rows [lr, lr + left + right] are this value's window
      const int a = lr, b_end = lr + left + right + 1;
      int ka = a <= i0 ? 0 : (a - i0 + 7) / 8;      // whole blocks [ka, kb): staged rows i0 + 8k ...
      int kb = b_end <= i0 ? 0 : (b_end - i0) / 8;
      if (kb > n_blk) kb = n_blk;
      double mean = 0;
      if (ka >= kb) {
        for (int i = a; i < b_end; i++) mean += xs[(size_t)i * dim + d];
      } else {
        for (int i = a; i < i0 + 8 * ka; i++) mean += xs[(size_t)i * dim + d];
        for (int k = ka; k < kb; k++) mean += bs[(size_t)k * dim + d];
        for (int i = i0 + 8 * kb; i < b_end; i++) mean += xs[(size_t)i * dim + d];
      }
      mean /= (left + right + 1);
      dst[(r0 + lr) * dim + d] = xs[(size_t)(lr + left) * dim + d] - mean;
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
      L[s] = std::max(L[s], L[i] + m.own_left);
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
    eof[u] = feat_last_frame(h, ns) + 1;
    if (eof[u] < 1) raise(AASR_ERR_SHORT_AUDIO, "audio shorter than frame");
  }
  h->d_frame_off.ensure(n + 1);
  h->d_pcm_off.ensure(n + 1);
  h->d_first.ensure(n);
  h->d_eof.ensure(n);
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
    AASR_HIP(hipMemcpyAsync(h->d_frame_off.p, s_fo, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    AASR_HIP(hipMemcpyAsync(h->d_pcm_off.p, s_po, (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    AASR_HIP(hipMemcpyAsync(h->d_first.p, s_first, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    AASR_HIP(hipMemcpyAsync(h->d_eof.p, s_eof, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    AASR_HIP(hipEventRecord(h->stage_event, stream));
    h->stage_busy = true;
  }
  DevBatch db{n, h->d_frame_off.p, h->d_pcm_off.p, h->d_first.p, h->d_eof.p};
  AudioPrm ap{base.width, base.advance, base.emph, base.copy_borders};

  auto rows_of = [&](int i) { return total + (int64_t)n * (L[i] + R[i]); };
  auto map_of = [&](int i, int s) {
    return SrcMap{(L[s] + R[s]) - (L[i] + R[i]), L[s] - L[i]};
  };

  for (int i = 0; i <= target; i++) {
    if (L[i] < 0) continue;
    FeatModule &m = h->mods[i];
    const int64_t rows = rows_of(i);
    const int span = L[i] + R[i];
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
        }
        fp.hamming = m.fft.hamming.p;
        fp.twiddle = m.fft.twiddle.p;
        fp.stwiddle = m.fft.stwiddle.p;
        fp.perm = m.fft.perm.p;
        fp.magnitude = m.magnitude;
        fp.take_log = m.take_log;
        const size_t smem = (size_t)4 * fp.nc * sizeof(float) * 2;
        const unsigned blocks = (unsigned)((rows + 3) / 4);
        if (s0 == 0 && consumers[0] == 1 && target != 0) {
          hipLaunchKernelGGL(k_fft<true>, dim3(blocks), dim3(256), smem, stream, db, d_pcm, ap,
                             (const double *)nullptr, sm, L[i], R[i], rows, fp, dst);
        } else {
          hipLaunchKernelGGL(k_fft<false>, dim3(blocks), dim3(256), smem, stream, db, d_pcm, ap,
                             src, sm, L[i], R[i], rows, fp, dst);
        }
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
      case MOD_MEAN_SUBTRACTOR: {
        constexpr int MS_ROWS = 64;
        const size_t ms_src = (size_t)(MS_ROWS + m.cms_left + m.cms_right);
        const size_t ms_smem = (ms_src + ms_src / 8 + 1) * m.dim * 8;
        if (ms_smem <= 60 * 1024)
          hipLaunchKernelGGL(k_mean_subtract_tiled<MS_ROWS>, dim3((unsigned)((rows + MS_ROWS - 1) / MS_ROWS)),
                             dim3(256), ms_smem, stream, db, src, sm, span, L[i], rows, m.dim, m.cms_left,
                             m.cms_right, dst);
        else
          hipLaunchKernelGGL(k_mean_subtract, dim3(grid_for(nelem)), dim3(256), 0, stream, db, src,
                             sm, span, rows, m.dim, m.cms_left, m.cms_right, dst);
        break;
      }
    }
    AASR_HIP(hipGetLastError());
  }
  const int64_t nout = total * h->mods[target].dim;
  if (out_f32)
    hipLaunchKernelGGL(k_emit<float>, dim3(grid_for(nout)), dim3(256), 0, stream,
                       (const double *)h->bufs[target].p, nout, out_f32);
  if (out_f64)
    hipLaunchKernelGGL(k_emit<double>, dim3(grid_for(nout)), dim3(256), 0, stream,
                       (const double *)h->bufs[target].p, nout, out_f64);
  AASR_HIP(hipGetLastError());
}

}  // namespace aasr
