// lna_device.h -- device helpers of the LNA normalisation / packing shared by lna_encode.hip and
// the fused cluster merge + LNA kernel of gmm_cluster.hip (see lna_encode.hip for the reference
// lines they restate).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace aasr {

#define LN2_D 0.69314718055994530942
#define LOG_TINY_D (-115.12925464970228420090)  // log(1e-50)
#define LN_FLT_MIN_F (-87.33654475f)             // ln(2^-126)

// log of (float)exp(ll) as the reference stores it; -inf when it flushes to 0.
// Inside the denormal band the stored value is q * 2^-149 with
// q = rint(2^149 * e^ll) in [1, 2^23].  q is computed in float: the product
// ll * log2(e) is carried as hi + lo (error-free split), 2^(hi + 149) comes from
// v_exp_f32 (the addition is exact) and lo is applied as a first-order factor, so
// q is right except within ~2 float ulps of a rounding tie -- at most one quantum,
// i.e. <= 2.4e-7 relative for q > 2^22 and exact for small q.  log(q) is taken in
// float (error < 2e-6); the band is wave-divergent and frequent for weak states,
// so a double exp/log pair here used to dominate the kernel.
__device__ __forceinline__ double float_cast_loglik(float ll) {
  if (ll >= LN_FLT_MIN_F) return (double)ll;
  if (ll < -103.98f) return -INFINITY;  // 2^149 e^ll < 0.5: rounds to zero
  const float L1 = 1.44269502162933349609375f;   // float(log2 e)
  const float L2 = 1.92596303350001e-08f;        // log2 e - L1
  const float hi = ll * L1;
  const float lo = fmaf(ll, L1, -hi) + ll * L2;
  const float t = __builtin_amdgcn_exp2f(hi + 149.0f) * (1.0f + lo * 0.69314718f);
  const float q = rintf(t);
  if (q <= 0.0f) return -INFINITY;
  return (double)(__builtin_amdgcn_logf(q) * 0.69314718f) - 149.0 * LN2_D;
}

template <class T>
__device__ __forceinline__ T wave_reduce_max(T v) {
  for (int o = 32; o > 0; o >>= 1) {
    T u = __shfl_xor(v, o, 64);
    v = u > v ? u : v;
  }
  return v;
}
__device__ __forceinline__ double wave_reduce_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void lna_store(float lp, int lnabytes, int64_t o,
                                          float *__restrict__ lp_out,
                                          uint8_t *__restrict__ bytes_out) {
  if (lp_out) lp_out[o] = lp;
  if (bytes_out) {
    if (lnabytes == 4) {
      ((float *)bytes_out)[o] = lp;
    } else {
      unsigned short code;
      // (double)lp < -36.008  <=>  lp < (float)-36.008: the float nearest to
      // -36.008 lies on its zero side, so no float falls between the two
      if (lp < -36.008f) {
        code = 0xffff;
      } else {
        int temp = (int)(-1820.0 * (double)lp + .5);
        unsigned b0 = (temp >> 8) & 255, b1 = temp & 255;
        code = (unsigned short)(b0 | (b1 << 8));  // big-endian on disk
      }
      ((unsigned short *)bytes_out)[o] = code;
    }
  }
}

// Values stay float in registers; only elements inside the reference's float
// denormal band (rare) take the double-precision quantisation detour.
__device__ __forceinline__ float cast_f(float ll) {
  return ll >= LN_FLT_MIN_F ? ll : (float)float_cast_loglik(ll);
}


// One frame's normalisation and packing for a group of 256 threads that hold the row in
// registers (v[j] = state tid + 256 j, -inf beyond S): max, Z / e^max in double, log, pack.
// `red` is the group's 8 doubles of LDS.  ONE group per workgroup: the number of __syncthreads() a call passes depends
// on the frame (4 on the fast path below, 6 when the fast path's test fails, 2 for 4-byte output), so two groups of one
// workgroup working on different frames would meet different barrier counts.  (An earlier form executed four
// unconditional barriers so that several groups -- k_cluster_merge_lna, since removed: four frames side by side --
// could call it together; whoever revives that has to make the count uniform again.)  tid / wave / lane are relative
// to the group, which is the workgroup.
// JSAFE: elements j < JSAFE (states tid + 256 j) are known to lie below S -- the launcher picks the smallest VPT that
// covers S, so S > 256 * (the next smaller VPT) -- and need no range test: the compiler otherwise keeps one execution
// mask per element (spilled to lanes of a vector register, read back with two v_readlane per use)
template <int VPT, int JSAFE = 0>
__device__ __forceinline__ void lna_row_from_registers(const float (&v)[VPT], int S, int tid, int wave, int lane,
                                                       double *red, int normalize, int lnabytes, int64_t f,
                                                       bool store, float *__restrict__ lp_out,
                                                       uint8_t *__restrict__ bytes_out) {
  double logz = 0.0;
  // Fast path (2-byte codes, no float output wanted): when the frame's best state is a normal float and within 51 nats
  // of 1.0 -- every real frame -- the float-denormal band of the reference's storage (ll in [ln 2^-150, ln 2^-126)) can
  // neither reach the maximum, nor move Z (its terms are below 2^-52 of the largest, the sum is formed from float
  // exponentials anyway), nor produce a code other than FF FF (lp < -87.3 + 51 < -36.008, and a value the reference
  // flushes to the floor gives FF FF as well).  So the quantisation detour -- ~25 vector instructions per value that
  // every wave executes as soon as ONE of its 64 states sits in the band, half of this VALU-bound kernel's work
  // (rocprofv3: VALU busy 78 % of the cycles) -- is skipped and the raw values are used.  Anything else (4-byte output,
  // lp_out, a frame whose best state is below e^-51) takes the exact path below.
  if (lnabytes == 2 && !lp_out) {   // uniform over the grid
    float mr = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPT; j++) mr = fmaxf(mr, v[j]);
    mr = wave_reduce_max(mr);
    if (lane == 0) red[wave] = (double)mr;
    __syncthreads();
    mr = (float)fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    if (mr >= -51.0f) {   // uniform over the group: the barriers below are passed by all of it
      double z = 0.0;
      if (normalize) {
#pragma unroll
        for (int j = 0; j < VPT; j++) z += (double)__builtin_amdgcn_exp2f((v[j] - mr) * 1.44269504088896340736f);
      }
      z = wave_reduce_sum(z);
      if (lane == 0) red[4 + wave] = z;
      __syncthreads();
      z = (red[4] + red[5]) + (red[6] + red[7]);
      __syncthreads();
      if (normalize) logz = (double)mr + log(z);
      if (!store) return;
#pragma unroll
      for (int j = 0; j < VPT; j++) {
        const int i = tid + 256 * j;
        if (j < JSAFE || i < S) {
          // band and flushed values: lp < -36.008 either way; the others exactly as the exact path
          const float lp = fmaxf((float)((double)v[j] - logz), (float)LOG_TINY_D);
          lna_store(lp, 2, f * (int64_t)S + i, nullptr, bytes_out);
        }
      }
      return;
    }
  }
  if (normalize) {  // uniform over the workgroup
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPT; j++) m = fmaxf(m, cast_f(v[j]));
    m = wave_reduce_max(m);
    if (lane == 0) red[wave] = (double)m;
    __syncthreads();
    m = (float)fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    double z = 0.0;
    if (m > -INFINITY) {
#pragma unroll
      for (int j = 0; j < VPT; j++)
        z += (double)__builtin_amdgcn_exp2f((cast_f(v[j]) - m) * 1.44269504088896340736f);
    }
    z = wave_reduce_sum(z);
    if (lane == 0) red[4 + wave] = z;
    __syncthreads();
    z = (red[4] + red[5]) + (red[6] + red[7]);
    __syncthreads();
    if (m > -INFINITY) logz = (double)m + log(z);  // all zero -> Z = 1 (phone_probs.cc:231-232)
  }
  if (!store) return;
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    const int i = tid + 256 * j;
    if (j < JSAFE || i < S) {
      const double vd = v[j] >= LN_FLT_MIN_F ? (double)v[j] : float_cast_loglik(v[j]);
      // safe_log's floor after the rounding to float instead of before it (rounding is monotone and
      // (float)log(1e-50) is the floor's own float, -inf - logz stays -inf): one f32 max for a double compare +
      // two selects -- the pass is bound by its vector instructions, not by HBM
      const float lp = fmaxf((float)(vd - logz), (float)LOG_TINY_D);
      lna_store(lp, lnabytes, f * (int64_t)S + i, lp_out, bytes_out);
    }
  }
}

}  // namespace aasr
