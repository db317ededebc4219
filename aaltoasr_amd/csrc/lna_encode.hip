// lna_encode.hip -- per-frame state normalisation + LNA packing on device.
//
// Replaces the tail of the phone_probs frame loop (aku/phone_probs.cc:224-262;
// PPToolbox: aku/PhoneProbsToolbox.cc:84-131):
//   obs[i] = (float) state_likelihood(i);  Z = sum_i (double) obs[i]
//   if (no-normalization || Z == 0) Z = 1
//   obs[i] = (float) safe_log(obs[i] / Z)            (floor log(1e-50))
//   2-byte: obs < -36.008 -> FF FF, else big-endian (int)(-1820*obs + .5)
//   4-byte: little-endian float
// The input here is the LOG state likelihood (float32, >= log(1e-50)), so the
// reference's float storage of the LINEAR likelihood is emulated:
//   ll <  ln(2^-150)            -> (float)lik == 0  -> output log(1e-50)
//   ln(2^-150) <= ll < ln(2^-126) -> denormal: q = rint(lik * 2^149) quanta
//   otherwise                   -> normal float, relative rounding 6e-8 (kept)
//
// One workgroup per frame; HBM-bound: S*(4 in + lnabytes out) bytes per frame.
// (A wave-per-frame variant without workgroup barriers was measured 2.5 ms slower
// on 449 280 x 3125: it reads the row three times with 4-byte loads.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "lna_device.h"

namespace aasr {

__global__ __launch_bounds__(256) void k_state_norm_lna(
    const float *__restrict__ loglik, int64_t F, int S, int64_t in_pitch, int normalize,
    int lnabytes, float *__restrict__ lp_out, uint8_t *__restrict__ bytes_out) {
  __shared__ double red[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int64_t f = blockIdx.x; f < F; f += gridDim.x) {
    const float *row = loglik + f * in_pitch;
    double logz = 0.0;
    if (normalize) {
      // pass 1: max of the float-cast log-likelihoods
      double m = -INFINITY;
      for (int i = tid; i < S; i += 256) {
        double v = float_cast_loglik(row[i]);
        m = v > m ? v : m;
      }
      m = wave_reduce_max(m);
      if (lane == 0) red[wave] = m;
      __syncthreads();
      m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
      __syncthreads();
      if (m > -INFINITY) {
        // pass 2: Z / exp(m) accumulated in double
        double z = 0.0;
        for (int i = tid; i < S; i += 256) {
          double v = float_cast_loglik(row[i]);
          z += (double)expf((float)(v - m));
        }
        z = wave_reduce_sum(z);
        if (lane == 0) red[4 + wave] = z;
        __syncthreads();
        z = (red[4] + red[5]) + (red[6] + red[7]);
        __syncthreads();
        logz = m + log(z);
      }  // all zero -> Z = 1 (phone_probs.cc:231-232)
    }
    // pass 3: normalise, cast, pack
    for (int i = tid; i < S; i += 256) {
      double v = float_cast_loglik(row[i]);
      double lpd = v - logz;
      if (!(lpd >= LOG_TINY_D)) lpd = LOG_TINY_D;  // safe_log floor (also -inf)
      float lp = (float)lpd;
      int64_t o = f * (int64_t)S + i;
      if (lp_out) lp_out[o] = lp;
      if (bytes_out) {
        if (lnabytes == 4) {
          ((float *)bytes_out)[o] = lp;
        } else {
          unsigned short code;
          if ((double)lp < -36.008) {
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
  }
}


// Register-resident variant for S <= 256*VPT: one global read of the row, the
// three passes (max, sum, pack) run on registers.
template <int VPT>
__global__ __launch_bounds__(256) void k_state_norm_lna_reg(
    const float *__restrict__ loglik, int64_t F, int S, int64_t in_pitch, int normalize, int lnabytes,
    float *__restrict__ lp_out, uint8_t *__restrict__ bytes_out, const int32_t *__restrict__ colmap) {
  __shared__ double red[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // the launcher takes the smallest instance that covers S: S > 256 * (the next smaller instance's VPT)
  constexpr int JSAFE = VPT == 4 ? 0 : VPT == 8 ? 4 : VPT == 10 ? 8 : VPT == 13 ? 10 : VPT == 16 ? 13 : 0;
  // colmap (engine-internal score layout of a routed model, gmm_engine_colmap): state i sits in column colmap[i] of a
  // score row; the same for every frame, so a thread looks its columns up once
  int col[VPT];
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    const int i = tid + 256 * j;
    col[j] = (colmap && (j < JSAFE || i < S)) ? colmap[i] : i;
  }
  // a workgroup walks frames blockIdx.x, + gridDim.x, ...; the next frame's row is requested
  // before this one is reduced (the four barriers of a frame leave nothing else to hide the
  // load latency behind)
  float vn[VPT];
  {
    const float *row = loglik + (int64_t)blockIdx.x * in_pitch;
#pragma unroll
    for (int j = 0; j < VPT; j++) {
      const int i = tid + 256 * j;
      vn[j] = ((j < JSAFE || i < S) && (int64_t)blockIdx.x < F) ? row[col[j]] : -INFINITY;
    }
  }
  for (int64_t f = blockIdx.x; f < F; f += gridDim.x) {
    float v[VPT];
#pragma unroll
    for (int j = 0; j < VPT; j++) v[j] = vn[j];
    if (f + gridDim.x < F) {
      const float *row = loglik + (f + gridDim.x) * in_pitch;
#pragma unroll
      for (int j = 0; j < VPT; j++) {
        const int i = tid + 256 * j;
        vn[j] = (j < JSAFE || i < S) ? row[col[j]] : -INFINITY;
      }
    }
    lna_row_from_registers<VPT, JSAFE>(v, S, tid, wave, lane, red, normalize, lnabytes, f, true, lp_out, bytes_out);
  }
}

// AASR_PREC_F64: the tail of the phone_probs frame loop as written (aku/phone_probs.cc:224-233), from the
// LINEAR state likelihoods in double: obs = (float) lik; Z = sum of (double) obs (Z == 0 or -N: 1);
// obs = (float) safe_log(obs / Z) with the division and the logarithm in double.  Against the
// reference only the order of the Z sum (a tree here, i = 0, 1, ... there) and the device's log()
// can differ, by ~1e-16 relative before the rounding to float.
__global__ __launch_bounds__(256) void k_state_norm_lna_f64(const double *__restrict__ lik, int64_t F, int S,
                                                            int normalize, int lnabytes, float *__restrict__ lp_out,
                                                            uint8_t *__restrict__ bytes_out) {
  __shared__ double red[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int64_t f = blockIdx.x; f < F; f += gridDim.x) {
    const double *row = lik + f * (int64_t)S;
    double z = 1.0;
    if (normalize) {
      double part = 0.0;
      for (int i = tid; i < S; i += 256) part += (double)(float)row[i];
      part = wave_reduce_sum(part);
      __syncthreads();  // red[] of the previous frame has been read
      if (lane == 0) red[wave] = part;
      __syncthreads();
      z = (red[0] + red[1]) + (red[2] + red[3]);
      if (z == 0) z = 1.0;
    }
    for (int i = tid; i < S; i += 256) {
      const float o = (float)row[i];
      const double q = (double)o / z;
      const float lp = (float)(q < 1e-50 ? LOG_TINY_D : log(q));
      lna_store(lp, lnabytes, f * (int64_t)S + i, lp_out, bytes_out);
    }
  }
}

void lna_encode_f64_launch(const double *d_lik, int64_t F, int S, int normalize, int lnabytes, float *d_lp,
                           uint8_t *d_bytes, hipStream_t stream) {
  if (F <= 0 || S <= 0) return;
  const int64_t blocks = std::min<int64_t>(F, 8192);
  hipLaunchKernelGGL(k_state_norm_lna_f64, dim3((unsigned)blocks), dim3(256), 0, stream, d_lik, F, S, normalize, lnabytes,
                     d_lp, d_bytes);
  AASR_HIP(hipGetLastError());
}

void lna_encode_launch(const float *d_loglik, int64_t F, int S, int normalize,
                       int lnabytes, float *d_lp, uint8_t *d_bytes, hipStream_t stream, int64_t in_pitch,
                       const int32_t *d_colmap) {
  if (in_pitch <= 0) in_pitch = S;  // row stride of the input in floats
  if (F <= 0 || S <= 0) return;
  if (d_colmap && S > 256 * 16) raise(AASR_ERR_UNSUPPORTED, "a column map needs the register-resident LNA kernels (S <= 4096)");
  int64_t blocks = F < (1 << 20) ? F : (1 << 20);
  // register-resident variants: a few workgroups per CU, each walking many frames with the next
  // row prefetched
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (S <= 256 * 16) blocks = std::min<int64_t>(blocks, (int64_t)cus * 32);  // measured: 2.56 ms at 4 per CU, 2.16 at 32, flat above
  if (S <= 256 * 4)
    hipLaunchKernelGGL(k_state_norm_lna_reg<4>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       d_loglik, F, S, in_pitch, normalize, lnabytes, d_lp, d_bytes, d_colmap);
  else if (S <= 256 * 8)
    hipLaunchKernelGGL(k_state_norm_lna_reg<8>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       d_loglik, F, S, in_pitch, normalize, lnabytes, d_lp, d_bytes, d_colmap);
  else if (S <= 256 * 10)
    hipLaunchKernelGGL(k_state_norm_lna_reg<10>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       d_loglik, F, S, in_pitch, normalize, lnabytes, d_lp, d_bytes, d_colmap);
  else if (S <= 256 * 13)
    hipLaunchKernelGGL(k_state_norm_lna_reg<13>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       d_loglik, F, S, in_pitch, normalize, lnabytes, d_lp, d_bytes, d_colmap);
  else if (S <= 256 * 16)
    hipLaunchKernelGGL(k_state_norm_lna_reg<16>, dim3((unsigned)blocks), dim3(256), 0, stream,
                       d_loglik, F, S, in_pitch, normalize, lnabytes, d_lp, d_bytes, d_colmap);
  else
    hipLaunchKernelGGL(k_state_norm_lna, dim3((unsigned)blocks), dim3(256), 0, stream, d_loglik,
                       F, S, in_pitch, normalize, lnabytes, d_lp, d_bytes);
  AASR_HIP(hipGetLastError());
}

}  // namespace aasr
