// gmm_cluster.hip -- Gaussian clustering for the scoring path.
//
// Replaces, for a block of frames,
//   PDFPool::read_clustering              aku/Distributions.cc:3114-3170
//     Gaussian::merge (unit weights)      aku/Distributions.cc:853-898
//   HmmSet::set_clustering_min_evals      aku/HmmSet.cc:1359-1366
//   PDFPool::precompute_likelihoods, cluster branch   aku/Distributions.cc:2684-2722
//   PDFPool::compute_likelihood (cache rule)           aku/Distributions.cc:2636-2644
//
// What the reference does per frame: evaluate every cluster centre, pop the
// centres best first and evaluate their member Gaussians exactly until both
// `min_clusters` clusters and `min_gaussians` Gaussians are done; every other
// Gaussian gets its centre's likelihood.  A cached value <= 0 is not trusted by
// PDFPool::compute_likelihood, so a Gaussian whose centre underflowed to 0 (or
// that is in no cluster) is still evaluated exactly when a mixture reads it.
//
// The mixture sum is linear, so a state's likelihood splits into
//     sum over components whose exact value is used            (matrix kernel)
//   + sum over clusters c of W[s][c] * centre likelihood(c)    (sparse, per state)
// with W[s][c] the total weight of the state's components in cluster c.  Per
// block of frames that is five launches:
//   k_cluster_centres  f64 centre log-likelihoods, same operation order as
//                      DiagonalGaussian::compute_log_likelihood (:1040-1062), so
//                      the ranking of the centres is the reference's ranking;
//   k_cluster_select   one wave per 64 frames: per frame the threshold T with
//                      "centre >= T  <=>  popped by the reference's first loop",
//                      found by a multi-level 64-bucket histogram selection
//                      (exact: buckets are a monotone function of the value);
//                      writes one bit per (cluster, frame) -- 1 = exact -- and
//                      the linear centre values 2^(log2e*ll + ref) of the rest;
//   k_cluster_expand   the bits per packed row of the track layout in use;
//   track kernel <CL>  the regular scoring kernel (f32 or bf16x3) with the bits
//                      applied to its accumulators (one v_cndmask per register,
//                      masks fetched with 64-byte scalar loads), no floor;
//   k_cluster_merge    out = log(max(exact part + sum_c W[s][c] centre_c, 1e-50)).
// Gaussians the expanded form cannot hold (outlier routing / ill-conditioned models, gmm.h) take their
// exact values from the centred kernel under the same bits (gmm_outliers_masked_launch /
// gmm_centred_masked_launch); centres without a common f32 exponent switch the pass to log2 centre
// values and k_cluster_merge_log.
// The matrix work is not reduced (on this machine evaluating every Gaussian is
// cheaper than gathering per-frame cluster subsets); the point of this path is
// output parity with recognisers configured with -C/--eval-ming.
// Ties: the reference ranks the centres by their LINEAR likelihood exp(ll) in a
// std::priority_queue, so centres that underflowed to 0.0 all tie (ll below
// ln 2^-1075) and so do empty clusters (an "invalid" centre: ll = 0, likelihood
// 1).  When the loop's stopping point falls inside such a group, which members
// are popped -- and how many clusters the loop counts -- depends on libstdc++'s
// heap order.  Denormal likelihoods (ll between ln 2^-1075 and ln 2^-1022) are
// small multiples of 2^-1074, so different centres tie there too (lin_key).  k_cluster_select detects that case per frame (the histogram
// selection ends on a group of equal keys of which only a part is needed) and
// hands the frame to k_cluster_select_heap, which replays push_heap / pop_heap
// for that frame exactly (one thread per frame).  Not modelled: two different
// ll whose exp() rounds to the same double (|ll| < 1 and a gap of one ulp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <type_traits>
#include <sstream>

#include "gmm.h"

namespace aasr {

static const double kLog2eD = 1.4426950408889634074;
// exp(x) rounds to 0.0 in double below ln(2^-1075)
#define AASR_EXP_UNDERFLOW_LL (-745.13321910194122)
#define AASR_LOG_TINY_F (-115.12925464970228f)

// Ranking key of a centre.  The reference ranks by the double exp(ll) (aku/Distributions.cc
// :2688-2691).  Where exp(ll) is a normal double, ll itself is used (exp is strictly increasing
// there to within its last-bit rounding).  Below ln 2^-1022 the result is a denormal -- a small
// integer multiple q of 2^-1074 -- or 0: different ll share one likelihood, so the key is built
// from q: -2000 + q 2^-42 (below every normal key; equal likelihoods give equal keys; large q
// may merge neighbours, which only sends a frame to the exact replay).  Key -2000 <=> exp(ll) == 0.
#define AASR_KEY_ZERO (-2000.0)
__device__ __forceinline__ double lin_key(double ll) {
  if (ll >= -708.39641853226408) return ll;
  if (ll < AASR_EXP_UNDERFLOW_LL - 1.0) return AASR_KEY_ZERO;
  // q = exp(ll) / 2^-1074 rounded to an integer, without relying on denormal results of the
  // device's exp(): exp(ll + 1074 ln 2), the sum carried as s + e (TwoSum) so that the argument
  // is exact to ~1e-30 and q is as accurate as exp() itself (1 ulp: it differs from the
  // reference's rounding only when the true value lies within ~1e-16 q of a half integer)
  const double H = 744.4400719213812, Hlo = 4.422444340918698e-14;
  const double sum = ll + H;
  const double bb = sum - ll;
  const double e = (ll - (sum - bb)) + (H - bb);
  double q = exp(sum);
  q += q * (e + Hlo);
  return AASR_KEY_ZERO + rint(q) * 0x1p-42;  // one ulp of 2000 per quantum: exact for q < 2^41
}

// The same key as a float, for the selection's fast path (k_cluster_select<KPL, float>): rounding is monotone, so
// sorting by the float keys orders the centres as the doubles do except inside groups the rounding made equal --
// and a group of equal keys at the stopping point is what sends a frame to the replay anyway, which then ranks by
// the doubles.  Normal range: (float)ll == (float)lin_key(ll).  Denormal band: -1900 + log2(q) (q = the likelihood
// in units of 2^-1074; below every normal key, above the zero key, monotone in q); zero likelihood: the zero key.
__device__ __forceinline__ float lin_key32(double ll) {
  if (ll >= -708.39641853226408) return (float)ll;
  const double k = lin_key(ll);
  if (!(k > AASR_KEY_ZERO)) return (float)AASR_KEY_ZERO;
  const double q = (k - AASR_KEY_ZERO) * 0x1p42;   // exact: the integer q
  return -1900.0f + __log2f((float)q);
}

// diagnostic (aasr_debug_cluster_pass_bytes): the scratch budget of a pass, so that a test can force several passes
static double g_pass_bytes = 0;
static int g_last_passes = 0;   // passes of the most recent clustered run (aasr_debug_cluster_last_passes)
// diagnostic switch (aasr_debug_cluster_heap): every frame takes the queue replay
static int g_force_heap = AASR_EXPERIMENT_ENV("AASR_CLUSTER_HEAP") ? atoi(AASR_EXPERIMENT_ENV("AASR_CLUSTER_HEAP")) : 0;

// ---------------------------------------------------------------- centres --
// thread = frame (float values parked in LDS, [dimension][thread]); the records
// of 8 clusters at a time are staged in LDS (double buffered) and read back as
// 16-byte broadcasts, so every thread runs 8 independent accumulation chains.
// The dimension loop is a real loop: fully unrolled, the compiler hoists every
// operand load of a group and spills.
constexpr int kCentreThreads = 256;

template <typename XT>  // float frames, or double (AASR_PREC_F64)
__global__ __launch_bounds__(kCentreThreads) void k_cluster_centres(
    const XT *__restrict__ frames, int64_t F, int dim, int dimp,
    const double *__restrict__ rec, const double *__restrict__ cconst, int groups,
    int groups_per_y, double *__restrict__ ll64, int64_t Cs) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  double *pbuf = (double *)smem_c;                              // [2][dimp][8][2]
  XT *xs = (XT *)(smem_c + (size_t)2 * dimp * 16 * 8);          // [dimp][kCentreThreads]
  const int tid = threadIdx.x;
  const int64_t f = (int64_t)blockIdx.x * kCentreThreads + tid;
  const int64_t fc = f < F ? f : F - 1;
  for (int d = 0; d < dimp; d++) xs[d * kCentreThreads + tid] = d < dim ? frames[fc * dim + d] : (XT)0;
  const int g_begin = blockIdx.y * groups_per_y;
  const int g_end = min(groups, g_begin + groups_per_y);
  const int rec_doubles = dimp * 16;
  for (int i = tid; i < rec_doubles; i += kCentreThreads)
    pbuf[i] = rec[(size_t)g_begin * rec_doubles + i];
  __syncthreads();
  for (int cg = g_begin; cg < g_end; cg++) {
    const double *cur = pbuf + ((cg - g_begin) & 1) * rec_doubles;
    double *nxt = pbuf + ((cg - g_begin + 1) & 1) * rec_doubles;
    if (cg + 1 < g_end)
      for (int i = tid; i < rec_doubles; i += kCentreThreads)
        nxt[i] = rec[(size_t)(cg + 1) * rec_doubles + i];
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.0;
#pragma unroll 2
    for (int d = 0; d < dimp; d++) {
      const double x = (double)xs[d * kCentreThreads + tid];
      const f64x2 *r = (const f64x2 *)(cur + d * 16);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const f64x2 mp = r[j];  // (mean, precision): one broadcast read
        const double dd = x - mp.x;
        acc[j] += dd * dd * mp.y;  // (d*d)*prec, then the sum: reference order
      }
    }
    if (f < F) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        double ll = acc[j] * -0.5;
        ll += cconst[cg * 8 + j];
        // stored as the ranking key (== ll wherever exp(ll) is a normal double); the selection
        // kernels never see the rare denormal branch and its exp()
        ll64[f * Cs + (int64_t)cg * 8 + j] = lin_key(ll);
      }
    }
    __syncthreads();
  }
}

// The same centre log-likelihoods for FLOAT frames in the expanded form, ll = c' + sum_d x (a + b x) with
// a = p mu, b = -p/2, c' = c - 1/2 sum p mu^2 (all formed in double on the host): two f64 FMAs per frame,
// cluster and dimension instead of the reference order's four operations, and the records of dimension d + 1
// requested while dimension d is computed (the reference-order kernel waits out an LDS round trip per
// dimension: it sat at 42 % of its f64 rate).  Against the reference order the key moves by ~1e-13 relative
// (cancellation at f64 precision), which can re-order two centres only when their likelihoods agree to twelve
// digits; structured ties -- duplicate, empty and underflowing centres -- come out identical as before, because
// identical parameters give identical arithmetic.  AASR_PREC_F64 keeps the reference-order kernel.
__global__ __launch_bounds__(kCentreThreads) void k_cluster_centres_fma(
    const float *__restrict__ frames, int64_t F, int dim, int dimp, const double *__restrict__ rec,
    const double *__restrict__ cconst, int groups, int groups_per_y, double *__restrict__ ll64, int64_t Cs) {
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  double *pbuf = (double *)smem_c;                                 // [2][dimp][8][2]
  float *xs = (float *)(smem_c + (size_t)2 * dimp * 16 * 8);       // [dimp][nt]
  const int tid = threadIdx.x;
  const int nt = blockDim.x;   // 256, or 64 for models of more than 64 dimensions (LDS: dimp x nt floats)
  const int64_t f = (int64_t)blockIdx.x * nt + tid;
  const int64_t fc = f < F ? f : F - 1;
  for (int d = 0; d < dimp; d++) xs[d * nt + tid] = d < dim ? frames[fc * dim + d] : 0.0f;
  const int g_begin = blockIdx.y * groups_per_y;
  const int g_end = min(groups, g_begin + groups_per_y);
  const int rec_doubles = dimp * 16;
  for (int i = tid; i < rec_doubles; i += nt) pbuf[i] = rec[(size_t)g_begin * rec_doubles + i];
  __syncthreads();
  for (int cg = g_begin; cg < g_end; cg++) {
    const double *cur = pbuf + ((cg - g_begin) & 1) * rec_doubles;
    double *nxt = pbuf + ((cg - g_begin + 1) & 1) * rec_doubles;
    if (cg + 1 < g_end)
      for (int i = tid; i < rec_doubles; i += nt) nxt[i] = rec[(size_t)(cg + 1) * rec_doubles + i];
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.0;
    f64x2 r0[8], r1[8];
    float x0 = xs[tid], x1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) r0[j] = ((const f64x2 *)cur)[j];
#pragma unroll 1
    for (int d = 0; d < dimp; d += 2) {   // dimp is a multiple of 8
      x1 = xs[(d + 1) * nt + tid];
#pragma unroll
      for (int j = 0; j < 8; j++) r1[j] = ((const f64x2 *)(cur + (d + 1) * 16))[j];
      __builtin_amdgcn_sched_barrier(0);
      {
        const double x = (double)x0;
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = __builtin_fma(x, __builtin_fma(r0[j].y, x, r0[j].x), acc[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (d + 2 < dimp) {
        x0 = xs[(d + 2) * nt + tid];
#pragma unroll
        for (int j = 0; j < 8; j++) r0[j] = ((const f64x2 *)(cur + (d + 2) * 16))[j];
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const double x = (double)x1;
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = __builtin_fma(x, __builtin_fma(r1[j].y, x, r1[j].x), acc[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (f < F) {
#pragma unroll
      for (int j = 0; j < 8; j++) ll64[f * Cs + (int64_t)cg * 8 + j] = lin_key(acc[j] + cconst[cg * 8 + j]);
    }
    __syncthreads();
  }
}

// The expanded form is a dense contraction over K = 2 dim + 1, so it also runs on the f64 MATRIX pipe
// (v_mfma_f64_16x16x4_f64: the same 78.6 TFLOP/s as the f64 vector rate on this part, but an operand is fetched
// once per 16 x 16 block instead of once per frame, cluster and dimension -- the vector kernels above are bound by
// their LDS broadcast reads, 16 bytes per pair and dimension on a pipe the whole CU shares).  A wave keeps the
// K x 32 operand of its 32 frames in registers (x, x^2 exact in double, 1), the coefficient tiles of 16 clusters
// stream through LDS ([k step][lane] doubles, host-packed: one MFMA operand is 512 contiguous bytes), 8 waves =
// 256 frames per workgroup.  Same keys as k_cluster_centres_fma to ~1e-13 (another summation order).
constexpr int kMfmaCentreWaves = 8;
typedef double f64x4 __attribute__((ext_vector_type(4)));
// Where the keys go: as doubles [frame][Cs] (key32 null, list null); as floats [frame][Cs] (key32 set: the selection's
// fast path); or -- list mode -- the doubles of the frames tie_list[0 .. count) only, row i for list entry i (the frames
// the float selection could not settle: k_cluster_select_heap ranks them by the exact keys; a workgroup walks the list
// in strides, the count is read on the device).  Same arithmetic in all three.
template <int KS>   // k steps of 4: 4 KS >= 2 dim + 1
__global__ __launch_bounds__(64 * kMfmaCentreWaves) void k_cluster_centres_mfma(
    const float *__restrict__ frames, int64_t F, int dim, const double *__restrict__ bpack, int tiles,
    int tiles_per_y, double *__restrict__ ll64, int64_t Cs, float *__restrict__ key32,
    const int32_t *__restrict__ list, int64_t list_cap) {
  __shared__ __attribute__((aligned(16))) double btile[2][KS * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int i16 = lane & 15, kq = lane >> 4;
  if (list) F = min((int64_t)list[list_cap], list_cap);
  for (int64_t chunk = blockIdx.x; chunk * (32 * kMfmaCentreWaves) < F; chunk += gridDim.x) {
  const int64_t f0 = chunk * (32 * kMfmaCentreWaves) + wave * 32;
  double a[2][KS];
#pragma unroll
  for (int nb = 0; nb < 2; nb++) {
    int64_t f = f0 + nb * 16 + i16;
    if (f > F - 1) f = F - 1;
    const float *xr = frames + (list ? (int64_t)list[f] : f) * dim;
#pragma unroll
    for (int q = 0; q < KS; q++) {
      const int K = 4 * q + kq;
      double v = 0.0;
      if (K < dim) v = (double)xr[K];
      else if (K < 2 * dim) {
        const double x = (double)xr[K - dim];
        v = x * x;
      } else if (K == 2 * dim) v = 1.0;
      a[nb][q] = v;
    }
  }
  const int t_begin = blockIdx.y * tiles_per_y;
  const int t_end = min(tiles, t_begin + tiles_per_y);
  constexpr int kTileDoubles = KS * 64;
  for (int i = tid; i < kTileDoubles; i += 64 * kMfmaCentreWaves) btile[0][i] = bpack[(size_t)t_begin * kTileDoubles + i];
  __syncthreads();
  for (int t = t_begin; t < t_end; t++) {
    const double *cur = btile[(t - t_begin) & 1];
    double *nxt = btile[(t - t_begin + 1) & 1];
    // the next tile: requested here, parked in registers under the matrix instructions, written to LDS behind them
    // (load and LDS store side by side, the wait for the load sat in FRONT of the matrix loop: a global-memory
    // latency per tile, the matrix pipe 57 % busy)
    constexpr int kPre = (kTileDoubles + 64 * kMfmaCentreWaves - 1) / (64 * kMfmaCentreWaves);
    double pre[kPre];
    if (t + 1 < t_end) {
#pragma unroll
      for (int u = 0; u < kPre; u++) {
        const int i = tid + u * 64 * kMfmaCentreWaves;
        pre[u] = i < kTileDoubles ? bpack[(size_t)(t + 1) * kTileDoubles + i] : 0.0;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    f64x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < KS; q++) {
      const double b = cur[q * 64 + lane];
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][q], b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][q], b, c1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < t_end) {
#pragma unroll
      for (int u = 0; u < kPre; u++) {
        const int i = tid + u * 64 * kMfmaCentreWaves;
        if (i < kTileDoubles) nxt[i] = pre[u];
      }
    }
    // the buffer swap's barrier sits HERE, not behind the stores of the keys: every wave is through its matrix loop
    // (nobody reads `cur` any more, `nxt` is complete), and the stores drain under the next tile's instructions
    __syncthreads();
    // D[i = 4 r + lane / 16][j = lane % 16] (register r of the f64 16x16 result): frame f0 + 16 nb + 4 r + kq,
    // cluster 16 t + i16
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int64_t fa = f0 + 4 * r + kq, fb = fa + 16;
      const bool col = (int64_t)t * 16 + i16 < Cs;   // Cs is a multiple of 8: the last tile may be half a tile
      if (key32) {
        if (col && fa < F) key32[fa * Cs + (int64_t)t * 16 + i16] = lin_key32(c0[r]);
        if (col && fb < F) key32[fb * Cs + (int64_t)t * 16 + i16] = lin_key32(c1[r]);
      } else {
        if (col && fa < F) ll64[fa * Cs + (int64_t)t * 16 + i16] = lin_key(c0[r]);
        if (col && fb < F) ll64[fb * Cs + (int64_t)t * 16 + i16] = lin_key(c1[r]);
      }
    }
  }
  __syncthreads();   // the next chunk refills btile[0]
  }   // chunk
}

// ----------------------------------------------------------------- select --
// Wave-wide reductions and the bucket scan on the DPP data path (row shifts / mirrors / broadcasts
// inside the VALU) instead of ds_bpermute: a __shfl_xor reduction of a double is 12 dependent LDS-crossbar
// round trips, and the selection loop is a chain of three such reductions per level and frame.
//   0xB1 / 0x4E quad_perm [1,0,3,2] / [2,3,0,1], 0x141 row_half_mirror, 0x140 row_mirror: after the four
//   steps every lane holds its 16-lane row's result; 0x142 row_bcast:15 (rows 1, 3) and 0x143 row_bcast:31
//   (rows 2, 3) carry it across rows, so lane 63 ends with the wave's; v_readlane broadcasts it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  const unsigned long long o = __builtin_bit_cast(unsigned long long, old), x = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)x, CTRL, ROW_MASK, 0xF, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(x >> 32), CTRL, ROW_MASK, 0xF, false);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ double readlane63_f64(double v) {
  const unsigned long long x = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 63);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
__device__ __forceinline__ double wave_min_f64(double v) {
  v = fmin(v, dpp_f64<0xB1, 0xF>(v, v));
  v = fmin(v, dpp_f64<0x4E, 0xF>(v, v));
  v = fmin(v, dpp_f64<0x141, 0xF>(v, v));
  v = fmin(v, dpp_f64<0x140, 0xF>(v, v));
  v = fmin(v, dpp_f64<0x142, 0xA>(v, v));
  v = fmin(v, dpp_f64<0x143, 0xC>(v, v));
  return readlane63_f64(v);
}
__device__ __forceinline__ double wave_max_f64(double v) {
  v = fmax(v, dpp_f64<0xB1, 0xF>(v, v));
  v = fmax(v, dpp_f64<0x4E, 0xF>(v, v));
  v = fmax(v, dpp_f64<0x141, 0xF>(v, v));
  v = fmax(v, dpp_f64<0x140, 0xF>(v, v));
  v = fmax(v, dpp_f64<0x142, 0xA>(v, v));
  v = fmax(v, dpp_f64<0x143, 0xC>(v, v));
  return readlane63_f64(v);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_min_f32(float v) {
  v = fminf(v, dpp_f32<0xB1, 0xF>(v, v));
  v = fminf(v, dpp_f32<0x4E, 0xF>(v, v));
  v = fminf(v, dpp_f32<0x141, 0xF>(v, v));
  v = fminf(v, dpp_f32<0x140, 0xF>(v, v));
  v = fminf(v, dpp_f32<0x142, 0xA>(v, v));
  v = fminf(v, dpp_f32<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max_f32(float v) {
  v = fmaxf(v, dpp_f32<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_f32<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_f32<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_f32<0x140, 0xF>(v, v));
  v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_min_key(double v) { return wave_min_f64(v); }
__device__ __forceinline__ double wave_max_key(double v) { return wave_max_f64(v); }
__device__ __forceinline__ float wave_min_key(float v) { return wave_min_f32(v); }
__device__ __forceinline__ float wave_max_key(float v) { return wave_max_f32(v); }
// wave-wide sum of an int (result in every lane)
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
  return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum over the lanes of an unsigned (Hillis-Steele inside the 16-lane rows with zero
// fill: 0x111 / 0x112 / 0x114 / 0x118 = row_shr 1 / 2 / 4 / 8, then the row totals across rows)
__device__ __forceinline__ unsigned wave_scan_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}

// 64 x 64 bit transpose over the lanes of a wave: lane r enters with row r (lo = columns 0-31, hi = 32-63) and
// leaves with column r (lo = rows 0-31, hi = rows 32-63).  Block swaps of 32, 16, ... 1 between lane pairs.
__device__ __forceinline__ void wave_bit_transpose64(unsigned &lo, unsigned &hi, int lane) {
  {
    const bool lower = lane & 32;
    const unsigned recv = (unsigned)__shfl_xor((int)(lower ? lo : hi), 32);
    if (lower) lo = recv;
    else hi = recv;
  }
#pragma unroll
  for (int step = 0; step < 5; step++) {
    const int j = 16 >> step;
    const unsigned m = j == 16 ? 0x0000FFFFu : j == 8 ? 0x00FF00FFu : j == 4 ? 0x0F0F0F0Fu : j == 2 ? 0x33333333u : 0x55555555u;
    const bool lower = lane & j;
    const unsigned keep = lower ? ~m : m;
    // partner lane ^ j: quad permutes for 1 and 2, a rotation by 8 inside the 16-lane row for 8 (DPP, no LDS
    // crossbar round trip); 4 and 16 have no DPP form on gfx9 and go through ds_bpermute
    unsigned plo, phi;
    if (j == 1) {
      plo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xF, 0xF, false);
      phi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xF, 0xF, false);
    } else if (j == 2) {
      plo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xF, 0xF, false);
      phi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xF, 0xF, false);
    } else if (j == 8) {
      plo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xF, 0xF, false);
      phi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xF, 0xF, false);
    } else {
      plo = (unsigned)__shfl_xor((int)lo, j);
      phi = (unsigned)__shfl_xor((int)hi, j);
    }
    plo &= keep;
    phi &= keep;
    lo = (lo & keep) | (lower ? plo >> j : plo << j);
    hi = (hi & keep) | (lower ? phi >> j : phi << j);
  }
}

// One wave per 64 consecutive frames (= one word of the selection masks);
// lane l holds clusters l, 64 + l, ...  (KPL per lane).
// four waves per SIMD (<= 128 VGPRs): a 250 k-frame sub-pass is 3906 single-wave workgroups, which
// then fit the chip's 4096 wave slots in one round (with three per SIMD: 1.27 rounds)
// The stopping point of one frame: the threshold T such that the clusters with key >= T are the ones the reference's
// priority queue pops until min_clusters and min_gaussians are both met (aku/Distributions.cc:2684-2722) -- a radix
// selection over the wave: 64 buckets between the candidates' extremes, cluster counts and Gaussian counts per bucket
// in one 64-bit LDS histogram, the bucket that holds the stopping point becomes the next level's range.  tie: the
// stopping point lies inside a group of equal keys that is not taken as a whole (the queue's order among equals
// decides).  v: lane l holds the keys of clusters l, 64 + l, ...; cand: the valid ones; wave-uniform control flow.
template <int KPL, typename KT>
__device__ __forceinline__ void select_threshold(const KT (&v)[KPL], unsigned long long cand, const int (&sz)[KPL], int lane,
                                                 unsigned long long *hist, int need_c, int need_g, KT &T, bool &tie) {
  for (int level = 0; level < 64; level++) {
    KT lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int j = 0; j < KPL; j++)
      if ((cand >> j) & 1) {
        lo = v[j] < lo ? v[j] : lo;
        hi = v[j] > hi ? v[j] : hi;
      }
    lo = wave_min_key(lo);
    hi = wave_max_key(hi);
    if (!(hi >= lo)) {  // no candidate left: the request exceeds what is there
      T = -INFINITY;
      break;
    }
    const KT scale = (KT)64 / (hi - lo);
    if (!(hi > lo) || !(scale < (sizeof(KT) == 8 ? (KT)1.0e300 : (KT)1.0e37f))) {  // one value left (single centre or ties)
      int n_tied = 0;
#pragma unroll
      for (int j = 0; j < KPL; j++) n_tied += (int)((cand >> j) & 1);
      n_tied = wave_sum_i32(n_tied);
      // every pop takes one cluster: with need_c >= n_tied the whole group goes whatever the
      // order; otherwise the queue's order among equals decides -> replay
      if (n_tied == 1 || need_c >= n_tied) T = lo;
      else tie = true;
      break;
    }
    hist[lane] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < KPL; j++)
      if ((cand >> j) & 1) {
        int b = (int)((hi - v[j]) * scale);
        b = b > 63 ? 63 : b;
        atomicAdd(&hist[b], (1ull << 32) | (unsigned long long)(unsigned)sz[j]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long h = hist[lane];
    // inclusive prefix over buckets 0..lane; counts and sizes are separate 32-bit fields (no carry between them)
    const unsigned long long cum =
        ((unsigned long long)wave_scan_u32((unsigned)(h >> 32)) << 32) | wave_scan_u32((unsigned)h);
    const long long cumc = (long long)(cum >> 32), cumg = (long long)(cum & 0xffffffffull);
    const bool sat = (need_c - cumc <= 0) && (need_g - cumg <= 0);
    const unsigned long long ballot = __ballot(sat);
    if (ballot == 0ull) {
      T = -INFINITY;
      break;
    }
    const int B = __ffsll((long long)ballot) - 1;
    const unsigned long long excl = __shfl(cum - h, B);
    need_c -= (int)(excl >> 32);
    need_g -= (int)(excl & 0xffffffffull);
    // the boundary lies in bucket B: better buckets are taken, worse ones are not
#pragma unroll
    for (int j = 0; j < KPL; j++)
      if ((cand >> j) & 1) {
        int b = (int)((hi - v[j]) * scale);
        b = b > 63 ? 63 : b;
        if (b != B) cand &= ~(1ull << j);
      }
    __builtin_amdgcn_wave_barrier();
  }
}

// what the merge reads for (frame, cluster): 0 / -inf where the members are evaluated exactly, else the centre's
// likelihood on the track kernels' reference exponent (or as log2: ClusterState::log_merge)
__device__ __forceinline__ float centre_value(float key, bool use_exact, bool log_vals, double ref) {
  if (log_vals) return use_exact ? -INFINITY : (float)((double)key * kLog2eD);
  return use_exact ? 0.0f : exp2f((float)((double)key * kLog2eD + ref));
}

// KT: the ranking keys as doubles (k_cluster_centres*), or as floats (lin_key32; half the bytes, half the registers,
// f32 reductions and bucket arithmetic) -- then a frame whose stopping point falls into a group of equal keys goes to
// the replay even where the doubles would have settled it, and the replay ranks by the doubles.
template <int KPL, typename KT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_cluster_select(
    const KT *__restrict__ ll64, int64_t F, int C, int64_t Cs,
    const int32_t *__restrict__ csize, int min_clusters, int min_gaussians, double ref,
    unsigned long long *__restrict__ maskw, float *__restrict__ cval,
    int32_t *__restrict__ n_exact, int32_t *__restrict__ tie_list, int64_t tie_cap,
    int force_heap) {
  __shared__ unsigned long long hist[64];
  __shared__ unsigned long long bits[64][KPL];  // [frame in word][cluster slot] ballots
  const bool log_vals = ref != ref;  // NaN: centre values as log2 (-inf where exact), for k_cluster_merge_log
  const int lane = threadIdx.x;
  int sz[KPL];
#pragma unroll
  for (int j = 0; j < KPL; j++) {
    const int c = j * 64 + lane;
    sz[j] = c < C ? csize[c] : 0;
  }
  const int64_t word = blockIdx.x;
  // the next frame's keys are requested while this frame is selected (a frame's loop is one long
  // dependent chain: nothing else hides the load)
  constexpr bool kPrefetch = KPL * sizeof(KT) <= 16 * sizeof(double);  // 2 x KPL keys: beyond that the registers are not there
  KT vn[kPrefetch ? KPL : 1];
  if (kPrefetch) {
#pragma unroll
    for (int j = 0; j < KPL; j++) {
      const int c = j * 64 + lane;
      vn[j] = (c < C && word * 64 < F) ? ll64[word * 64 * Cs + c] : (KT)0;
    }
  }
  for (int fi = 0; fi < 64; fi++) {
    const int64_t f = word * 64 + fi;
    if (f >= F) {  // wave-uniform
#pragma unroll
      for (int j = 0; j < KPL; j++)
        if (lane == 0) bits[fi][j] = 0ull;
      continue;
    }
    KT v[KPL];
    unsigned long long cand = 0;
#pragma unroll
    for (int j = 0; j < KPL; j++) {
      const int c = j * 64 + lane;
      // ranking key (k_cluster_centres): the reference compares exp(ll)
      const KT x = kPrefetch ? vn[kPrefetch ? j : 0] : (c < C ? ll64[f * Cs + c] : (KT)0);
      v[j] = x;
      if (c < C && x == x) cand |= 1ull << j;
    }
    if (kPrefetch && fi + 1 < 64 && f + 1 < F) {
      const KT *row = ll64 + (f + 1) * Cs;
#pragma unroll
      for (int j = 0; j < KPL; j++) {
        const int c = j * 64 + lane;
        vn[kPrefetch ? j : 0] = c < C ? row[c] : (KT)0;
      }
    }
    int need_c = min_clusters, need_g = min_gaussians;
    KT T = INFINITY;  // key >= T  <=>  members evaluated exactly
    bool tie = false;     // wave-uniform: the boundary lies inside a group of equal keys
    if ((need_c > 0 || need_g > 0) && !force_heap) select_threshold<KPL, KT>(v, cand, sz, lane, hist, need_c, need_g, T, tie);
    if (force_heap) tie = min_clusters > 0 || min_gaussians > 0;
    if (tie && lane == 0) {
      // slot 0 counts; a list that is full (cannot happen: one entry per frame) drops nothing
      const int at = atomicAdd(&tie_list[tie_cap], 1);
      if (at < tie_cap) tie_list[at] = (int32_t)f;
    }
    int exact = 0;
#pragma unroll
    for (int j = 0; j < KPL; j++) {
      const int c = j * 64 + lane;
      const bool valid = c < C;
      const bool sel = valid && v[j] >= T;
      exact += sel ? 1 : 0;
      // a centre whose likelihood is 0.0 in double is not trusted by
      // PDFPool::compute_likelihood: its members are evaluated exactly
      const bool use_exact = sel || !(v[j] > AASR_KEY_ZERO);
      const unsigned long long bal = __ballot(valid && use_exact);
      if (lane == 0) bits[fi][j] = bal;
      // a key below the normal range stands for a likelihood under 2^-1022: 0.0f either way.  The value is a
      // function of the key AS A FLOAT in both instances, so that the replay -- which holds the doubles of the same
      // arithmetic -- writes the same bits as the float selection
      if (valid) cval[f * C + c] = centre_value((float)v[j], use_exact, log_vals, ref);
    }
    if (n_exact) {
      exact = wave_sum_i32(exact);
      if (lane == 0) n_exact[f] = exact;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // transpose 64x64 bit blocks: lane l assembles the word of cluster 64 j + l
#pragma unroll
  for (int j = 0; j < KPL; j++) {
    const int c = j * 64 + lane;
    // lane fi holds the ballot of frame fi (bit l = cluster 64 j + l); transposed, lane l holds its cluster's frames
    const unsigned long long bw = bits[lane][j];
    unsigned tlo = (unsigned)bw, thi = (unsigned)(bw >> 32);
    wave_bit_transpose64(tlo, thi, lane);
    const unsigned long long w = (unsigned long long)tlo | ((unsigned long long)thi << 32);
    if (c < C) maskw[word * (C + 1) + c] = w;
  }
  if (lane == 0) maskw[word * (C + 1) + C] = ~0ull;  // rows in no cluster: always exact
}

// Frames the FLOAT selection left open -- a group of equal float keys at the stopping point, which for two centres
// within a float ulp of each other is an artefact of the rounding (~2e-4 of the frames of configs[1]; sending them
// to the replay would cost its launch-long 3 ms as soon as one frame is listed).  The selection again on doubles, one
// wave per listed frame: the keys are computed here (the expanded form of k_cluster_centres_fma), the outputs of the
// frame rewritten (its mask bits with atomics, as the replay does; the centre values from the float keys, so that they
// are the same function of the same number on every path).  Groups equal in double as well -- duplicate, empty,
// underflowing centres -- go on to tie_list and the replay.
template <int KPL>
__global__ __launch_bounds__(64) void k_cluster_select_pending(
    const float *__restrict__ frames, int dim, int dimp, const double *__restrict__ rec_fma,
    const double *__restrict__ cconst_fma, const float *__restrict__ key32, int C, int64_t Cs,
    const int32_t *__restrict__ csize, int min_clusters, int min_gaussians, double ref,
    unsigned long long *__restrict__ maskw, float *__restrict__ cval, int32_t *__restrict__ n_exact,
    const int32_t *__restrict__ pend_list, int64_t pend_cap, int32_t *__restrict__ tie_list, int64_t tie_cap,
    int force_heap) {
  __shared__ unsigned long long hist[64];
  const bool log_vals = ref != ref;
  const int lane = threadIdx.x;
  const int count = (int)min((int64_t)pend_list[pend_cap], pend_cap);
  if ((int)blockIdx.x >= count) return;
  int sz[KPL];
#pragma unroll
  for (int j = 0; j < KPL; j++) {
    const int c = j * 64 + lane;
    sz[j] = c < C ? csize[c] : 0;
  }
  for (int i = blockIdx.x; i < count; i += gridDim.x) {
    const int64_t f = pend_list[i];
    const float *xr = frames + f * dim;
    double v[KPL];
    unsigned long long cand = 0;
#pragma unroll
    for (int j = 0; j < KPL; j++) {
      const int c = j * 64 + lane;
      v[j] = 0.0;
      if (c < C) {
        const double *r = rec_fma + ((size_t)(c / 8) * dimp) * 16 + 2 * (size_t)(c % 8);
        double acc = 0.0;
        for (int d = 0; d < dim; d++) {
          const double x = (double)xr[d];
          acc = __builtin_fma(x, __builtin_fma(r[(size_t)d * 16 + 1], x, r[(size_t)d * 16]), acc);
        }
        v[j] = lin_key(acc + cconst_fma[c]);
        if (v[j] == v[j]) cand |= 1ull << j;
      }
    }
    int need_c = min_clusters, need_g = min_gaussians;
    double T = INFINITY;
    bool tie = false;
    if ((need_c > 0 || need_g > 0) && !force_heap) select_threshold<KPL, double>(v, cand, sz, lane, hist, need_c, need_g, T, tie);
    if (force_heap) tie = min_clusters > 0 || min_gaussians > 0;
    if (tie) {   // wave-uniform
      if (lane == 0) {
        const int at = atomicAdd(&tie_list[tie_cap], 1);
        if (at < tie_cap) tie_list[at] = (int32_t)f;
      }
      continue;
    }
    const int64_t word = f >> 6;
    const unsigned long long bit = 1ull << (f & 63);
    int exact = 0;
#pragma unroll
    for (int j = 0; j < KPL; j++) {
      const int c = j * 64 + lane;
      if (c < C) {
        const bool sel = v[j] >= T;
        exact += sel ? 1 : 0;
        const bool use_exact = sel || !(v[j] > AASR_KEY_ZERO);
        unsigned long long *w = maskw + word * (C + 1) + c;
        if (use_exact) atomicOr(w, bit);
        else atomicAnd(w, ~bit);
        cval[f * C + c] = centre_value(key32[f * Cs + c], use_exact, log_vals, ref);
      }
    }
    if (n_exact) {
      exact = wave_sum_i32(exact);
      if (lane == 0) n_exact[f] = exact;
    }
  }
}

// ------------------------------------------------------------ heap replay --
// PDFPool::precompute_likelihoods' cluster loop (aku/Distributions.cc:2684-2722) for the frames
// k_cluster_select could not settle: std::priority_queue<pair<int,double>, vector, cl_compare>
// (aku/Distributions.hh:291-299) is push_heap per cluster in index order, then top / pop_heap
// until both minimum counts are met; libstdc++'s __push_heap / __adjust_heap (bits/stl_heap.h)
// are replayed step by step, so equal likelihoods leave the queue in the reference build's order.
// One thread per listed frame; the heap lives in global scratch, [position][thread] so that the
// lanes of a wave touch neighbouring words.  pop_heap parks the popped element at the end of the
// shrinking array, so afterwards positions >= n hold the clusters that were popped.
__device__ __forceinline__ void heap_push(double *key, int32_t *idx, int64_t st, int hole, int top,
                                          double vk, int32_t vi) {
  int parent = (hole - 1) / 2;
  while (hole > top && key[parent * st] < vk) {
    key[hole * st] = key[parent * st];
    idx[hole * st] = idx[parent * st];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  key[hole * st] = vk;
  idx[hole * st] = vi;
}

__global__ __launch_bounds__(64) void k_cluster_select_heap(
    const double *__restrict__ ll64, int C, int64_t Cs, const int32_t *__restrict__ csize,
    int min_clusters, int min_gaussians, double ref, unsigned long long *__restrict__ maskw,
    float *__restrict__ cval, int32_t *__restrict__ n_exact, const int32_t *__restrict__ tie_list,
    int64_t tie_cap, double *__restrict__ heap_key, int32_t *__restrict__ heap_idx, int rows_by_position) {
  const int tid = blockIdx.x * 64 + threadIdx.x;
  const int64_t st = kHeapThreads;
  const bool log_vals = ref != ref;
  const int count = min((int64_t)tie_list[tie_cap], tie_cap);
  double *key = heap_key + tid;
  int32_t *idx = heap_idx + tid;
  for (int i = tid; i < count; i += kHeapThreads) {
    const int64_t f = tie_list[i];
    // the float selection's leftovers: row i of the list-mode centre pass holds the doubles of list entry i
    const double *row = ll64 + (rows_by_position ? (int64_t)i : f) * Cs;
    for (int c = 0; c < C; c++) {
      heap_push(key, idx, st, c, 0, row[c], c);
    }
    int n = C, clusters_done = 0, gauss_done = 0;
    while ((clusters_done < min_clusters || gauss_done < min_gaussians) && n > 0) {
      clusters_done++;
      gauss_done += csize[idx[0]];
      // pop_heap: the top goes to the last position, the former last element is sifted from the root
      const int len = n - 1;
      if (len >= 1) {
        const double vk = key[len * st];
        const int32_t vi = idx[len * st];
        key[len * st] = key[0];
        idx[len * st] = idx[0];
        int hole = 0, child = 0;
        while (child < (len - 1) / 2) {
          child = 2 * (child + 1);
          if (key[child * st] < key[(child - 1) * st]) child--;
          key[hole * st] = key[child * st];
          idx[hole * st] = idx[child * st];
          hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
          child = 2 * (child + 1);
          key[hole * st] = key[(child - 1) * st];
          idx[hole * st] = idx[(child - 1) * st];
          hole = child - 1;
        }
        heap_push(key, idx, st, hole, 0, vk, vi);
      }
      n--;
    }
    n_exact[f] = clusters_done;
    const int64_t word = f >> 6;
    const unsigned long long bit = 1ull << (f & 63);
    for (int p = 0; p < C; p++) {
      const int32_t c = idx[p * st];
      const bool use_exact = p >= n || !(key[p * st] > AASR_KEY_ZERO);
      unsigned long long *w = maskw + word * (C + 1) + c;
      if (use_exact) atomicOr(w, bit);
      else atomicAnd(w, ~bit);
      cval[f * C + c] = centre_value((float)key[p * st], use_exact, log_vals, ref);
    }
  }
}

// ----------------------------------------------------------------- expand --
// Per-lane selection bits for the track kernels (see ClusterArgs in gmm_score.hip): for the wave
// that owns the 64 frames of `word`, lane (n, h) and tile t,
//   bit ((mb*4 + q)*4 + e)*2 + side  =  selected(row t*64 + 32 mb + 8q + 4h + e, frame 32 side + n).
// A workgroup takes one word and a run of tiles: the word's cluster masks (C + 1 words, the last
// one = "in no cluster": all ones) are staged in LDS once, then each thread builds the words of
// its lane for tile after tile (32 row lookups shared by both sides).
constexpr int kExpandTiles = 4;  // tiles in flight per workgroup (256 threads = 4 waves, one tile each)

__global__ __launch_bounds__(256) void k_cluster_expand(
    const unsigned long long *__restrict__ maskw, int mask_stride,
    const int32_t *__restrict__ crow, int64_t rows_padded, int tiles_per_block,
    unsigned long long *__restrict__ maskrow) {
  extern __shared__ unsigned long long cm[];  // [mask_stride] cluster masks of this 64-frame word
  const int64_t word = blockIdx.y;
  for (int c = threadIdx.x; c < mask_stride; c += 256) cm[c] = maskw[word * mask_stride + c];
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int64_t n_tiles = rows_padded / TILE_ROWS;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_block;
  const int64_t t1 = t0 + tiles_per_block < n_tiles ? t0 + tiles_per_block : n_tiles;
  for (int64_t t = t0 + (threadIdx.x >> 6); t < t1; t += kExpandTiles) {
    // Lane r fetches the 64 frame bits of tile row r; a 64 x 64 bit transpose over the lanes (block swaps of
    // 32, 16, ... 1: six exchanges per dword instead of 32 LDS reads + 64 bit extracts per lane) leaves lane c
    // with the 64 row bits of frame c.  Lane (n, h) then needs frames n and 32 + n -- its own column and
    // lane ^ 32's -- restricted to the rows 8q + 4h + e of its track: nibble h of every byte, the rows of block
    // mb = 1 shifted into the free nibbles.  Bit layout of the word: side * 32 + 8q + 4mb + e.
    const unsigned long long r64 = cm[crow[t * TILE_ROWS + lane]];
    unsigned lo = (unsigned)r64, hi = (unsigned)(r64 >> 32);
    wave_bit_transpose64(lo, hi, lane);
    const unsigned olo = (unsigned)__shfl_xor((int)lo, 32), ohi = (unsigned)__shfl_xor((int)hi, 32);
    const unsigned alo = h ? olo : lo, ahi = h ? ohi : hi;   // frame n
    const unsigned blo = h ? lo : olo, bhi = h ? hi : ohi;   // frame 32 + n
    const int sh = 4 * h;
    const unsigned pa = ((alo >> sh) & 0x0F0F0F0Fu) | (((ahi >> sh) & 0x0F0F0F0Fu) << 4);
    const unsigned pb = ((blo >> sh) & 0x0F0F0F0Fu) | (((bhi >> sh) & 0x0F0F0F0Fu) << 4);
    maskrow[word * rows_padded + t * TILE_ROWS + lane] = (unsigned long long)pa | ((unsigned long long)pb << 32);
  }
}

// ------------------------------------------------------------------ merge --
// out[f][s] = log(max(exact part + sum_c W[s][c] * centre value(f, c), 1e-50)).
// A workgroup of 1024 threads owns kMergeFrames frames x 2048 states per step:
// the frames' centre values are staged in LDS transposed to [cluster][frame], so
// one 16-byte LDS read serves four frames of a (state, cluster) pair; every
// thread keeps the centre weights of its two states in registers.
constexpr int kMergeFrames = 8;
// k_cluster_expand stages one 64-frame word of every cluster's mask in LDS (8 bytes per cluster, 160 KB per workgroup)
constexpr int kMaxClusters = 16384;
constexpr int kMergeSPT = 2;  // states per thread

template <int NNZ, int kMergeThreads, bool DST>
__global__ __launch_bounds__(kMergeThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_cluster_merge(
    float *__restrict__ out, int64_t F, int64_t S, const float *__restrict__ cval, int C,
    const int32_t *__restrict__ w_cluster, const float *__restrict__ w_weight, int nnz,
    float ref, int frames_per_block, int cstride, int64_t pitch, const int32_t *__restrict__ colmap,
    float *__restrict__ dst, int64_t dst_pitch) {
  // DST (engine parts): the merged value goes to column = state of `dst` (the public layout) instead of back in place --
  // the exact parts are read through the column map, so the pass needs no gather of the columns afterwards (an instance
  // of its own: as a run-time branch the in-place form spilled four registers more)
  // [C][cstride]: cstride = 12 floats spreads the 16-byte reads of different
  // clusters over all banks (8 would put every read on 4 bank groups)
  extern __shared__ __attribute__((aligned(16))) float cv[];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  int64_t st[kMergeSPT], col[kMergeSPT];   // col: the state's column in a score row (engine parts: gmm_engine_colmap)
  bool live[kMergeSPT];
  unsigned wcp[kMergeSPT][(NNZ + 1) / 2];  // LDS offsets of the clusters, two 16-bit values per register
  float ww[kMergeSPT][NNZ];
#pragma unroll
  for (int u = 0; u < kMergeSPT; u++) {
    st[u] = ((int64_t)blockIdx.x * kMergeSPT + u) * kMergeThreads + tid;
    live[u] = st[u] < S;
    const int64_t sc = live[u] ? st[u] : S - 1;
    col[u] = colmap ? colmap[sc] : sc;
#pragma unroll
    for (int j = 0; j < NNZ; j += 2) {
      const unsigned c0 = j < nnz ? (unsigned)w_cluster[(int64_t)j * S + sc] : 0u;
      const unsigned c1 = j + 1 < nnz && j + 1 < NNZ ? (unsigned)w_cluster[(int64_t)(j + 1) * S + sc] : 0u;
      wcp[u][j / 2] = c0 | (c1 << 16);
    }
#pragma unroll
    for (int j = 0; j < NNZ; j++) ww[u][j] = j < nnz ? w_weight[(int64_t)j * S + sc] : 0.0f;
  }
  const int64_t f_begin = (int64_t)blockIdx.y * frames_per_block;
  const int64_t f_end = min(F, f_begin + frames_per_block);
  const float ref_ln = ref * 0.69314718055994530942f;
  // Software pipeline: the exact parts of step i+1 are requested before step i is computed (every
  // thread of the workgroup runs in lock step between the two barriers of a step, so nothing else
  // hides the global-memory latency; 85.7 -> 77.2 ms per clustered pass).  Carrying the centre
  // values the same way would need 8 more registers than the 128 a 1024-thread workgroup has
  // (measured with spills: slower).
  float onext[kMergeSPT][kMergeFrames];
  auto request = [&](int64_t fg) {
    const int nf = (int)min((int64_t)kMergeFrames, f_end - fg);
#pragma unroll
    for (int u = 0; u < kMergeSPT; u++)
#pragma unroll
      for (int k = 0; k < kMergeFrames; k++)
        onext[u][k] = (live[u] && k < nf) ? out[(fg + k) * pitch + col[u]] : 0.0f;
  };
  if (f_begin < f_end) request(f_begin);
  for (int64_t fg = f_begin; fg < f_end; fg += kMergeFrames) {
    const int nf = (int)min((int64_t)kMergeFrames, f_end - fg);
    __syncthreads();
    // frame-major so that no thread divides by C: the kMergeFrames loads of a cluster are independent
    for (int c = tid; c < C; c += kMergeThreads) {
      float v[kMergeFrames];
#pragma unroll
      for (int k = 0; k < kMergeFrames; k++) v[k] = k < nf ? cval[(fg + k) * C + c] : 0.0f;
#pragma unroll
      for (int k = 0; k < kMergeFrames; k++) cv[c * cstride + k] = v[k];
    }
    float ocur[kMergeSPT][kMergeFrames];
#pragma unroll
    for (int u = 0; u < kMergeSPT; u++)
#pragma unroll
      for (int k = 0; k < kMergeFrames; k++) ocur[u][k] = onext[u][k];
    __syncthreads();
    if (fg + kMergeFrames < f_end) request(fg + kMergeFrames);
#pragma unroll
    for (int u = 0; u < kMergeSPT; u++) {
      if (!live[u]) continue;
      // x = log2 of the exact part on the shared reference exponent.  Outlier-routed Gaussians
      // (centred kernel) can carry constants the track rows' reference was not chosen for: above
      // 2^60 the exact part is kept in the log domain and the centres' share enters as log2(1 + A 2^-x)
      float lin[kMergeFrames], xk[kMergeFrames];
#pragma unroll
      for (int k = 0; k < kMergeFrames; k++) {
        xk[k] = fmaf(ocur[u][k], 1.4426950408889634f, ref);
        lin[k] = (k < nf && xk[k] <= 60.0f) ? exp2f(xk[k]) : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < NNZ; j++) {
        const unsigned cidx = (j & 1) ? (wcp[u][j / 2] >> 16) : (wcp[u][j / 2] & 0xffffu);
        const f32x4 *p = (const f32x4 *)(cv + cidx * cstride);
        const f32x4 a = p[0], b = p[1];
        const float w = ww[u][j];
        lin[0] = fmaf(w, a.x, lin[0]);
        lin[1] = fmaf(w, a.y, lin[1]);
        lin[2] = fmaf(w, a.z, lin[2]);
        lin[3] = fmaf(w, a.w, lin[3]);
        lin[4] = fmaf(w, b.x, lin[4]);
        lin[5] = fmaf(w, b.y, lin[5]);
        lin[6] = fmaf(w, b.z, lin[6]);
        lin[7] = fmaf(w, b.w, lin[7]);
      }
      for (int j = NNZ; j < nnz; j++) {
        const float w = w_weight[(int64_t)j * S + st[u]];
        const float *p = cv + w_cluster[(int64_t)j * S + st[u]] * cstride;
#pragma unroll
        for (int k = 0; k < kMergeFrames; k++) lin[k] = fmaf(w, p[k], lin[k]);
      }
#pragma unroll
      for (int k = 0; k < kMergeFrames; k++)
        if (k < nf) {
          const float l2 = xk[k] <= 60.0f ? __log2f(lin[k]) : xk[k] + __log2f(1.0f + lin[k] * exp2f(-xk[k]));
          const float l = fmaf(l2, 0.69314718055994530942f, -ref_ln);
          if (DST) dst[(fg + k) * dst_pitch + st[u]] = fmaxf(l, AASR_LOG_TINY_F);
          else out[(fg + k) * pitch + col[u]] = fmaxf(l, AASR_LOG_TINY_F);
        }
    }
  }
}

// ------------------------------------------------------------------- host --

// Centre of one cluster: Gaussian::merge with unit weights into a
// DiagonalGaussian (aku/Distributions.cc:853-898, 1208-1228, 1273-1288).
static void merge_centre(const HostModel &m, const std::vector<int32_t> &members, double *mean,
                         double *prec, double *cst) {
  const int D = m.dim;
  double weight_sum = 0;
  for (size_t i = 0; i < members.size(); i++) weight_sum += 1.0;
  if (weight_sum < 1e-15) weight_sum = 1;
  const double scale = 1.0 / weight_sum;
  double c = 1;
  for (int d = 0; d < D; d++) {
    double nm = 0, nc = 0;
    for (int32_t g : members) {
      const double mu = m.mean[(size_t)g * D + d];
      // Gaussian::merge reads every member's covariance (get_covariance) and the diagonal target keeps the
      // diagonal of the result: for a full-covariance member that is the diagonal of its matrix
      const double var = (m.any_full() && m.is_full[(size_t)g]) ? m.cov[((size_t)g * D + d) * D + d]
                                                                 : m.var[(size_t)g * D + d];
      nc += 1.0 * (var + mu * mu);
      nm += 1.0 * mu;
    }
    nm *= scale;
    nc *= scale;
    nc += -1.0 * nm * nm;
    mean[d] = nm;
    prec[d] = nc > 0 ? 1 / nc : 0;
  }
  for (int d = 0; d < D; d++) c *= prec[d];
  if (c > 0) c = std::log(std::sqrt(c));
  *cst = c;
}

// cluster of every packed row of a track layout
static void build_crow(const aasr_gmm *g, const TrackLayout &L, const ClusterState &cl,
                       DevBuf<int32_t> &out) {
  std::vector<int32_t> crow(L.row_gauss.size(), cl.C);
  for (size_t r = 0; r < crow.size(); r++) {
    const int32_t gi = L.row_gauss[r];
    if (gi >= 0 && cl.g2c[(size_t)gi] >= 0) crow[r] = cl.g2c[(size_t)gi];
  }
  out.upload(crow.data(), crow.size());
}

// cluster of every record of the centred kernel's operand (comps = mixture-component indices; empty =
// every component in order)
static void build_crow_comps(const aasr_gmm *g, const std::vector<int32_t> &comps, const ClusterState &cl,
                             DevBuf<int32_t> &out) {
  const HostModel &m = g->host;
  const size_t n = comps.empty() ? m.mix_idx.size() : comps.size();
  std::vector<int32_t> crow(std::max<size_t>(n, 1), cl.C);
  for (size_t r = 0; r < n; r++) {
    const int32_t gi = m.mix_idx[comps.empty() ? r : (size_t)comps[r]];
    if (cl.g2c[(size_t)gi] >= 0) crow[r] = cl.g2c[(size_t)gi];
  }
  out.upload(crow.data(), crow.size());
}

void gmm_set_clustering(aasr_gmm *g, int32_t n_clusters, int64_t n_pairs,
                        const int32_t *gauss_index, const int32_t *cluster_index) {
  if (!g->dim_parts.empty() && ((g->host.dim + 7) / 8 * 8) * (2 * 16 * 8 + 64 * 4) > 160 * 1024)
    raise(AASR_ERR_UNSUPPORTED, "Gaussian clustering on a model of %d dimensions: up to ~300 dimensions", g->host.dim);
  ClusterState &cl = g->cl;
  if (n_clusters <= 0) {
    cl = ClusterState();
    return;
  }
  const HostModel &m = g->host;
  if (m.any_full() && m.n_transforms > 0)
    raise(AASR_ERR_UNSUPPORTED, "Gaussian clustering over an adapted full-covariance pool is not built");
  if (n_clusters > 0.3 * (double)m.G)
    raise(AASR_ERR_INVALID,
          "PDFPool::read_clustering(): Number of clusters (%d) seems insensible compared to the "
          "number of Gaussians (%ld).", n_clusters, (long)m.G);
  if (n_clusters > kMaxClusters)
    raise(AASR_ERR_UNSUPPORTED, "more than %d clusters are not built (%d asked)", kMaxClusters, n_clusters);
  if (n_pairs > 0 && (!gauss_index || !cluster_index))
    raise(AASR_ERR_INVALID, "aasr_gmm_set_clustering: null argument");
  if (g->dim_parts.empty() && !g->class_routing && !g->paired.ok && !g->tracks.ok && !g->centred_ok && !(m.any_full() && g->full.ok))
    raise(AASR_ERR_UNSUPPORTED,
          "Gaussian clustering needs the fixed-reference track kernels or the centred kernel, and this "
          "model has neither");
  std::vector<std::vector<int32_t>> members((size_t)n_clusters);
  std::vector<int32_t> g2c((size_t)m.G, -1);
  for (int64_t i = 0; i < n_pairs; i++) {
    const int32_t gi = gauss_index[i], ci = cluster_index[i];
    if (gi < 0 || gi >= m.G)
      raise(AASR_ERR_INVALID, "PDFPool::read_clustering(): Gauss index out of bounds");
    if (ci < 0 || ci >= n_clusters)
      raise(AASR_ERR_INVALID, "PDFPool::read_clustering(): Cluster index out of bounds");
    if (g2c[(size_t)gi] >= 0 && g2c[(size_t)gi] != ci)
      raise(AASR_ERR_INVALID, "Gaussian %d is listed in clusters %d and %d", gi, g2c[(size_t)gi], ci);
    g2c[(size_t)gi] = ci;
    members[(size_t)ci].push_back(gi);
  }
  ClusterState n;
  n.C = n_clusters;
  n.Cs = (n_clusters + 7) / 8 * 8;
  n.g2c = g2c;
  n.dimp = (m.dim + 7) / 8 * 8;
  // the linear centre values and the merge share the track kernels' reference exponent.  A model
  // without track layouts (only the centred kernel scores it), or a centre whose constant does not
  // fit under that exponent (clusters of variance-floored Gaussians: +121 nats at sigma 0.045), takes
  // the log-domain merge instead: centre values as log2, per-state (max, sum) -- no range limit.
  n.log_merge = !g->paired.ok && !g->tracks.ok;
  // (a model scored through engine parts: its first part's layout stands in for the model's own)
  const TrackLayout *part_layout = (!g->engine_parts.empty() && g->engine_parts[0].model->paired.ok)
                                       ? &g->engine_parts[0].model->paired : nullptr;
  if (n.log_merge && part_layout) n.log_merge = false;
  // the linear merge stages a frame group's centre values in LDS, [cluster][8 frames]: beyond 4 608 clusters they do
  // not fit, and the log-domain merge reads them through the caches
  if ((size_t)n_clusters * kMergeFrames * sizeof(float) > 144 * 1024) n.log_merge = true;
  n.ref_log2 = g->paired.ok ? g->paired.ref_log2 : g->tracks.ok ? g->tracks.ref_log2 : part_layout ? part_layout->ref_log2 : 0.0;
  n.csize_h.resize((size_t)n.Cs, 0);
  n.c_mean.assign((size_t)n.C * m.dim, 0.0);
  n.c_prec.assign((size_t)n.C * m.dim, 0.0);
  n.c_cst.assign((size_t)n.Cs, 0.0);
  std::vector<double> rec((size_t)n.Cs / 8 * n.dimp * 16, 0.0);
  for (int c = 0; c < n.C; c++) {
    n.csize_h[(size_t)c] = (int32_t)members[(size_t)c].size();
    merge_centre(m, members[(size_t)c], &n.c_mean[(size_t)c * m.dim], &n.c_prec[(size_t)c * m.dim],
                 &n.c_cst[(size_t)c]);
    // the linear centre values share the track kernels' reference exponent
    if (!(n.c_cst[(size_t)c] * kLog2eD + n.ref_log2 <= 126.0)) n.log_merge = true;
    for (int d = 0; d < m.dim; d++) {
      const size_t at = ((size_t)(c / 8) * n.dimp + d) * 16 + 2 * (size_t)(c % 8);
      rec[at] = n.c_mean[(size_t)c * m.dim + d];
      rec[at + 1] = n.c_prec[(size_t)c * m.dim + d];
    }
  }
  n.rec.upload(rec.data(), rec.size());
  n.cconst.upload(n.c_cst.data(), n.c_cst.size());
  {  // expanded-form records (a, b) = (p mu, -p / 2) and constants c - 1/2 sum p mu^2 for k_cluster_centres_fma
    std::vector<double> rec2(rec.size(), 0.0), cst2((size_t)n.Cs, 0.0);
    for (int c = 0; c < n.C; c++) {
      double q = 0;
      for (int d = 0; d < m.dim; d++) {
        const double mu = n.c_mean[(size_t)c * m.dim + d], pr = n.c_prec[(size_t)c * m.dim + d];
        const size_t at = ((size_t)(c / 8) * n.dimp + d) * 16 + 2 * (size_t)(c % 8);
        rec2[at] = pr * mu;
        rec2[at + 1] = -0.5 * pr;
        q += pr * mu * mu;
      }
      cst2[(size_t)c] = n.c_cst[(size_t)c] - 0.5 * q;
    }
    n.rec_fma.upload(rec2.data(), rec2.size());
    n.cconst_fma.upload(cst2.data(), cst2.size());
    // k_cluster_centres_mfma: [tile of 16 clusters][k step][lane] = coefficient K = 4 q + lane / 16 of cluster
    // 16 t + lane % 16 (K < dim: p mu; < 2 dim: -p / 2; == 2 dim: the constant; Cs is a multiple of 8, the tiles
    // round up to 16 with zero columns that land in the padding of a row of ll64 or are skipped)
    n.mfma_ks = (2 * m.dim + 1 + 3) / 4;
    const int tiles16 = (n.Cs + 15) / 16;
    std::vector<double> bp((size_t)tiles16 * n.mfma_ks * 64, 0.0);
    for (int c = 0; c < n.C; c++)
      for (int K = 0; K <= 2 * m.dim; K++) {
        double v;
        if (K < m.dim) v = n.c_prec[(size_t)c * m.dim + K] * n.c_mean[(size_t)c * m.dim + K];
        else if (K < 2 * m.dim) v = -0.5 * n.c_prec[(size_t)c * m.dim + (K - m.dim)];
        else v = cst2[(size_t)c];
        bp[((size_t)(c / 16) * n.mfma_ks + K / 4) * 64 + (size_t)(K % 4) * 16 + (c % 16)] = v;
      }
    n.bpack.upload(bp.data(), bp.size());
  }
  n.csize.upload(n.csize_h.data(), n.csize_h.size());
  // W[s][c]: weight of state s carried by cluster c (components outside every cluster are
  // always exact and carry none)
  std::vector<std::map<int32_t, double>> W((size_t)m.S);
  int nnz = 1;
  for (int64_t s = 0; s < m.S; s++) {
    for (int32_t k = m.mix_off[s]; k < m.mix_off[s + 1]; k++) {
      const int32_t c = g2c[(size_t)m.mix_idx[k]];
      if (c >= 0) W[(size_t)s][c] += m.mix_w[(size_t)k];
    }
    nnz = std::max(nnz, (int)W[(size_t)s].size());
  }
  std::vector<int32_t> wc((size_t)nnz * m.S, 0);
  std::vector<float> ww((size_t)nnz * m.S, 0.0f);
  for (int64_t s = 0; s < m.S; s++) {
    int j = 0;
    for (auto &kv : W[(size_t)s]) {
      wc[(size_t)j * m.S + s] = kv.first;
      ww[(size_t)j * m.S + s] = (float)kv.second;
      j++;
    }
  }
  n.nnz = nnz;
  n.w_cluster.upload(wc.data(), wc.size());
  n.w_weight.upload(ww.data(), ww.size());
  if (g->paired.ok) build_crow(g, g->paired, n, n.crow[0]);
  if (g->tracks.ok) build_crow(g, g->tracks, n, n.crow[1]);
  n.loaded = true;
  for (auto &sub : g->class_models)
    if (sub) sub->cl = ClusterState();  // the classes' views of the previous clustering
  for (auto &part : g->engine_parts)    // ... and the engine parts' (rebuilt by the next clustered pass: another
    if (part.model) part.model->cl = ClusterState();   // assignment with the same cluster count must not keep the old rows)
  // thresholds and the on/off state survive a re-read like the reference's members do
  n.enabled = cl.enabled;
  n.min_clusters = cl.min_clusters;
  n.min_gaussians = cl.min_gaussians;
  cl = std::move(n);
}

// PDFPool::read_clustering's reader (aku/Distributions.cc:3121-3148):
//   in >> nclusters;  while (in) { int g, c; in >> g >> c; checks; push; }
// The iteration that runs into end-of-file still executes its body with the
// operands of the previous iteration (the extraction leaves them untouched when
// the stream sentry fails), so the last pair of a file is pushed twice: that
// Gaussian is merged into its centre with weight 2 and counted twice by the
// min-Gaussians test.  Kept.
void gmm_read_clustering(aasr_gmm *g, const char *path) {
  std::ifstream in(path);
  if (!in)
    raise(AASR_ERR_IO, "PDFPool::read_clustering(): could not open %s", path);
  long n = 0;
  if (!(in >> n) || n <= 0) raise(AASR_ERR_INVALID, "%s: cluster count missing", path);
  std::vector<int32_t> gi, ci;
  long a, b;
  while (in >> a) {
    if (!(in >> b)) raise(AASR_ERR_INVALID, "%s: Gaussian index %ld without a cluster index", path, a);
    if (a < 0 || b < 0) raise(AASR_ERR_INVALID, "%s: negative index", path);
    gi.push_back((int32_t)a);
    ci.push_back((int32_t)b);
  }
  if (!in.eof()) raise(AASR_ERR_INVALID, "%s: unexpected text after %zu pairs", path, gi.size());
  if (gi.empty()) raise(AASR_ERR_INVALID, "%s: no Gaussian-cluster pairs", path);
  gi.push_back(gi.back());
  ci.push_back(ci.back());
  gmm_set_clustering(g, (int32_t)n, (int64_t)gi.size(), gi.data(), ci.data());
}

void gmm_set_clustering_min_evals(aasr_gmm *g, double min_clusters, double min_gaussians) {
  ClusterState &cl = g->cl;
  if (!cl.loaded)
    raise(AASR_ERR_INVALID, "set_clustering_min_evals(): no clustering has been read");
  // HmmSet::set_clustering_min_evals (aku/HmmSet.cc:1359-1366)
  cl.min_clusters = (int)(min_clusters * cl.C);
  cl.min_gaussians = (int)(min_gaussians * (double)g->G);
  cl.enabled = true;
}

// keys: 0 doubles per frame, 1 floats per frame (cl.key32), 2 list mode (the frames of cl.tie_list, doubles by position)
template <int KS>
static void launch_centres_mfma_t(aasr_gmm *g, const float *d_frames, int64_t F, hipStream_t stream, int keys) {
  ClusterState &cl = g->cl;
  const int tiles = (int)((cl.Cs + 15) / 16);
  const int64_t bx = (F + 32 * kMfmaCentreWaves - 1) / (32 * kMfmaCentreWaves);
  // cut the cluster tiles over blockIdx.y so that the grid fills the chip in whole rounds (2 workgroups of 8
  // waves per CU)
  const double slots = 2.0 * (double)(g->num_cus > 0 ? g->num_cus : 256);
  int ny = 1;
  double best = 0;
  for (int y = 1; y <= std::min(tiles, 16); y++) {
    const int ty = (tiles + y - 1) / y;
    const int yy = (tiles + ty - 1) / ty;
    const double x = (double)bx * yy / slots;
    const double eff = x < 1.0 ? x : x / std::ceil(x);
    if (eff > best + 0.02) {
      best = eff;
      ny = yy;
    }
  }
  const int tpy = (tiles + ny - 1) / ny;
  ny = (tiles + tpy - 1) / tpy;
  if (keys == 2) {
    // normally an empty list: few workgroups, each walking the list in strides of the grid
    const int64_t gx = std::min<int64_t>(bx, std::max<int64_t>(1, (int64_t)slots / ny));
    hipLaunchKernelGGL(k_cluster_centres_mfma<KS>, dim3((unsigned)gx, (unsigned)ny), dim3(64 * kMfmaCentreWaves), 0,
                       stream, d_frames, F, g->dim, cl.bpack.p, tiles, tpy, cl.ll64.p, (int64_t)cl.Cs, (float *)nullptr,
                       cl.tie_list.p, (int64_t)cl.Fs);
  } else {
    hipLaunchKernelGGL(k_cluster_centres_mfma<KS>, dim3((unsigned)bx, (unsigned)ny), dim3(64 * kMfmaCentreWaves), 0,
                       stream, d_frames, F, g->dim, cl.bpack.p, tiles, tpy, cl.ll64.p, (int64_t)cl.Cs,
                       keys == 1 ? cl.key32.p : (float *)nullptr, (const int32_t *)nullptr, (int64_t)0);
  }
  AASR_HIP(hipGetLastError());
}

static bool launch_centres_mfma(aasr_gmm *g, const float *d_frames, int64_t F, hipStream_t stream, int keys = 0) {
  switch (g->cl.mfma_ks) {
#define AASR_CASE(N)                                        \
  case N:                                                   \
    launch_centres_mfma_t<N>(g, d_frames, F, stream, keys); \
    return true;
    // k steps of 4 covering K = 2 dim + 1: every dimension up to 63
    AASR_CASE(1) AASR_CASE(2) AASR_CASE(3) AASR_CASE(4) AASR_CASE(5) AASR_CASE(6) AASR_CASE(7) AASR_CASE(8)
    AASR_CASE(9) AASR_CASE(10) AASR_CASE(11) AASR_CASE(12) AASR_CASE(13) AASR_CASE(14) AASR_CASE(15) AASR_CASE(16)
    AASR_CASE(17) AASR_CASE(18) AASR_CASE(19) AASR_CASE(20) AASR_CASE(21) AASR_CASE(22) AASR_CASE(23) AASR_CASE(24)
    AASR_CASE(25) AASR_CASE(26) AASR_CASE(27) AASR_CASE(28) AASR_CASE(29) AASR_CASE(30) AASR_CASE(31) AASR_CASE(32)
#undef AASR_CASE
    default:
      return false;   // (the vector kernel)
  }
}

template <typename XT>
static void launch_centres(aasr_gmm *g, const XT *d_frames, int64_t F, hipStream_t stream) {
  ClusterState &cl = g->cl;
  const int groups = cl.Cs / 8;
  const int64_t bx = (F + kCentreThreads - 1) / kCentreThreads;
  const int smem = 2 * cl.dimp * 16 * 8 + cl.dimp * kCentreThreads * (int)sizeof(XT);
  // cut the cluster groups over blockIdx.y so that the grid is a whole number of rounds over the
  // resident workgroup slots (a workgroup walks all its groups: with one cut, 977 workgroups on
  // 768 slots ran 1.27 rounds at half occupancy)
  const double slots = (double)(g->num_cus > 0 ? g->num_cus : 256) * std::max(1, (160 * 1024) / std::max(smem, 1));
  int ny = 1;
  double best = 0;
  for (int y = 1; y <= std::min(groups, 16); y++) {
    const int gy = (groups + y - 1) / y;
    const int yy = (groups + gy - 1) / gy;
    const double x = (double)bx * yy / slots;
    const double eff = x < 1.0 ? x : x / std::ceil(x);
    if (eff > best + 0.02) {
      best = eff;
      ny = yy;
    }
  }
  const int gpy = (groups + ny - 1) / ny;
  ny = (groups + gpy - 1) / gpy;
  static bool attr_set[64] = {false};  // one flag array per instantiation (XT)
  if (!attr_set[g->device & 63]) {
    AASR_HIP(hipFuncSetAttribute((const void *)k_cluster_centres<XT>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                 2 * 64 * 16 * 8 + 64 * kCentreThreads * (int)sizeof(XT)));
    attr_set[g->device & 63] = true;
  }
  // float frames: the expanded FMA form (AASR_CLUSTER_CENTRES_REF=1 keeps the reference operation order)
  static const int ref_order = AASR_EXPERIMENT_ENV("AASR_CLUSTER_CENTRES_REF") ? atoi(AASR_EXPERIMENT_ENV("AASR_CLUSTER_CENTRES_REF")) : 0;
  if constexpr (std::is_same<XT, float>::value) {
    if (cl.dimp > 64) {
      // feature dimension > 63 (no matrix instance: K = 2 dim + 1 > 128): the expanded FMA form with one wave per
      // workgroup, so that a frame column of any length fits LDS
      if (!cl.rec_fma.p) raise(AASR_ERR_UNSUPPORTED, "no centre kernel for dimension %d", g->dim);
      const int nt = 64;
      const int smem_w = 2 * cl.dimp * 16 * 8 + cl.dimp * nt * (int)sizeof(float);
      if (smem_w > 160 * 1024) raise(AASR_ERR_UNSUPPORTED, "Gaussian clustering: dimension %d does not fit the centre kernel", g->dim);
      static bool attr_wide[64] = {false};
      if (!attr_wide[g->device & 63]) {
        AASR_HIP(hipFuncSetAttribute((const void *)k_cluster_centres_fma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_wide[g->device & 63] = true;
      }
      const int64_t bxw = (F + nt - 1) / nt;
      hipLaunchKernelGGL(k_cluster_centres_fma, dim3((unsigned)bxw, 1), dim3(nt), smem_w, stream, d_frames, F, g->dim, cl.dimp,
                         cl.rec_fma.p, cl.cconst_fma.p, groups, groups, cl.ll64.p, (int64_t)cl.Cs);
      AASR_HIP(hipGetLastError());
      return;
    }
    if (!ref_order && cl.rec_fma.p) {
      static bool attr_fma[64] = {false};
      if (!attr_fma[g->device & 63]) {
        AASR_HIP(hipFuncSetAttribute((const void *)k_cluster_centres_fma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
        attr_fma[g->device & 63] = true;
      }
      // the matrix-pipe form where an instance exists for the dimension
      static const int use_mfma = AASR_EXPERIMENT_ENV("AASR_CLUSTER_CENTRES_MFMA") ? atoi(AASR_EXPERIMENT_ENV("AASR_CLUSTER_CENTRES_MFMA")) : 1;
      if (use_mfma && cl.bpack.p && launch_centres_mfma(g, d_frames, F, stream)) return;
      hipLaunchKernelGGL(k_cluster_centres_fma, dim3((unsigned)bx, (unsigned)ny), dim3(kCentreThreads), smem, stream,
                         d_frames, F, g->dim, cl.dimp, cl.rec_fma.p, cl.cconst_fma.p, groups, gpy, cl.ll64.p,
                         (int64_t)cl.Cs);
      AASR_HIP(hipGetLastError());
      return;
    }
  }
  hipLaunchKernelGGL(k_cluster_centres<XT>, dim3((unsigned)bx, (unsigned)ny), dim3(kCentreThreads), smem,
                     stream, d_frames, F, g->dim, cl.dimp, cl.rec.p, cl.cconst.p, groups, gpy,
                     cl.ll64.p, (int64_t)cl.Cs);
  AASR_HIP(hipGetLastError());
}

// sub-pass [s0, s0 + F) of the current pass: masks, centre values and counts land at
// their place in the pass-wide buffers (s0 is a multiple of 64)
// d_tie_frames: the float-key fast path -- the frames of this sub-pass, so that the leftovers' exact keys can be
// computed (list-mode centre pass) before the replay; null: the keys are doubles in cl.ll64, one row per frame
template <int KPL>
static void launch_select_t(aasr_gmm *g, int64_t s0, int64_t F, hipStream_t stream, const float *d_tie_frames) {
  ClusterState &cl = g->cl;
  const int64_t words = (F + 63) / 64;
  const int force_heap = g_force_heap;
  const double ref_arg = cl.log_merge ? (double)NAN : cl.ref_log2;
  AASR_HIP(hipMemsetAsync(cl.tie_list.p + cl.Fs, 0, sizeof(int32_t), stream));
  if (d_tie_frames) {
    AASR_HIP(hipMemsetAsync(cl.pend_list.p + cl.Fs, 0, sizeof(int32_t), stream));
    hipLaunchKernelGGL((k_cluster_select<KPL, float>), dim3((unsigned)words), dim3(64), 0, stream, cl.key32.p, F,
                       cl.C, (int64_t)cl.Cs, cl.csize.p, cl.min_clusters, cl.min_gaussians, ref_arg,
                       cl.maskw.p + (size_t)(s0 / 64) * (cl.C + 1), cl.cval.p + (size_t)s0 * cl.C,
                       cl.n_exact.p + s0, cl.pend_list.p, cl.Fs, force_heap);
    AASR_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_cluster_select_pending<KPL>, dim3(256), dim3(64), 0, stream, d_tie_frames, g->dim, cl.dimp,
                       cl.rec_fma.p, cl.cconst_fma.p, cl.key32.p, cl.C, (int64_t)cl.Cs, cl.csize.p, cl.min_clusters,
                       cl.min_gaussians, ref_arg, cl.maskw.p + (size_t)(s0 / 64) * (cl.C + 1), cl.cval.p + (size_t)s0 * cl.C,
                       cl.n_exact.p + s0, cl.pend_list.p, (int64_t)cl.Fs, cl.tie_list.p, (int64_t)cl.Fs, force_heap);
    AASR_HIP(hipGetLastError());
    if (!launch_centres_mfma(g, d_tie_frames, F, stream, 2)) raise(AASR_ERR_UNSUPPORTED, "no matrix centre kernel");
  } else {
    hipLaunchKernelGGL((k_cluster_select<KPL, double>), dim3((unsigned)words), dim3(64), 0, stream, cl.ll64.p, F,
                       cl.C, (int64_t)cl.Cs, cl.csize.p, cl.min_clusters, cl.min_gaussians, ref_arg,
                       cl.maskw.p + (size_t)(s0 / 64) * (cl.C + 1), cl.cval.p + (size_t)s0 * cl.C,
                       cl.n_exact.p + s0, cl.tie_list.p, cl.Fs, force_heap);
    AASR_HIP(hipGetLastError());
  }
  // frames left to the queue replay (normally none: the kernel's threads find an empty list)
  hipLaunchKernelGGL(k_cluster_select_heap, dim3(kHeapThreads / 64), dim3(64), 0, stream, cl.ll64.p, cl.C,
                     (int64_t)cl.Cs, cl.csize.p, cl.min_clusters, cl.min_gaussians, ref_arg,
                     cl.maskw.p + (size_t)(s0 / 64) * (cl.C + 1), cl.cval.p + (size_t)s0 * cl.C,
                     cl.n_exact.p + s0, cl.tie_list.p, cl.Fs, cl.heap_key.p, cl.heap_idx.p, d_tie_frames ? 1 : 0);
  AASR_HIP(hipGetLastError());
}

// More than 64 x 64 clusters: a lane of the selection wave cannot hold its share of a frame's keys.  Every frame then
// takes the queue replay (k_cluster_select_heap, one thread per frame: exact, an order of magnitude slower -- the
// reference's own clusterings have ~1 000 clusters); the masks start out empty with the "in no cluster" column set.
__global__ void k_cluster_all_to_replay(unsigned long long *__restrict__ maskw, int64_t words, int c1,
                                        int32_t *__restrict__ tie_list, int64_t tie_cap, int64_t F) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words * c1) maskw[i] = (i % c1 == c1 - 1) ? ~0ull : 0ull;
  if (i < F) tie_list[i] = (int32_t)i;
  if (i == 0) tie_list[tie_cap] = (int32_t)F;
}

static void launch_select_replay(aasr_gmm *g, int64_t s0, int64_t F, hipStream_t stream) {
  ClusterState &cl = g->cl;
  const int64_t words = (F + 63) / 64;
  const double ref_arg = cl.log_merge ? (double)NAN : cl.ref_log2;
  const int64_t n = std::max<int64_t>(words * (cl.C + 1), F);
  hipLaunchKernelGGL(k_cluster_all_to_replay, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                     cl.maskw.p + (size_t)(s0 / 64) * (cl.C + 1), words, cl.C + 1, cl.tie_list.p, cl.Fs, F);
  AASR_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_cluster_select_heap, dim3(kHeapThreads / 64), dim3(64), 0, stream, cl.ll64.p, cl.C,
                     (int64_t)cl.Cs, cl.csize.p, cl.min_clusters, cl.min_gaussians, ref_arg,
                     cl.maskw.p + (size_t)(s0 / 64) * (cl.C + 1), cl.cval.p + (size_t)s0 * cl.C,
                     cl.n_exact.p + s0, cl.tie_list.p, cl.Fs, cl.heap_key.p, cl.heap_idx.p, 0);
  AASR_HIP(hipGetLastError());
}

static void launch_select(aasr_gmm *g, int64_t s0, int64_t F, hipStream_t stream, const float *d_tie_frames = nullptr) {
  const int kpl = (g->cl.C + 63) / 64;
  if (kpl > 64) {
    launch_select_replay(g, s0, F, stream);
    return;
  }
  if (kpl <= 4) launch_select_t<4>(g, s0, F, stream, d_tie_frames);
  else if (kpl <= 16) launch_select_t<16>(g, s0, F, stream, d_tie_frames);
  else if (kpl <= 32) launch_select_t<32>(g, s0, F, stream, d_tie_frames);
  else launch_select_t<64>(g, s0, F, stream, d_tie_frames);
}

// The float-key fast path of a sub-pass: float frames, a matrix centre kernel for the dimension, at most 64 x 64
// clusters (AASR_CLUSTER_KEYS64=1 keeps the doubles throughout).
static bool cluster_fast_keys(const aasr_gmm *g) {
  static const int keys64 = AASR_EXPERIMENT_ENV("AASR_CLUSTER_KEYS64") ? atoi(AASR_EXPERIMENT_ENV("AASR_CLUSTER_KEYS64")) : 0;
  static const int ref_order = AASR_EXPERIMENT_ENV("AASR_CLUSTER_CENTRES_REF") ? atoi(AASR_EXPERIMENT_ENV("AASR_CLUSTER_CENTRES_REF")) : 0;
  static const int use_mfma = AASR_EXPERIMENT_ENV("AASR_CLUSTER_CENTRES_MFMA") ? atoi(AASR_EXPERIMENT_ENV("AASR_CLUSTER_CENTRES_MFMA")) : 1;
  const ClusterState &cl = g->cl;
  return !keys64 && !ref_order && use_mfma && cl.dimp <= 64 && cl.bpack.p && cl.rec_fma.p && cl.mfma_ks >= 1 && cl.mfma_ks <= 32 &&
         (cl.C + 63) / 64 <= 64;
}

// The merge in the log domain (ClusterState::log_merge): out = ln(e^out + sum_j w_j 2^cvl[c_j]) with a
// per-state maximum, for models whose centre values have no common f32 exponent.  Thread = state,
// blockIdx.y = frame; the frame's C log2 centre values come through the caches.  A fallback: 16 exp2
// per state and frame.
__global__ __launch_bounds__(256) void k_cluster_merge_log(float *__restrict__ out, int64_t F, int64_t S,
                                                           const float *__restrict__ cvl, int C,
                                                           const int32_t *__restrict__ w_cluster,
                                                           const float *__restrict__ w_weight, int nnz, int64_t pitch,
                                                           const int32_t *__restrict__ colmap) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const int64_t sc = colmap ? colmap[s] : s;
  for (int64_t f = blockIdx.y; f < F; f += gridDim.y) {
    const float *cv = cvl + f * C;
    const float x = out[f * pitch + sc] * 1.4426950408889634f;
    float m = x;
    for (int j = 0; j < nnz; j++) {
      const float w = w_weight[(int64_t)j * S + s];
      if (w > 0.0f) m = fmaxf(m, __log2f(w) + cv[w_cluster[(int64_t)j * S + s]]);
    }
    float l = AASR_LOG_TINY_F;
    if (m > -INFINITY) {
      float sum = exp2f(x - m);
      for (int j = 0; j < nnz; j++) {
        const float w = w_weight[(int64_t)j * S + s];
        if (w > 0.0f) sum += exp2f(__log2f(w) + cv[w_cluster[(int64_t)j * S + s]] - m);
      }
      l = (m + __log2f(sum)) * 0.69314718055994530942f;
    }
    out[f * pitch + sc] = fmaxf(l, AASR_LOG_TINY_F);
  }
}

static void launch_merge(aasr_gmm *g, float *d_out, int64_t F, int64_t pitch, hipStream_t stream, const int32_t *colmap = nullptr,
                         float *dst = nullptr, int64_t dst_pitch = 0);

template <int NNZ, int kMergeThreads, bool DST>
static void launch_merge_tt(aasr_gmm *g, float *d_out, int64_t F, int64_t pitch, hipStream_t stream, const int32_t *colmap,
                           float *dst, int64_t dst_pitch) {
  ClusterState &cl = g->cl;
  const int64_t bx = (g->S + kMergeThreads * kMergeSPT - 1) / (kMergeThreads * kMergeSPT);
  // enough workgroups to fill the chip, each walking a contiguous run of frames
  int64_t by = std::max<int64_t>(1, std::min<int64_t>((F + kMergeFrames - 1) / kMergeFrames, (1024 * (1024 / kMergeThreads)) / bx));
  int fpb = (int)((F + by - 1) / by);
  fpb = (fpb + kMergeFrames - 1) / kMergeFrames * kMergeFrames;
  by = (F + fpb - 1) / fpb;
  const int cstride = cl.C * 12 * 4 <= 144 * 1024 ? 12 : kMergeFrames;
  const int smem = cstride * cl.C * (int)sizeof(float);
  static bool attr_set[64] = {false};
  if (!attr_set[g->device & 63]) {
    AASR_HIP(hipFuncSetAttribute((const void *)k_cluster_merge<NNZ, kMergeThreads, DST>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    attr_set[g->device & 63] = true;
  }
  hipLaunchKernelGGL((k_cluster_merge<NNZ, kMergeThreads, DST>), dim3((unsigned)bx, (unsigned)by), dim3(kMergeThreads), smem,
                     stream, d_out, F, g->S, cl.cval.p, cl.C, cl.w_cluster.p, cl.w_weight.p, cl.nnz,
                     (float)cl.ref_log2, fpb, cstride, pitch, colmap, dst, dst_pitch);
  AASR_HIP(hipGetLastError());
}

template <int NNZ, int kMergeThreads>
static void launch_merge_t(aasr_gmm *g, float *d_out, int64_t F, int64_t pitch, hipStream_t stream, const int32_t *colmap,
                           float *dst, int64_t dst_pitch) {
  if (dst) launch_merge_tt<NNZ, kMergeThreads, true>(g, d_out, F, pitch, stream, colmap, dst, dst_pitch);
  else launch_merge_tt<NNZ, kMergeThreads, false>(g, d_out, F, pitch, stream, colmap, dst, dst_pitch);
}

static void launch_merge(aasr_gmm *g, float *d_out, int64_t F, int64_t pitch, hipStream_t stream, const int32_t *colmap,
                         float *dst, int64_t dst_pitch) {
  ClusterState &cl = g->cl;
  if (cl.log_merge) {
    hipLaunchKernelGGL(k_cluster_merge_log, dim3((unsigned)((g->S + 255) / 256), (unsigned)std::min<int64_t>(F, 8192)),
                       dim3(256), 0, stream, d_out, F, g->S, cl.cval.p, cl.C, cl.w_cluster.p, cl.w_weight.p, cl.nnz, pitch, colmap);
    AASR_HIP(hipGetLastError());
    return;
  }
  static const int threads = AASR_EXPERIMENT_ENV("AASR_MERGE_THREADS") ? atoi(AASR_EXPERIMENT_ENV("AASR_MERGE_THREADS")) : 1024;
  if (cl.nnz <= 8) launch_merge_t<8, 1024>(g, d_out, F, pitch, stream, colmap, dst, dst_pitch);
  else if (threads == 512) launch_merge_t<16, 512>(g, d_out, F, pitch, stream, colmap, dst, dst_pitch);
  else if (threads == 256) launch_merge_t<16, 256>(g, d_out, F, pitch, stream, colmap, dst, dst_pitch);
  else launch_merge_t<16, 1024>(g, d_out, F, pitch, stream, colmap, dst, dst_pitch);   // weights beyond 16 per state come from L2
}

// What a model's exact part needs besides the selection bits: the cluster of each of ITS rows /
// records (cl.crow*, built on first use from cl.g2c) and the expanded lane masks (cl.maskrow).
// Models the expanded form cannot hold (gmm.h, KAPPA_LIMIT): the exact values of the ill-conditioned
// Gaussians -- a minority next to the masked track kernel (outlier routing), or the whole model --
// come from the centred kernel under the same bits.
struct ExactPlan {
  bool dimsplit = false;   // feature dimension > 63: the dimension parts per Gaussian, the masks applied where they are added
  bool full = false;   // full-covariance pool: the factor-row kernel with masks
  bool all_centred, with_outliers;
  int which;
  int64_t mask_rows;
};

static ExactPlan exact_part_plan(aasr_gmm *g, ClusterState &cl) {
  ExactPlan p;
  if (!g->dim_parts.empty()) {
    p.dimsplit = true;
    p.all_centred = p.with_outliers = false;
    p.which = 0;
    p.mask_rows = 0;
    if (cl.crow_gauss.n != (size_t)g->G) {
      std::vector<int32_t> cg((size_t)g->G);
      for (int64_t i = 0; i < g->G; i++) cg[(size_t)i] = cl.g2c[(size_t)i] >= 0 ? cl.g2c[(size_t)i] : cl.C;
      cl.crow_gauss.upload(cg.data(), cg.size());
    }
    return p;
  }
  if (g->host.any_full()) {
    // the reference's cluster branch does not look at the type of the pool's Gaussians (aku/Distributions.cc:
    // 2684-2722): diagonal centres, exact members -- here the members' factor rows with selection masks
    if (!g->full.ok) raise(AASR_ERR_UNSUPPORTED, "full-covariance layout was not built for this model");
    p.full = true;
    p.all_centred = p.with_outliers = false;
    p.which = 0;
    if (cl.crow_full.n != g->full.row_gauss.size()) {
      std::vector<int32_t> crow(g->full.row_gauss.size(), cl.C);
      for (size_t r = 0; r < crow.size(); r++) {
        const int32_t gi = g->full.row_gauss[r];
        if (gi >= 0 && cl.g2c[(size_t)gi] >= 0) crow[r] = cl.g2c[(size_t)gi];
      }
      cl.crow_full.upload(crow.data(), crow.size());
    }
    p.mask_rows = g->full.rows_padded;
    return p;
  }
  p.all_centred = g->ill_conditioned || (!g->paired.ok && !g->tracks.ok);
  p.with_outliers = g->hyb_enabled && !p.all_centred;
  if (p.all_centred && !g->centred_ok)
    raise(AASR_ERR_UNSUPPORTED, "no centred kernel instance for dimension %d", g->dim);
  // the layout the exact part runs on: grouped unless it is missing or masked out
  p.which = (g->paired.ok && ((g->layout_mask & 1) || !g->tracks.ok)) ? 0 : 1;
  const TrackLayout &L = p.which == 0 ? g->paired : g->tracks;
  if (!p.all_centred && !L.ok) raise(AASR_ERR_UNSUPPORTED, "Gaussian clustering needs a track layout for this model");
  if (!p.all_centred && cl.crow[p.which].n != L.row_gauss.size()) build_crow(g, L, cl, cl.crow[p.which]);
  p.mask_rows = p.all_centred ? 0 : L.rows_padded;  // the centred kernel reads the cluster bits themselves
  if (p.with_outliers && cl.crow_hyb.n != std::max<size_t>(g->hyb_comps.size(), 1))
    build_crow_comps(g, g->hyb_comps, cl, cl.crow_hyb);
  if (p.all_centred && cl.crow_centred.n != std::max<size_t>(g->host.mix_idx.size(), 1))
    build_crow_comps(g, std::vector<int32_t>(), cl, cl.crow_centred);
  return p;
}

// tiles a workgroup of k_cluster_expand takes: its staged copy of the word's cluster masks (8 bytes per cluster) serves
// all of them, so as many as leave the grid a few thousand workgroups (AASR_EXPAND_TPB_MIN: 64 until round 5 for any size)
static int expand_tiles_per_block(int64_t n_tiles, int64_t words) {
#ifndef AASR_EXPAND_ALL_TILES
#define AASR_EXPAND_ALL_TILES 1
#endif
  if (!AASR_EXPAND_ALL_TILES) return (int)std::min<int64_t>(n_tiles, 64);
  const int64_t want_x = std::max<int64_t>(1, (4096 + words - 1) / words);   // workgroups along the tiles
  const int64_t tpb = (n_tiles + want_x - 1) / want_x;
  return (int)std::max<int64_t>(std::min<int64_t>(n_tiles, 16), tpb);
}

static void launch_expand(aasr_gmm *g, const unsigned long long *maskw, int c1, const int32_t *crow, int64_t rows_padded,
                          int64_t n_tiles, int tpb, int64_t words, unsigned long long *maskrow, hipStream_t stream) {
  static bool attr_set[64] = {false};
  if ((size_t)c1 * 8 > 48 * 1024 && !attr_set[g->device & 63]) {
    AASR_HIP(hipFuncSetAttribute((const void *)k_cluster_expand, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[g->device & 63] = true;
  }
  hipLaunchKernelGGL(k_cluster_expand, dim3((unsigned)((n_tiles + tpb - 1) / tpb), (unsigned)words), dim3(256),
                     (size_t)c1 * 8, stream, maskw, c1, crow, rows_padded, tpb, maskrow);
  AASR_HIP(hipGetLastError());
}

// out[f][s] = log of the sum over s's components whose cluster is evaluated exactly for frame f (no floor)
static void exact_part_launch(aasr_gmm *g, ClusterState &cl, const ExactPlan &p, const unsigned long long *maskw,
                              int c1, int64_t n, const float *fr_members, float *out, hipStream_t stream, int64_t pitch = 0) {
  const int64_t words = (n + 63) / 64;
  if (p.dimsplit) {
    // the members' frames arrive adapted where the pool has one transform (gmm_cluster_score_launch)
    gmm_dim_split_score(g, fr_members, n, out, false, stream, maskw, c1, cl.crow_gauss.p, g->xf_a.p != nullptr);
    return;
  }
  if (p.full) {
    const int64_t rows_padded = g->full.rows_padded;
    cl.maskrow.ensure((size_t)((n + 511) / 512 * 8) * (size_t)rows_padded);
    const int64_t n_tiles = rows_padded / TILE_ROWS;
    const int tpb = expand_tiles_per_block(n_tiles, words);
    launch_expand(g, maskw, c1, cl.crow_full.p, rows_padded, n_tiles, tpb, words, cl.maskrow.p, stream);
    gmm_full_masked_launch(g, fr_members, n, out, cl.maskrow.p, stream);
    return;
  }
  if (p.all_centred) {
    gmm_centred_masked_launch(g, fr_members, n, out, cl.crow_centred.p, maskw, c1, words, stream);
    // log|det| of an in-place global transform: the centred records do not carry it (the track kernels take it at
    // their output, the outlier merge adds it to the centred share)
    if (g->out_bias_ln != 0) gmm_add_bias_nofloor(out, n * g->S, (float)g->out_bias_ln, stream);
    return;
  }
  const TrackLayout &L = p.which == 0 ? g->paired : g->tracks;
  // the track kernels read the lane masks of whole workgroups (up to 512 frames = 8 words)
  cl.maskrow.ensure((size_t)((n + 511) / 512 * 8) * (size_t)L.rows_padded);
  const int64_t n_tiles = L.rows_padded / TILE_ROWS;
  const int tpb = expand_tiles_per_block(n_tiles, words);
  launch_expand(g, maskw, c1, cl.crow[p.which].p, L.rows_padded, n_tiles, tpb, words, cl.maskrow.p, stream);
  gmm_tracks_masked_launch(g, p.which, fr_members, n, out, cl.maskrow.p, stream, pitch);
  if (p.with_outliers) gmm_outliers_masked_launch(g, fr_members, n, out, cl.crow_hyb.p, maskw, c1, words, stream);
}

// A row pitch (state rows padded to whole cache lines, gmm_score_pitch_ok) is carried by the plain plan only: one
// masked track kernel and the merge.  Class sub-models, outlier routing, the centred and full-covariance kernels write
// dense rows.
// ... whatever the clustering state of the handle itself (an engine part's own exact part)
static bool gmm_cluster_plan_pitch_ok(const aasr_gmm *g) {
  return !g->class_routing && !g->host.any_full() && !g->host.factor_path() && !g->hyb_enabled && !g->ill_conditioned &&
         g->dim_parts.empty() && (g->paired.ok || g->tracks.ok);
}
bool gmm_cluster_pitch_ok(const aasr_gmm *g) {
  if (g->cl.enabled && gmm_engine_parts_clustered(g)) return true;   // gathered into any pitch
  return g->cl.enabled && !g->class_routing && !g->host.any_full() && !g->host.factor_path() && !g->hyb_enabled &&
         !g->ill_conditioned && (g->paired.ok || g->tracks.ok);
}

void gmm_cluster_score_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                              hipStream_t stream, int64_t pitch) {
  ClusterState &cl = g->cl;
  if (pitch <= 0) pitch = g->S;
  if (pitch != g->S && !gmm_cluster_pitch_ok(g))
    raise(AASR_ERR_UNSUPPORTED, "a row pitch under Gaussian clustering needs the plain track plan");
  if (!g->host.any_full() && g->host.factor_path() && !g->class_routing)
    raise(AASR_ERR_UNSUPPORTED, "Gaussian clustering over per-class transforms needs the class sub-models");
  if (g->host.any_full() && g->host.n_transforms > 0)
    raise(AASR_ERR_UNSUPPORTED, "Gaussian clustering over an adapted full-covariance pool is not built");
  // One global constrained-MLLR transform: the pool's Gaussians are AdaptedGaussians -- members are
  // evaluated on A f + b and scaled by |det| (the track kernels' output bias) -- while the cluster
  // centres are plain Gaussians on the frame itself (aku/ModelModules.hh:164-173,
  // aku/Distributions.cc:2688-2691): the centre kernel gets the frames, the masked scoring kernel
  // the adapted ones.  Per-class transforms (class routing, gmm.h): every class's sub-model gives
  // its exact part on its own adapted frames, the parts are added with their log|det|, then the
  // centres' share of ALL components enters in the merge.
  const bool classes = g->class_routing;
  const float *d_members = (!classes && g->xf_a.p) ? gmm_adapted_frames(g, d_frames, F, stream) : d_frames;
  ExactPlan plan{};
  std::vector<ExactPlan> sub_plans;
  int64_t mask_rows = 0;
  // Engine parts (gmm_plan_engine_parts: a model whose conditioning needs several pivots): every part gives the exact
  // part of ITS states under the same selection bits -- its own rows, masks, arithmetic -- into its column range of an
  // engine score row; the merge adds the centres' share through the column map, the columns are gathered back.  What a
  // clustered recogniser run (pyrectool: always) costs on a model fitted to data then stays next to the BASELINE model's
  // instead of falling to the centred form for every Gaussian.
  const bool parts = gmm_engine_parts_clustered(g);
  if (parts) {
    sub_plans.resize(g->engine_parts.size());
    for (size_t c = 0; c < g->engine_parts.size(); c++) {
      aasr_gmm *sub = g->engine_parts[c].model.get();
      if (!sub->cl.loaded || sub->cl.C != cl.C || sub->cl.g2c.size() != (size_t)sub->G) {
        sub->cl = ClusterState();
        sub->cl.C = cl.C;
        sub->cl.g2c.assign((size_t)sub->G, -1);
        for (int64_t i = 0; i < sub->G && i < (int64_t)sub->parent_gauss.size(); i++)
          sub->cl.g2c[(size_t)i] = cl.g2c[(size_t)sub->parent_gauss[(size_t)i]];
        sub->cl.loaded = true;
      }
      if (g->engine_parts[c].arith == 0) {
        sub->precision = g->precision;
        sub->use_bf16x3 = g->use_bf16x3;
      }
      sub->out_bias_ln = g->out_bias_ln;
      sub_plans[c] = exact_part_plan(sub, sub->cl);
      mask_rows = std::max(mask_rows, sub_plans[c].mask_rows);
    }
  } else if (classes) {
    sub_plans.resize(g->class_models.size());
    for (size_t c = 0; c < g->class_models.size(); c++) {
      aasr_gmm *sub = g->class_models[c].get();
      if (!sub) continue;
      if (!sub->cl.loaded || sub->cl.C != cl.C || sub->cl.g2c.size() != (size_t)sub->G) {
        // the parent's clustering seen from the class's own pool
        sub->cl = ClusterState();
        sub->cl.C = cl.C;
        sub->cl.g2c.resize((size_t)sub->G);
        for (int64_t i = 0; i < sub->G; i++) sub->cl.g2c[(size_t)i] = cl.g2c[(size_t)sub->parent_gauss[(size_t)i]];
        sub->cl.loaded = true;
      }
      sub->precision = g->precision;
      sub->use_bf16x3 = g->use_bf16x3;
      sub->layout_mask = g->layout_mask;
      sub_plans[c] = exact_part_plan(sub, sub->cl);
      mask_rows = std::max(mask_rows, sub_plans[c].mask_rows);
    }
  } else {
    // (a global transform over a model with centred-kernel Gaussians carries |det| in the component
    // weights -- HostModel::logw_bias -- so rows and centred records already hold it)
    plan = exact_part_plan(g, cl);
    mask_rows = plan.mask_rows;
  }
  // Frames per pass.  The track kernel and the merge run once per pass, so a pass is
  // as large as ~24 GB of scratch allow (1 bit per frame x packed row for the lane
  // masks, 4 B per frame x cluster for the centre values) and a whole number of rounds
  // of the track kernel (2 workgroups of 256 frames per CU); the f64 centre
  // log-likelihoods (8 B per frame x cluster) only live for a sub-pass of <= 8.6 GB (10^6 frames x 1000 clusters in one go: 42.7 ms per pass instead of 45.0 with five sub-passes of 2 GB, whose launches each end in a partly filled round of the one-wave selection workgroups; smaller sub-passes are worse still: 512 MB 52 ms, 256 MB 61 ms).
  const int64_t ep = parts ? gmm_engine_pitch_max(g) : 0;
  const double per_frame = (double)mask_rows / 8.0 + 4.125 * (double)(cl.C + 1) + 4.0 + (classes ? 4.0 * (double)g->S : 0.0) +
                           (parts ? 8.0 * (double)ep : 0.0);
  const int64_t round_frames = 2 * (int64_t)(g->num_cus > 0 ? g->num_cus : 256) * FRAMES_PER_BLOCK;
  const int64_t f_rounded = (F + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK * FRAMES_PER_BLOCK;
  int64_t fb = (int64_t)((g_pass_bytes > 0 ? g_pass_bytes : 24.0e9) / per_frame);
  if (fb >= round_frames) fb = fb / round_frames * round_frames;
  else fb = std::max<int64_t>(FRAMES_PER_BLOCK, fb / FRAMES_PER_BLOCK * FRAMES_PER_BLOCK);
  fb = std::min<int64_t>(fb, f_rounded);
  if (fb < f_rounded) {
    // several passes: of equal size (a short last pass pays the fixed costs of every kernel -- the one-wave selection
    // workgroups of 56 000 left-over frames took as long as those of 393 000)
    const int64_t n_pass = (f_rounded + fb - 1) / fb;
    fb = ((f_rounded + n_pass - 1) / n_pass + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK * FRAMES_PER_BLOCK;
  }
  const int64_t fb_pass = fb;
  static const double sub_bytes = AASR_EXPERIMENT_ENV("AASR_CLUSTER_SUB_BYTES") ? atof(AASR_EXPERIMENT_ENV("AASR_CLUSTER_SUB_BYTES")) : 8.6e9;
  int64_t fs = (int64_t)(sub_bytes / (8.0 * (double)cl.Cs));
  fs = std::min<int64_t>(fb, std::max<int64_t>(FRAMES_PER_BLOCK, fs / FRAMES_PER_BLOCK * FRAMES_PER_BLOCK));
  if (fb > cl.Fc || (size_t)fs * cl.Cs > cl.ll64.n) {
    fb = std::max(fb, cl.Fc);
    cl.ll64.alloc((size_t)fs * cl.Cs);
    if (cluster_fast_keys(g)) {
      cl.key32.alloc((size_t)fs * cl.Cs);
      cl.pend_list.alloc((size_t)fs + 1);
    }
    cl.cval.alloc((size_t)fb * cl.C);
    cl.maskw.alloc((size_t)(fb / 64) * (cl.C + 1));
    cl.n_exact.alloc((size_t)fb);
    cl.tie_list.alloc((size_t)fs + 1);
    cl.heap_key.ensure((size_t)cl.C * kHeapThreads);
    cl.heap_idx.ensure((size_t)cl.C * kHeapThreads);
    cl.Fc = fb;
    cl.Fs = fs;
  }
  const int64_t pass_frames = std::min<int64_t>(fb_pass, cl.Fc);   // (this call's balanced passes inside the allocation)
  g_last_passes = (int)((F + pass_frames - 1) / pass_frames);
  for (int64_t f0 = 0; f0 < F; f0 += pass_frames) {
    const int64_t n = std::min<int64_t>(pass_frames, F - f0);
    const float *fr = d_frames + f0 * g->dim;
    const float *fr_members = d_members + f0 * g->dim;
    float *out = d_out + f0 * pitch;
    for (int64_t s0 = 0; s0 < n; s0 += cl.Fs) {
      const int64_t ns = std::min<int64_t>(cl.Fs, n - s0);
      if (cluster_fast_keys(g) && cl.key32.n >= (size_t)ns * cl.Cs) {
        launch_centres_mfma(g, fr + s0 * g->dim, ns, stream, 1);
        launch_select(g, s0, ns, stream, fr + s0 * g->dim);
      } else {
        launch_centres(g, fr + s0 * g->dim, ns, stream);
        launch_select(g, s0, ns, stream);
      }
    }
    if (parts) {
      if ((size_t)n * ep > g->engine_scratch.n) {
        AASR_HIP(hipDeviceSynchronize());
        g->engine_scratch.ensure((size_t)cl.Fc * ep);
      }
      for (size_t c = 0; c < g->engine_parts.size(); c++) {
        auto &part = g->engine_parts[c];
        aasr_gmm *sub = part.model.get();
        float *o = g->engine_scratch.p + part.col0;
        if (gmm_cluster_plan_pitch_ok(sub)) {
          exact_part_launch(sub, sub->cl, sub_plans[c], cl.maskw.p, cl.C + 1, n, fr_members, o, stream, ep);
        } else {   // (the remainder's centred / outlier kernels write dense rows)
          if ((size_t)n * sub->S > g->engine_part_scratch.n) {
            AASR_HIP(hipDeviceSynchronize());
            g->engine_part_scratch.ensure((size_t)cl.Fc * sub->S);
          }
          exact_part_launch(sub, sub->cl, sub_plans[c], cl.maskw.p, cl.C + 1, n, fr_members, g->engine_part_scratch.p, stream);
          gmm_scatter_columns(g->engine_part_scratch.p, n, sub->S, o, ep, stream);
        }
      }
      if (cl.log_merge) {   // (the fallback merges in place)
        launch_merge(g, g->engine_scratch.p, n, ep, stream, g->engine_colmap.p);
        gmm_gather_engine_columns(g, g->engine_scratch.p, n, ep, out, pitch, stream);
      } else {
        launch_merge(g, g->engine_scratch.p, n, ep, stream, g->engine_colmap.p, out, pitch);
      }
      continue;
    }
    if (classes)
      gmm_classes_exact_launch(g, fr, n, out, [&](aasr_gmm *sub, size_t c, const float *xf, float *part) {
        exact_part_launch(sub, sub->cl, sub_plans[c], cl.maskw.p, cl.C + 1, n, xf, part, stream);
      }, stream);
    else
      exact_part_launch(g, cl, plan, cl.maskw.p, cl.C + 1, n, fr_members, out, stream, pitch);
    launch_merge(g, out, n, pitch, stream);
  }
}

// AASR_PREC_F64 with Gaussian clustering: the reference's cluster branch in double end to end --
// centres on the double frames, the same selection kernels, then the f64 scoring kernel taking for
// every component either its exact likelihood (x |det| under a global transform) or its cluster
// centre's (PDFPool::precompute_likelihoods, aku/Distributions.cc:2684-2722).  Per sub-pass, while
// the centre keys live.
void gmm_cluster_score_f64_launch(aasr_gmm *g, const double *d_frames, const double *d_members, int64_t F,
                                  double *d_out, int linear, double det, hipStream_t stream) {
  ClusterState &cl = g->cl;
  if (cl.crow_centred.n != std::max<size_t>(g->host.mix_idx.size(), 1))
    build_crow_comps(g, std::vector<int32_t>(), cl, cl.crow_centred);
  const int64_t f_rounded = (F + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK * FRAMES_PER_BLOCK;
  static const double sub_bytes = AASR_EXPERIMENT_ENV("AASR_CLUSTER_SUB_BYTES") ? atof(AASR_EXPERIMENT_ENV("AASR_CLUSTER_SUB_BYTES")) : 8.6e9;
  int64_t fs = (int64_t)(sub_bytes / (8.0 * (double)cl.Cs));
  fs = std::min<int64_t>(f_rounded, std::max<int64_t>(FRAMES_PER_BLOCK, fs / FRAMES_PER_BLOCK * FRAMES_PER_BLOCK));
  if (fs > cl.Fc || (size_t)fs * cl.Cs > cl.ll64.n) {  // pass == sub-pass here
    const int64_t fb = std::max(fs, cl.Fc);
    cl.ll64.alloc((size_t)fb * cl.Cs);
    cl.cval.alloc((size_t)fb * cl.C);
    cl.maskw.alloc((size_t)(fb / 64) * (cl.C + 1));
    cl.n_exact.alloc((size_t)fb);
    cl.tie_list.alloc((size_t)fb + 1);
    cl.heap_key.ensure((size_t)cl.C * kHeapThreads);
    cl.heap_idx.ensure((size_t)cl.C * kHeapThreads);
    cl.Fc = fb;
    cl.Fs = fb;
  }
  for (int64_t f0 = 0; f0 < F; f0 += cl.Fs) {
    const int64_t n = std::min<int64_t>(cl.Fs, F - f0);
    launch_centres(g, d_frames + f0 * g->dim, n, stream);
    launch_select(g, 0, n, stream);
    if (g->f64_classes > 0)  // regression classes: members on their class's frames (made from the raw ones)
      gmm_f64_classes_masked_launch(g, d_frames + f0 * g->dim, n, d_out + f0 * g->S, linear, cl.crow_centred.p,
                                    cl.maskw.p, cl.C + 1, cl.ll64.p, (int64_t)cl.Cs, cl.C, stream);
    else
      gmm_f64_masked_launch(g, d_members + f0 * g->dim, n, d_out + f0 * g->S, linear, det, cl.crow_centred.p,
                            cl.maskw.p, cl.C + 1, cl.ll64.p, (int64_t)cl.Cs, cl.C, stream);
  }
}

}  // namespace aasr

// Diagnostic (not part of the public ABI): clusters evaluated exactly for each of
// the first n frames of the most recent clustered scoring pass.
// Diagnostic: 1 = every frame's cluster selection goes through the priority-queue replay
// (k_cluster_select_heap) instead of the histogram selection; the two must agree.
extern "C" void aasr_debug_cluster_heap(int on) { aasr::g_force_heap = on; }
// Diagnostic: bytes of scratch a clustered pass may use (0: the default, 24 GB) -- small values force several passes.
extern "C" void aasr_debug_cluster_pass_bytes(double bytes) { aasr::g_pass_bytes = bytes; }
extern "C" int aasr_debug_cluster_last_passes(void) { return aasr::g_last_passes; }
// Diagnostic: frames of the last sub-pass that were handed to the replay.
extern "C" int aasr_debug_cluster_tie_frames(aasr_gmm *g) {
  if (!g || !g->cl.tie_list.p) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  int32_t n = 0;
  if (hipMemcpy(&n, g->cl.tie_list.p + g->cl.Fs, sizeof n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}

extern "C" int aasr_debug_cluster_exact_counts(aasr_gmm *g, int32_t *out, int64_t n) {
  if (!g || !g->cl.n_exact.p || n > g->cl.Fc) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpy(out, g->cl.n_exact.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost) ==
                 hipSuccess
             ? 0
             : -1;
}
