// gmm_score.hip -- frame x Gaussian scoring kernels for gfx950 (MI355X).
//
// Replaces the per-frame loop
//   HmmSet::precompute_likelihoods        aku/HmmSet.cc:484-501
//     PDFPool::precompute_likelihoods     aku/Distributions.cc:2663-2682
//       DiagonalGaussian::compute_log_likelihood   :1040-1062  (+ exp :1033)
//     Mixture::compute_likelihood         aku/Distributions.cc:2078-2086
// with one launch over a block of frames.
//
// Formulation.  ll_g(x) = c_g - 1/2 sum_d p_gd (x_d - mu_gd)^2 is expanded
// around a per-dimension pivot v (x' = x - v, mu' = mu - v) into a dense
// contraction over K = 2*dim + 1:
//     log2e * ll = sum_d [p mu' log2e] x'_d + [-p/2 log2e] x'_d^2 + C_g * 1
// so the frame x Gaussian quadratic forms are an exact-f32 GEMM that runs on
// the matrix cores (v_mfma_f32_32x32x2_f32: one instruction = one dimension's
// (x', x'^2) pair), with the mixture log-sum-exp fused behind it.  log(w) of
// the mixture weight is folded into C_g, the result is in log2 units so the
// epilogue is max -> v_exp_f32(x - max) -> add -> v_log_f32.
//
// Work decomposition (frame-stationary).  A workgroup of 4 waves owns 256
// frames; each wave keeps its 64 frames' K x 64 operand in VGPRs for the whole
// kernel (nkk x 2 registers) and the Gaussian rows stream past in tiles of 64
// rows: HBM/L2 -> LDS by global_load_lds (double buffered, one barrier per
// tile), LDS -> A fragments by ds_read_b128.  Each wave computes a 64-row x
// 64-frame tile as 2x2 MFMA blocks.  The epilogue dumps one 32-row block pair
// at a time to a wave-private LDS staging area laid out [row][frame], then
// every lane owns one frame and reduces the rows of each mixture segment
// (16 rows at a time in registers, merged online) -- no cross-lane traffic,
// arbitrary components per state, segments may span chunks and tiles.
//
// Roofline: FP32 matrix rate, 2*K flop per frame x row pair; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>

#include <cmath>
#include <cstdlib>

#include "gmm.h"

namespace aasr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define LN2_F 0.69314718055994530942f
#define LOG2E_F 1.4426950408889634074f
// log(1e-50): HmmSet clamps state likelihoods at util::tiny_for_log
// (aku/HmmSet.cc:497-498, aku/util.hh:131)
#define LOG_TINY_F (-115.12925464970228f)
#define NEG_BIG_F (-3.0e38f)
// finished state log-likelihoods are buffered per wave and written OUT_GROUP
// consecutive states at a time: 32 contiguous bytes per frame row
// Kernel ablations (AASR_DBG=bits: 1 matrix stream only, 2 no A-fragment reads, 16 no barriers,
// 256 / 512 no / half of the tile copies, 4 / 8 scheduling experiments) exist only in a build made
// with AASR_BUILD_ABLATION=1 (-DAASR_ABLATION=1); the product kernels carry none of the branches.
#ifndef AASR_ABLATION
#define AASR_ABLATION 0
#endif
#if AASR_ABLATION
#define AASR_DBG(bits) (dbg & (bits))
#else
#define AASR_DBG(bits) false
#endif

constexpr int OUT_GROUP = 8;

template <int NKK>
struct ScoreSmem {
  // [2 buffers][NKK/2][64 lanes][4] floats
  static constexpr int kTileFloats = (NKK / 2) * 64 * 4;
  static constexpr int kStageFloatsPerWave = CHUNK_ROWS * FRAMES_PER_WAVE;
  // per-wave output transposition buffer: OUT_GROUP finished states x 64 frames
  static constexpr int kOutFloatsPerWave = OUT_GROUP * FRAMES_PER_WAVE;
  static constexpr int kBytes =
      (2 * kTileFloats + WAVES_PER_BLOCK * (kStageFloatsPerWave + kOutFloatsPerWave)) * 4;
};

__device__ __forceinline__ void issue_tile_copy(const float *__restrict__ gtile,
                                                float *lds_buf, int tile_floats,
                                                int wave, int lane) {
  // 16 bytes per lane per issue; the LDS destination of a global_load_lds is
  // wave-uniform base + lane*16, i.e. lane-linear -- exactly the packed layout.
  const int chunks = tile_floats / 4;  // 16-byte pieces
  for (int c0 = wave * 64; c0 < chunks; c0 += WAVES_PER_BLOCK * 64) {
    const float *src = gtile + (size_t)(c0 + lane) * 4;
    float *dst = lds_buf + (size_t)c0 * 4;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)src,
        (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  }
}

// The same copy issued through inline assembly, i.e. invisible to the compiler's wait-count
// bookkeeping.  The builtin is modelled as a FLAT access that touches LDS and global memory at
// once; while one is outstanding the compiler degrades EVERY s_waitcnt lgkmcnt(n) to
// lgkmcnt(0), which serialises the A-fragment prefetch of the matrix stream against LDS latency.
// Callers must order the copy themselves: s_waitcnt vmcnt(0) + barrier before the tile is read.
__device__ __forceinline__ void issue_tile_copy_raw(const float *__restrict__ gtile, float *lds_buf,
                                                    int tile_floats, int wave, int lane,
                                                    int nwaves = WAVES_PER_BLOCK) {
  const int chunks = tile_floats / 4;
  for (int c0 = wave * 64; c0 < chunks; c0 += nwaves * 64) {
    const float *src = gtile + (size_t)(c0 + lane) * 4;
    // the LDS offset is the low half of the generic address (the aperture sits in the high half): no
    // addrspacecast, whose null check the compiler mis-selects in some instantiations
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_buf + (size_t)c0 * 4));
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(dst), "v"(src) : "memory", "m0");
  }
}

// Per-tile close bits through the scalar cache (SMEM, lgkmcnt).  A vector load here is a trap:
// its result is needed as a scalar, the compiler waits for it with s_waitcnt vmcnt(0), and
// vmcnt counts in issue order -- so the wave would sit until the tile copy issued just before
// it has landed (measured: 5 ms of 30 in the bf16x3 matrix stream).  t is wave-uniform.
typedef const __attribute__((address_space(4))) uint32_t *cmask32_ptr;
__device__ __forceinline__ unsigned sload_close_pair(const uint16_t *close_mask, int64_t t) {
  return ((cmask32_ptr)close_mask)[__builtin_amdgcn_readfirstlane((int)(t >> 1))];
}
// Bits of tile t out of a word requested earlier; the empty asm keeps the compiler from doing the
// extraction (and therefore the lgkmcnt wait) right behind the request.
__device__ __forceinline__ unsigned close16_of_pair(unsigned pair, int64_t t) {
  asm volatile("" : "+s"(pair));
  return (t & 1) ? pair >> 16 : pair & 0xffffu;
}
__device__ __forceinline__ unsigned sload_close16(const uint16_t *close_mask, int64_t t) {
  return close16_of_pair(sload_close_pair(close_mask, t), t);
}
__device__ __forceinline__ unsigned sload_close32(const uint32_t *close_mask, int64_t t) {
  return ((cmask32_ptr)close_mask)[__builtin_amdgcn_readfirstlane((int)t)];
}

// Reduce rows [a, b) of the staged chunk for this lane's frame, 16 rows at a
// time, merging into the running (m, s) pair:  sum_r 2^v_r = s * 2^m.
__device__ __forceinline__ void reduce_rows(const float *stage_col, int a, int b,
                                            float &m, float &s) {
  for (int r0 = a; r0 < b; r0 += 16) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      int r = r0 + i;
      int rc = r < b ? r : b - 1;
      float x = stage_col[rc * FRAMES_PER_WAVE];
      v[i] = r < b ? x : NEG_BIG_F;
    }
    float gm = v[0];
#pragma unroll
    for (int i = 1; i < 16; i++) gm = fmaxf(gm, v[i]);
    float mn = fmaxf(m, gm);
    float acc = s * __builtin_amdgcn_exp2f(m - mn);
#pragma unroll
    for (int i = 0; i < 16; i++) acc += __builtin_amdgcn_exp2f(v[i] - mn);
    m = mn;
    s = acc;
  }
}

// MODE 0: per-state mixture log-likelihoods (segmented log-sum-exp)
// MODE 1: raw per-row log-likelihoods (pool view), out[f][row]
template <int NKK, int MODE>
__global__ __launch_bounds__(256, 2) void k_gmm_diag_score(
    const float *__restrict__ frames, int64_t F, int dim,
    const float *__restrict__ pivot, const float *__restrict__ apack,
    int64_t tiles, const int32_t *__restrict__ chunk_seg_begin,
    const uint32_t *__restrict__ seg_desc, const int32_t *__restrict__ seg_out,
    float *__restrict__ out, int64_t out_cols, int64_t rows, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *smem = (float *)smem_raw;
  constexpr int kTileFloats = ScoreSmem<NKK>::kTileFloats;
  float *abuf0 = smem;
  float *abuf1 = smem + kTileFloats;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  float *stage = smem + 2 * kTileFloats + wave * ScoreSmem<NKK>::kStageFloatsPerWave;

  const int n = lane & 31;   // MFMA column (frame within a 32-block)
  const int h = lane >> 5;   // K parity held by this lane
  const int64_t f0 = (int64_t)blockIdx.x * FRAMES_PER_BLOCK + wave * FRAMES_PER_WAVE;

  // ---- frame operand: B[kk][nb] = h ? x'^2 : x'  (kk<dim), 1 at kk==dim/h==0
  float bf[NKK][2];
#pragma unroll
  for (int nb = 0; nb < 2; nb++) {
    int64_t f = f0 + nb * 32 + n;
    if (f > F - 1) f = F - 1;
    const float *xr = frames + f * dim;
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) {
      const int kc = kk < dim ? kk : 0;
      const float xc = xr[kc] - pivot[kc];
      float v = h ? xc * xc : xc;
      if (kk == dim) v = h ? 0.0f : 1.0f;
      if (kk > dim) v = 0.0f;
      bf[kk][nb] = v;
    }
  }

  float carry_m = NEG_BIG_F, carry_s = 0.0f;
  float *ost = smem + 2 * kTileFloats + WAVES_PER_BLOCK * ScoreSmem<NKK>::kStageFloatsPerWave +
               wave * ScoreSmem<NKK>::kOutFloatsPerWave;
  int n_closed = 0;  // states finished so far == index of the next state (MODE 0)

  // prologue: tile 0 -> buffer 0
  issue_tile_copy(apack, abuf0, kTileFloats, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int64_t my_frame = f0 + lane;  // epilogue: lane <-> frame
  const bool frame_ok = my_frame < F;

  for (int64_t t = 0; t < tiles; t++) {
    float *acur = (t & 1) ? abuf1 : abuf0;
    float *anext = (t & 1) ? abuf0 : abuf1;
    // Buffer `anext` was last read by the MFMA loop of tile t-1; every wave is
    // past the barrier that followed that loop, so it can be refilled now.
    if (t + 1 < tiles)
      issue_tile_copy(apack + (size_t)(t + 1) * kTileFloats, anext, kTileFloats, wave, lane);

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    const f32x4 *afrag = (const f32x4 *)acur + lane;
    // A fragments are fetched two kk-pairs ahead of their MFMAs
    f32x4 a0 = afrag[0];
    f32x4 a1 = afrag[(NKK / 2 > 1 ? 1 : 0) * 64];
#pragma unroll
    for (int q = 0; q < NKK / 2; q++) {
      const int qn = (q + 2 < NKK / 2) ? q + 2 : NKK / 2 - 1;
      f32x4 a2 = afrag[qn * 64];
      const f32x4 av = a0;  // {mb0 kk0, mb0 kk1, mb1 kk0, mb1 kk1}
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[2 * q][0], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[2 * q][1], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[2 * q][0], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[2 * q][1], c11, 0, 0, 0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[2 * q + 1][0], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[2 * q + 1][1], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[2 * q + 1][0], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[2 * q + 1][1], c11, 0, 0, 0);
      a0 = a1;
      a1 = a2;
    }

    // One barrier per tile, here: (a) every wave has finished reading `acur`,
    // (b) every wave's share of tile t+1 has landed (the global_load_lds were
    // issued before this tile's MFMAs; the only other outstanding vector-memory
    // ops are the previous tile's output stores, long retired).  The epilogue
    // below then runs without any inter-wave synchronisation and its stores
    // stay in flight across the next MFMA phase.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (AASR_DBG(1)) {  // ablation: MFMA only (keep the accumulators live)
      asm volatile("" ::"v"(c00), "v"(c01), "v"(c10), "v"(c11));
      continue;
    }
    // ---- epilogue, one 32-row chunk at a time
#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
      const f32x16 &ca = mb ? c10 : c00;
      const f32x16 &cb = mb ? c11 : c01;
      // C layout: lane (n,h), reg i -> row 8*(i/4) + 4*h + (i%4), col n
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int row = 8 * (i >> 2) + 4 * h + (i & 3);
        stage[row * FRAMES_PER_WAVE + n] = ca[i];
        stage[row * FRAMES_PER_WAVE + 32 + n] = cb[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const float *col = stage + lane;
      const int64_t chunk = t * 2 + mb;
      if (MODE == 0) {
        const int sb = chunk_seg_begin[chunk];
        const int se = chunk_seg_begin[chunk + 1];
        for (int si = sb; si < se; si++) {
          const uint32_t d = seg_desc[si];
          const int a = d & 0xff, b = (d >> 8) & 0xff;
          const bool cont = (d >> 16) & 1, open = (d >> 17) & 1;
          float m = cont ? carry_m : NEG_BIG_F;
          float s = cont ? carry_s : 0.0f;
          reduce_rows(col, a, b, m, s);
          if (open) {
            carry_m = m;
            carry_s = s;
          } else {
            float lg = __builtin_amdgcn_logf(s);  // log2
            float ll = fmaf(m, LN2_F, lg * LN2_F);
            ll = fmaxf(ll, LOG_TINY_F);
            // states close in index order: buffer [frame][k], k = n_closed % 8
            const int k = n_closed & (OUT_GROUP - 1);
            ost[lane * OUT_GROUP + k] = ll;
            n_closed++;
            if ((n_closed & (OUT_GROUP - 1)) == 0 || n_closed == (int)out_cols) {
              const int cnt = ((n_closed - 1) & (OUT_GROUP - 1)) + 1;
              const int s_base = n_closed - cnt;
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              const int kk2 = lane & (OUT_GROUP - 1);
#pragma unroll
              for (int i = 0; i < FRAMES_PER_WAVE / (64 / OUT_GROUP); i++) {
                const int j = i * (64 / OUT_GROUP) + (lane / OUT_GROUP);
                const float v = ost[j * OUT_GROUP + kk2];
                if (kk2 < cnt && f0 + j < F) out[(f0 + j) * out_cols + s_base + kk2] = v;
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
      } else {
        const int64_t rbase = chunk * CHUNK_ROWS;
        if (frame_ok) {
          for (int r = 0; r < CHUNK_ROWS; r++) {
            if (rbase + r < rows)
              out[my_frame * out_cols + rbase + r] = col[r * FRAMES_PER_WAVE] * LN2_F;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}


// ---------------------------------------------------------------------------
// Track layouts: in-register epilogue (see gmm_build_tracks()).
//
// The host lays every state on one of the two row tracks that lane halves
// h = 0 / 1 hold in their accumulator registers and folds a fixed reference
// 2^ref into the constants, so the epilogue is 16 v_exp_f32 + 16 adds per
// accumulator block with no LDS round trip, no running maximum and one 16-bit
// close mask per tile.  On gfx950 the f32 MFMA executes on the same lanes as
// the VALU (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so every VALU instruction removed
// from the epilogue is matrix time won back.
//   GROUPED: states 2j/2j+1 finish together; results are transposed through a
//            wave-private LDS buffer and written 32 consecutive states (128 B)
//            per frame row with 16-byte stores.
//   !GROUPED: the tracks close states independently; results are stored per
//            state (4-byte scatter, one store instruction per 32 frames).
// The row range can be cut (blockIdx.y) so that the grid has no tail round.
// ---------------------------------------------------------------------------
// Gaussian-clustering hook of the track kernels (CL = true; see gmm_cluster.hip).
// One bit per (packed row, frame): 1 = use the Gaussian's exact value, 0 = the row
// contributes nothing here (its cluster centre is added by k_cluster_merge).
// k_cluster_expand stores the bits PER LANE: lane (n, h) of the wave that owns frames
// f0 .. f0+63 holds, for one tile, the 64 accumulator values {mb, q, e, side}
// (rows 32 mb + 8q + 4h + e, frames f0 + 32 side + n), so maskrow[word][tile][lane] is one
// 64-bit word with bit ((mb*4 + q)*4 + e)*2 + side.  A wave fetches its 512 bytes for the NEXT
// tile with one coalesced vector load issued in the middle of the matrix stream; the first
// version read ready-made 64-lane masks through the scalar cache (6 GB per 10^6 frames that
// missed it: +7.6 ms of exposed waits).
struct ClusterArgs {
  const unsigned long long *maskrow = nullptr;
  int64_t rows_padded = 0;
  float floor_val = LOG_TINY_F;
};

// The table is read-only for the whole launch: addressing it through the
// constant address space lets the compiler use scalar loads (plain global loads
// are not scalarised in a kernel that also stores).
// value if this lane's bit `idx` (compile-time) of the tile's word is set, a large negative
// exponent otherwise
__device__ __forceinline__ float mask_select(float x, unsigned long long bits, int idx) {
  const unsigned half = idx < 32 ? (unsigned)bits : (unsigned)(bits >> 32);
  // two instructions per value: the bit sign-extended to a lane mask (v_bfe_i32), then a bitfield
  // insert picks x or the constant (v_bfi_b32) -- and / compare / select is three
  const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)half, idx & 31, 1);
  const unsigned r = (__builtin_bit_cast(unsigned, x) & m) | (__builtin_bit_cast(unsigned, NEG_BIG_F) & ~m);
  return __builtin_bit_cast(float, r);
}

template <int NKK, bool GROUPED>
struct TrackSmem {
  static constexpr int OG = TRACK_OUT_GROUP;
  static constexpr int kTileFloats = (NKK / 2) * 64 * 4;
  static constexpr int kOutStride = OG + 4;  // 16-byte aligned rows for ds_read_b128
  static constexpr int kOutFloatsPerWave = GROUPED ? FRAMES_PER_WAVE * kOutStride : 0;
  static constexpr int kBytes = (2 * kTileFloats + WAVES_PER_BLOCK * kOutFloatsPerWave) * 4;
};

template <int NKK, bool GROUPED, bool CL>
__global__ __launch_bounds__(256, 2) void k_gmm_diag_score_tracks(
    const float *__restrict__ frames, int64_t F, int dim, const float *__restrict__ pivot,
    const float *__restrict__ apack, const int32_t *__restrict__ split_row,
    const uint16_t *__restrict__ close_mask, const int32_t *__restrict__ sid, int sid_stride,
    float *__restrict__ out, int64_t S, int64_t pitch, float ref_ln, int dbg, ClusterArgs cl) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *smem = (float *)smem_raw;
  constexpr int OG = TRACK_OUT_GROUP;
  constexpr int kTileFloats = TrackSmem<NKK, GROUPED>::kTileFloats;
  constexpr int kOS = TrackSmem<NKK, GROUPED>::kOutStride;
  float *abuf0 = smem;
  float *abuf1 = smem + kTileFloats;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  float *ost = smem + 2 * kTileFloats + wave * TrackSmem<NKK, GROUPED>::kOutFloatsPerWave;
  const int n = lane & 31;
  const int h = lane >> 5;
  const int64_t f0 = (int64_t)blockIdx.x * FRAMES_PER_BLOCK + wave * FRAMES_PER_WAVE;

  float bf[NKK][2];
#pragma unroll
  for (int nb = 0; nb < 2; nb++) {
    int64_t f = f0 + nb * 32 + n;
    if (f > F - 1) f = F - 1;
    const float *xr = frames + f * dim;
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) {
      const int kc = kk < dim ? kk : 0;
      const float xc = xr[kc] - pivot[kc];
      float v = h ? xc * xc : xc;
      if (kk == dim) v = h ? 0.0f : 1.0f;
      if (kk > dim) v = 0.0f;
      bf[kk][nb] = v;
    }
  }

  // this workgroup's share of the rows: tiles [t_begin, t_end)
  const int64_t t_begin = split_row[4 * blockIdx.y];
  const int64_t t_end = split_row[4 * blockIdx.y + 4];
  issue_tile_copy_raw(apack + (size_t)t_begin * kTileFloats, abuf0, kTileFloats, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float s0 = 0.0f, s1 = 0.0f;  // running sum_k 2^(v_k) of this lane's open state, frames n / 32+n
  // closes so far on this lane's track (GROUPED: pairs closed, same on both tracks)
  int closes = split_row[4 * blockIdx.y + 1 + (GROUPED ? 0 : h)];
  const int32_t *my_sid = sid + h * sid_stride;
  int next_sid = GROUPED ? 0 : my_sid[closes];
  float *orow0 = out + (f0 + n) * pitch;       // !GROUPED: this lane's two output rows
  float *orow1 = out + (f0 + 32 + n) * pitch;
  const bool ok0 = f0 + n < F, ok1 = f0 + 32 + n < F;
  const float floor_val = CL ? cl.floor_val : LOG_TINY_F;
  // this wave's 64 frames are one word of the selection masks; its per-lane bits of tile t
  const unsigned long long *mrow =
      CL ? cl.maskrow + (size_t)(f0 >> 6) * cl.rows_padded + lane : nullptr;
  unsigned long long bits_next = 0;
  if (CL && split_row[4 * blockIdx.y] < split_row[4 * blockIdx.y + 4])
    bits_next = mrow[(size_t)split_row[4 * blockIdx.y] * TILE_ROWS];

  // close bits of the next tile are requested (scalar) right after the barrier, one tile ahead
  unsigned pair_next = t_begin < t_end ? sload_close_pair(close_mask, t_begin) : 0u;
  for (int64_t t = t_begin; t < t_end; t++) {
    const int par = (int)((t - t_begin) & 1);
    float *acur = par ? abuf1 : abuf0;
    float *anext = par ? abuf0 : abuf1;
    if (t + 1 < t_end)
      issue_tile_copy_raw(apack + (size_t)(t + 1) * kTileFloats, anext, kTileFloats, wave, lane);
    const unsigned mask16 = close16_of_pair(pair_next, t);
    // GROUPED: both tracks carry the same bits -> wave-uniform branch
    const unsigned mask = GROUPED ? (mask16 & 0xffu) : (h ? (mask16 >> 8) : (mask16 & 0xffu));
    // this tile's selection bits arrived during the previous tile; the next tile's are requested
    // here and waited for by the vmcnt(0) in front of the end-of-tile barrier
    const unsigned long long bits = bits_next;
    if (CL && t + 1 < t_end) bits_next = mrow[(size_t)(t + 1) * TILE_ROWS];

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    const f32x4 *afrag = (const f32x4 *)acur + lane;
    f32x4 a0 = afrag[0];
    f32x4 a1 = afrag[(NKK / 2 > 1 ? 1 : 0) * 64];
#pragma unroll
    for (int q = 0; q < NKK / 2; q++) {
      // fetched two kk-pairs ahead; the scheduling barriers keep the compiler from sinking the
      // read to its first use (which exposes one LDS round trip per 8 MFMAs)
      const int qn = (q + 2 < NKK / 2) ? q + 2 : NKK / 2 - 1;
      __builtin_amdgcn_sched_barrier(0);
      f32x4 a2 = afrag[qn * 64];
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 av = a0;
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[2 * q][0], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[2 * q][1], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[2 * q][0], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[2 * q][1], c11, 0, 0, 0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[2 * q + 1][0], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[2 * q + 1][1], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[2 * q + 1][0], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[2 * q + 1][1], c11, 0, 0, 0);
      a0 = a1;
      a1 = a2;
    }

    // One barrier per tile: every wave is done reading `acur` and every wave's
    // share of tile t+1 has landed; the epilogue then needs no inter-wave sync.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < t_end) pair_next = sload_close_pair(close_mask, t + 1);

    if (AASR_DBG(1)) {  // ablation: MFMA only
      asm volatile("" ::"v"(c00), "v"(c01), "v"(c10), "v"(c11));
      continue;
    }

#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
      const f32x16 &ca = mb ? c10 : c00;
      const f32x16 &cb = mb ? c11 : c01;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        // this lane's quad q of the block: accumulator registers 4q .. 4q+3
        float va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          va[e] = ca[4 * q + e];
          vb[e] = cb[4 * q + e];
          if (CL) {
            va[e] = mask_select(va[e], bits, 8 * q + 4 * mb + e);        // k_cluster_expand's bit layout
            vb[e] = mask_select(vb[e], bits, 32 + 8 * q + 4 * mb + e);
          }
        }
        float e0 = __builtin_amdgcn_exp2f(va[0]) + __builtin_amdgcn_exp2f(va[1]);
        float e1 = __builtin_amdgcn_exp2f(va[2]) + __builtin_amdgcn_exp2f(va[3]);
        float g0 = __builtin_amdgcn_exp2f(vb[0]) + __builtin_amdgcn_exp2f(vb[1]);
        float g1 = __builtin_amdgcn_exp2f(vb[2]) + __builtin_amdgcn_exp2f(vb[3]);
        s0 += e0 + e1;
        s1 += g0 + g1;
        if ((mask >> (mb * 4 + q)) & 1) {
          float l0 = fmaf(__builtin_amdgcn_logf(s0), LN2_F, -ref_ln);
          float l1 = fmaf(__builtin_amdgcn_logf(s1), LN2_F, -ref_ln);
          l0 = fmaxf(l0, floor_val);
          l1 = fmaxf(l1, floor_val);
          s0 = 0.0f;
          s1 = 0.0f;
          closes++;
          if (!GROUPED) {
            if (ok0) orow0[next_sid] = l0;
            if (ok1) orow1[next_sid] = l1;
            next_sid = my_sid[closes];  // list is padded by one entry
          } else {
            const int pairs_closed = closes;
            const int slot = ((2 * (pairs_closed - 1)) & (OG - 1)) + h;
            ost[n * kOS + slot] = l0;
            ost[(32 + n) * kOS + slot] = l1;
            const int64_t closed = 2 * (int64_t)pairs_closed < S ? 2 * (int64_t)pairs_closed : S;
            if ((((2 * pairs_closed) & (OG - 1)) == 0 || 2 * (int64_t)pairs_closed >= S) &&
                !AASR_DBG(16)) {
              const int64_t s_base = ((closed - 1) / OG) * OG;
              const int cnt = (int)(closed - s_base);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if (cnt == OG && f0 + FRAMES_PER_WAVE <= F) {
                // full group: each lane moves 4 consecutive states (16 B) of one
                // frame row; 8 lanes cover the 32-state group, 8 rows per instruction
                const int k4 = lane & 7, r8 = lane >> 3;
                float *op = out + (f0 + r8) * pitch + s_base + 4 * k4;
                const float *ip = ost + r8 * kOS + 4 * k4;
#pragma unroll
                for (int i = 0; i < FRAMES_PER_WAVE / 8; i++) {
                  const f32x4 v = *(const f32x4 *)(ip + i * 8 * kOS);
                  // rows of the [F x S] output are only 4-byte aligned
                  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                  *(f32x4u *)(op + (int64_t)i * 8 * pitch) = v;
                }
              } else {
                constexpr int RPI = 64 / OG;  // frame rows per store instruction
                const int k = lane & (OG - 1);
#pragma unroll 4
                for (int i = 0; i < FRAMES_PER_WAVE / RPI; i++) {
                  const int row = i * RPI + lane / OG;
                  const float v = ost[row * kOS + k];
                  if (k < cnt && f0 + row < F) out[(f0 + row) * pitch + s_base + k] = v;
                }
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
      }
    }
  }
}

template <int NKK, bool GROUPED, bool CL>
static void launch_tracks_t(const aasr_gmm *g, const TrackLayout &L, const float *d_frames,
                            int64_t F, float *d_out, hipStream_t stream, const ClusterArgs &cl,
                            int64_t pitch) {
  const int64_t blocks = (F + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
  const int smem = TrackSmem<NKK, GROUPED>::kBytes;
  static const int dbg = AASR_EXPERIMENT_ENV("AASR_DBG") ? atoi(AASR_EXPERIMENT_ENV("AASR_DBG")) : 0;
  static bool attr_set[64] = {false};
  auto kern = k_gmm_diag_score_tracks<NKK, GROUPED, CL>;
  if (!attr_set[g->device & 63]) {
    AASR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[g->device & 63] = true;
  }
  // Row-range cuts: pick the number of cuts R that leaves the smallest tail
  // round on the chip (2 workgroups per CU resident), preferring fewer cuts.
  static const int force_r = AASR_EXPERIMENT_ENV("AASR_SPLITS") ? atoi(AASR_EXPERIMENT_ENV("AASR_SPLITS")) : 0;
  const double slots = 2.0 * (g->num_cus > 0 ? g->num_cus : 256);
  int R = 1;
  double best_eff = 0;
  for (int r = 1; r <= L.max_splits; r++) {
    double x = (double)blocks * r / slots;
    double eff = x / std::ceil(x);
    if (eff > best_eff + 0.005) {
      best_eff = eff;
      R = r;
    }
  }
  if (force_r >= 1 && force_r <= L.max_splits) R = force_r;
  const int32_t *split_row = L.splits.p + (size_t)(R - 1) * (L.split_cap + 1) * 4;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)R), dim3(256), smem, stream, d_frames, F,
                     g->dim, g->d_pivot.p, L.rows.a.p, split_row, L.close.p, L.sid.p, L.sid_stride,
                     d_out, g->S, pitch, L.ref_ln - (float)g->out_bias_ln, dbg, cl);
  AASR_HIP(hipGetLastError());
}

static bool launch_tracks(const aasr_gmm *g, const TrackLayout &L, const float *d_frames, int64_t F,
                          float *d_out, hipStream_t stream, const ClusterArgs *cl = nullptr,
                          int64_t pitch = 0) {
  if (pitch <= 0) pitch = g->S;
  const ClusterArgs none;
  switch (L.rows.nkk) {
#define AASR_CASE(N)                                                                        \
  case N:                                                                                   \
    if (cl) {                                                                               \
      if (L.grouped) launch_tracks_t<N, true, true>(g, L, d_frames, F, d_out, stream, *cl, pitch); \
      else launch_tracks_t<N, false, true>(g, L, d_frames, F, d_out, stream, *cl, pitch);          \
    } else {                                                                                \
      if (L.grouped) launch_tracks_t<N, true, false>(g, L, d_frames, F, d_out, stream, none, pitch); \
      else launch_tracks_t<N, false, false>(g, L, d_frames, F, d_out, stream, none, pitch);        \
    }                                                                                       \
    return true;
    AASR_CASE(8) AASR_CASE(14) AASR_CASE(20) AASR_CASE(26) AASR_CASE(32) AASR_CASE(40)
    AASR_CASE(48) AASR_CASE(64)
#undef AASR_CASE
    default:
      return false;
  }
}

// ---------------------------------------------------------------------------
// bf16x3 variant of the track kernel (AASR_PREC_BF16X3).
//
// The f32 MFMA shares its lanes with the VALU and runs at 1/16 of the bf16
// matrix rate.  Here both operands are split into three bf16 terms
// (x = x1 + x2 + x3, 8 significant bits each, so the split is exact to 2^-24) and
// the six products of order <= 2^-16 are accumulated in f32 by
// v_mfma_f32_32x32x16_bf16:  a1b3 + a2b2 + a3b1 + a1b2 + a2b1 + a1b1, i.e. six
// K = 16 MFMAs per 16 values of K -- 0.375x the matrix cycles of the f32 form,
// f32-class accuracy (dropped terms are 2^-24 relative; every MFMA rounds once
// per 16 products instead of once per product), and the bf16 pipe co-executes
// with the VALU epilogue of the other wave on the SIMD.
// K order (constant first, then interleaved): k = 0: the constant (B = 1), k = 1: the constant's remainder (f16x2; B = 1),
// k = 2 + 2 d: linear term of dimension d, k = 3 + 2 d: its quadratic term, zero beyond; K = 16 NK16 >= 2 dim + 2.  A
// dimension's two terms -- p mu' x' and -p/2 x'^2, each as large as the conditioning estimates say and of opposite sign
// -- meet inside ONE matrix instruction, whose 16 products are summed before the f32 accumulator rounds, and the chain
// starts from the constant (which holds -kappa/2): the running sum then moves from C towards the result by
// (kappa_d - z_d^2)/2 per dimension and never leaves their range.  (Until round 5 the order was all linear terms, the
// constant, then all quadratic terms: the accumulator climbed to the linear terms' sum, ~kappa + sqrt(kappa) |z| log2
// units, and every later instruction rounded at that magnitude -- the dominant error of both split forms on models
// fitted to data, 1.6e-4 on visible values where this order gives 6e-5.)
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_bits_rne(float x) {
  unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// three-term split of two floats, packed pairwise (lo = first value)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &p1, unsigned &p2,
                                            unsigned &p3) {
  unsigned a1 = bf16_bits_rne(x0), b1 = bf16_bits_rne(x1);
  float r0 = x0 - __uint_as_float(a1 << 16), r1 = x1 - __uint_as_float(b1 << 16);
  unsigned a2 = bf16_bits_rne(r0), b2 = bf16_bits_rne(r1);
  r0 -= __uint_as_float(a2 << 16);
  r1 -= __uint_as_float(b2 << 16);
  unsigned a3 = bf16_bits_rne(r0), b3 = bf16_bits_rne(r1);
  p1 = a1 | (b1 << 16);
  p2 = a2 | (b2 << 16);
  p3 = a3 | (b3 << 16);
}

// ---------------------------------------------------------------------------
// f16x2 variant (AASR_PREC_F16X2): the same kernel with both operands carried as TWO fp16 terms
// (hi = fp16(x), lo = fp16(x - hi): 22 significant bits) and the three products hi*hi, hi*lo, lo*hi
// accumulated in f32 by v_mfma_f32_32x32x16_f16 -- half the matrix instructions of the bf16x3 form.
// What it gives up is 2 bits per operand: measured on 10^7 states of the configs[1] model the worst
// state-level error is 3.4e-5 against 1.9e-5 (tools/exp_fp16_split.py), and the error grows with the
// model's conditioning estimate as the other forms' does, so it is only chosen below tighter limits
// (KAPPA_LIMIT_F16, gmm.h); models above them keep the bf16x3 form.  The constant rides in TWO K slots
// (k = 2 dim and k = 2 dim + 1, the frame operand is 1 in both): 44 bits, so the largest term of the sum
// loses nothing.  fp16 range: the frame operand is clamped to |x - pivot| <= kF16Clamp (its square stays
// finite); load-time eligibility guarantees that a frame that far out is at the 1e-50 floor either way.
// ---------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int NS>
__device__ __forceinline__ f32x16 mfma_split(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
  if constexpr (NS == 3)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two-term fp16 split of two floats, packed pairwise (lo half = first value)
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned &p1, unsigned &p2) {
  const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
  const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
  p1 = __builtin_bit_cast(unsigned, (f16x2){h0, h1});
  p2 = __builtin_bit_cast(unsigned, (f16x2){l0, l1});
}

// ---------------------------------------------------------------------------
// Frame operand of the two-term kernels, one unit: the 8 K-slot values of (frame row xr, slab j, K half h) around `pivot`
// -- (x - pivot), the dimension's clamp and the column's power-of-two scale, the square for the quadratic slots, 1 in the
// constant's slots -- the arithmetic of k_frame_operand (below), shared with the multi-pivot instances of
// k_gmm_diag_score_pl, which form their group's operand in their prologue (round 6: one image per pivot group through
// HBM was 320 B per frame and group, 0.46 ms of a fitted model's 11.1).  f16tab: [2 KH] column scales, [KH] clamps.
// ---------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void fop_unit_f16(const float *__restrict__ xr, int dim, const float *__restrict__ pivot,
                                             const float *__restrict__ f16tab, int KH, int j, int h, int sc,
                                             unsigned w1[4], unsigned w2[4]) {
  const int k0 = 16 * j + 8 * h;
  const int d0 = sc ? 7 * j + (h ? 3 : -1) : (k0 >> 1) - 1;
  const bool whole = !sc && d0 >= 0 && d0 + 4 <= dim;   // uniform per K half
  float v[8];
  if (whole) {
    const f32x4u a = *(const f32x4u *)(xr + d0);
    const f32x4u b = *(const f32x4u *)(pivot + d0);
    const f32x4u c = *(const f32x4u *)(f16tab + 2 * KH + d0);
    const f32x4u e0 = *(const f32x4u *)(f16tab + k0);
    const f32x4u e1 = *(const f32x4u *)(f16tab + k0 + 4);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float xc = a[i] - b[i];
      const float xq = fminf(fmaxf(xc, -c[i]), c[i]);   // fp16 range: the dimension's clamp (pack_f16x2)
      v[2 * i] = xq * (i < 2 ? e0[2 * i] : e1[2 * i - 4]);
      v[2 * i + 1] = (xq * xq) * (i < 2 ? e0[2 * i + 1] : e1[2 * i - 3]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int k = k0 + i;
      const int d = (sc ? (h == 0 && i < 2) : k < 2) ? -1 : d0 + (i >> 1);
      const int dc = d >= 0 && d < dim ? d : 0;
      const float xc = xr[dc] - pivot[dc];
      const float lim = f16tab[2 * KH + dc];
      const float xq = fminf(fmaxf(xc, -lim), lim);
      float val = (k & 1) ? xq * xq : xq;
      if (d < 0) val = 1.0f;   // the constant and its remainder
      else if (d >= dim) val = 0.0f;
      v[i] = val * f16tab[k];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) split2_pair(v[2 * i], v[2 * i + 1], w1[i], w2[i]);
}

template <int NK16, bool GROUPED, bool WIDE = false, int NS = 3>
struct Bf16Smem {
  static constexpr int kTileBytes = NK16 * NS * 2 * 64 * 16;
  // States per output group.  4-wave form: 16 (LDS budget of 2 workgroups per CU).  8-wave form: 32
  // where three tile buffers + eight staging areas of stride 34 still fit 160 KB -- a group is then
  // a whole 128-byte L2 line of a padded output row, written by one store instruction.
  static constexpr bool kBig = WIDE && GROUPED && 3 * kTileBytes + 8 * FRAMES_PER_WAVE * 34 * 4 <= 160 * 1024;
  static constexpr int OG = kBig ? 32 : 16;
  static constexpr int kOutStride = kBig ? 34 : 20;
  static constexpr int kOutFloatsPerWave = GROUPED ? FRAMES_PER_WAVE * kOutStride : 0;
};

// WIDE: one workgroup of 8 waves (512 frames) per CU instead of two of 4 waves, so a tile is
// fetched from L2 once per 512 frames -- the L2 -> LDS tile traffic is what the power-capped
// matrix stream pays for (measured: no traffic -6.3 ms, half of it -2.1 ms of 34.9).  The two wave
// groups run the same tile sequence half a tile apart (group 1 lags by one barrier; every wave
// passes two barriers per tile, one in the middle of its stream), which puts one group's
// epilogue under the other group's matrix stream; three tile buffers make the lag legal.
template <int NK16, bool GROUPED, bool CL, bool WIDE, int NS>
__global__ __launch_bounds__(WIDE ? 512 : 256, WIDE ? 1 : 2) void k_gmm_diag_score_bf16x3(
    const float *__restrict__ frames, int64_t F, int dim, const float *__restrict__ pivot,
    const uint16_t *__restrict__ apack, const int32_t *__restrict__ split_row,
    const uint16_t *__restrict__ close_mask, const int32_t *__restrict__ sid, int sid_stride,
    float *__restrict__ out, int64_t S, int64_t pitch, float ref_ln, int dbg, ClusterArgs cl) {
  // three bf16 terms only: the two-term fp16 arithmetic (per-column scales, per-dimension clamps: pack_f16x2) lives in
  // k_gmm_diag_score_pl; the NS == 2 paths below are what is left of its first home and know neither
  static_assert(NS == 3, "k_gmm_diag_score_bf16x3 is instantiated for the three-term form only");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int OG = Bf16Smem<NK16, GROUPED, WIDE, NS>::OG;
  constexpr int kTileBytes = Bf16Smem<NK16, GROUPED, WIDE, NS>::kTileBytes;
  constexpr int kTileFloats = kTileBytes / 4;
  constexpr int kOS = Bf16Smem<NK16, GROUPED, WIDE, NS>::kOutStride;
  constexpr int NW = WIDE ? 8 : 4;    // waves per workgroup
  constexpr int NBUF = WIDE ? 3 : 2;  // tile buffers
  // WIDE: slabs before the mid-stream barrier.  Between two barriers one wave of a SIMD runs the slabs behind its
  // mid-stream barrier while its partner runs an epilogue FOLLOWED BY the slabs in front of it, so the partner's
  // stretch is epilogue + JMID slabs of matrix time against (NK16 - JMID) slabs here.  With 30 MFMAs per slab
  // (bf16x3) the epilogue (~1500 cycles: 64 v_exp_f32 at quarter rate + the adds) is the smaller part and the even
  // split is fine; with 12 (f16x2) it is four slabs' worth, so it is paired with ONE slab: 1500 + 384 against 1536
  // cycles and a matrix pipe that is busy 1920 of them (measured: JMID = 3 leaves 59.6 M cycles per launch
  // for 22.9 M of matrix work, the critical wave being epilogue + 36 MFMAs long).
  constexpr int JMID = NS == 2 ? 1 : (NK16 + 1) / 2;
  // f16x2: the fragments of slab j + 1 are requested at the top of slab j into a second register set (12 MFMAs =
  // 384 cycles ahead); the rolling refill of the three-term form would leave them 4 MFMAs
  constexpr int NAB = NS == 2 ? 2 : 1;
  float *abuf0 = (float *)smem_raw;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int group = WIDE ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;
  float *ost = abuf0 + NBUF * kTileFloats + wave * Bf16Smem<NK16, GROUPED, WIDE, NS>::kOutFloatsPerWave;
  const int n = lane & 31;
  const int h = lane >> 5;  // K half of a slab held by this lane AND its row track
  const int64_t f0 = (int64_t)blockIdx.x * (NW * FRAMES_PER_WAVE) + wave * FRAMES_PER_WAVE;

  // ---- frame operand: lane (n, h) holds k = 16*j + 8*h + i, i < 8, of slab j
  u32x4 bq[NK16][NS][2];
#pragma unroll
  for (int nb = 0; nb < 2; nb++) {
    int64_t f = f0 + nb * 32 + n;
    if (f > F - 1) f = F - 1;
    const float *xr = frames + f * dim;
#pragma unroll
    for (int j = 0; j < NK16; j++) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int k = 16 * j + 8 * h + i;
        const int d = (k >> 1) - 1;   // k = 0 / 1: the constant's slots
        const int dc = d >= 0 && d < dim ? d : 0;
        const float xc = xr[dc] - pivot[dc];
        float xq = xc;
        if (NS == 2) xq = fminf(fmaxf(xc, -kF16Clamp), kF16Clamp);  // fp16 range (see the f16x2 note above)
        float val = (k & 1) ? xq * xq : xq;
        if (d < 0) val = (k == 0 || NS == 2) ? 1.0f : 0.0f;
        else if (d >= dim) val = 0.0f;
        v[i] = val;
      }
      if constexpr (NS == 3) {
        unsigned w1[4], w2[4], w3[4];
#pragma unroll
        for (int i = 0; i < 4; i++) split3_pair(v[2 * i], v[2 * i + 1], w1[i], w2[i], w3[i]);
        bq[j][0][nb] = u32x4{w1[0], w1[1], w1[2], w1[3]};
        bq[j][1][nb] = u32x4{w2[0], w2[1], w2[2], w2[3]};
        bq[j][2][nb] = u32x4{w3[0], w3[1], w3[2], w3[3]};
      } else {
        unsigned w1[4], w2[4];
#pragma unroll
        for (int i = 0; i < 4; i++) split2_pair(v[2 * i], v[2 * i + 1], w1[i], w2[i]);
        bq[j][0][nb] = u32x4{w1[0], w1[1], w1[2], w1[3]};
        bq[j][1][nb] = u32x4{w2[0], w2[1], w2[2], w2[3]};
      }
    }
  }

  const int64_t t_begin = split_row[4 * blockIdx.y];
  const int64_t t_end = split_row[4 * blockIdx.y + 4];
  const float *apf = (const float *)apack;
  issue_tile_copy_raw(apf + (size_t)t_begin * kTileFloats, abuf0, kTileFloats, wave, lane, NW);
  if (WIDE && t_begin + 1 < t_end)
    issue_tile_copy_raw(apf + (size_t)(t_begin + 1) * kTileFloats, abuf0 + kTileFloats, kTileFloats, wave, lane, NW);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (WIDE && group == 1) __builtin_amdgcn_s_barrier();  // the lagging group's "end of tile -1"

  float s0 = 0.0f, s1 = 0.0f;
  int closes = split_row[4 * blockIdx.y + 1 + (GROUPED ? 0 : h)];
  const int32_t *my_sid = sid + h * sid_stride;
  int next_sid = GROUPED ? 0 : my_sid[closes];
  float *orow0 = out + (f0 + n) * pitch;  // pitch: row stride of `out` in floats (>= S)
  float *orow1 = out + (f0 + 32 + n) * pitch;
  const bool ok0 = f0 + n < F, ok1 = f0 + 32 + n < F;
  const float floor_val = CL ? cl.floor_val : LOG_TINY_F;
  // this wave's 64 frames are one word of the selection masks; its per-lane bits of tile t
  const unsigned long long *mrow =
      CL ? cl.maskrow + (size_t)(f0 >> 6) * cl.rows_padded + lane : nullptr;
  unsigned long long bits_next = 0;
  if (CL && split_row[4 * blockIdx.y] < split_row[4 * blockIdx.y + 4])
    bits_next = mrow[(size_t)split_row[4 * blockIdx.y] * TILE_ROWS];

  if (AASR_DBG(4)) {  // experiment: de-phase co-resident workgroups
    unsigned hsh = ((unsigned)blockIdx.x + 977u * blockIdx.y) * 2654435761u;
    int bucket = (hsh >> 28) & 15;
    for (int i = 0; i < bucket; i++) __builtin_amdgcn_s_sleep(8);
  }
  if (AASR_DBG(8)) {  // experiment: raise priority of every second workgroup
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(2);
  }
  // Close bits of a tile: a VECTOR load issued in the middle of the previous tile's matrix stream
  // and turned into a scalar after that tile's barrier, whose vmcnt(0) covers it.  Two traps are
  // avoided this way: loaded at the top of a tile, the compiler waits for it with vmcnt(0) right
  // behind the tile copy (vmcnt is in issue order; measured 5 ms of 30 in the matrix stream), and a
  // scalar load anywhere in the loop turns every LDS wait of the stream into lgkmcnt(0).
  unsigned mask16_next = t_begin < t_end ? (unsigned)__builtin_amdgcn_readfirstlane((int)close_mask[t_begin]) : 0u;
  unsigned mask_v = 0;
  u32x4 afr[NAB][NS][2];  // A fragments of the current slab, [register set][split][row block]
  if (t_begin < t_end) {
#pragma unroll
    for (int sp = NS - 1; sp >= 0; sp--) {
      afr[0][sp][0] = ((const u32x4 *)abuf0 + lane)[(sp * 2 + 0) * 64];
      afr[0][sp][1] = ((const u32x4 *)abuf0 + lane)[(sp * 2 + 1) * 64];
    }
  }
  int bi = 0;  // buffer of the current tile
  for (int64_t t = t_begin; t < t_end; t++) {
    float *acur = abuf0 + bi * kTileFloats;
    const int bn = bi + 1 < NBUF ? bi + 1 : 0;         // buffer of tile t+1
    const int bnn = bn + 1 < NBUF ? bn + 1 : 0;        // WIDE: buffer of tile t+2 (held tile t-1)
    float *anext = abuf0 + bn * kTileFloats;
    if (!WIDE) {
      if (t + 1 < t_end && !AASR_DBG(256) && !(AASR_DBG(512) && (t & 1)))  // ablations: 256 no tile traffic, 512 half of it
        issue_tile_copy_raw(apf + (size_t)(t + 1) * kTileFloats, anext, kTileFloats, wave, lane, NW);
    } else if (group == 1 && t + 2 < t_end && !AASR_DBG(256)) {
      // both groups are past tile t-1 once the lagging group has passed its end-of-tile barrier
      issue_tile_copy_raw(apf + (size_t)(t + 2) * kTileFloats, abuf0 + bnn * kTileFloats, kTileFloats, wave, lane, NW);
    }
    bi = bn;
    const unsigned mask16 = mask16_next;
    const unsigned mask = GROUPED ? (mask16 & 0xffu) : (h ? (mask16 >> 8) : (mask16 & 0xffu));
    // this tile's selection bits arrived during the previous tile; the next tile's are requested
    // here and waited for by the vmcnt(0) in front of the end-of-tile barrier
    const unsigned long long bits = bits_next;
    if (CL && t + 1 < t_end) bits_next = mrow[(size_t)(t + 1) * TILE_ROWS];

    if (AASR_DBG(32)) __builtin_amdgcn_s_setprio(3);
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    const u32x4 *afrag = (const u32x4 *)acur + lane;  // [slab][split][mb][64 lanes]
    // Rolling A-fragment prefetch.  The products of a slab are ordered by the A split they use,
    // (a3,b1) | (a2,b2) (a2,b1) | (a1,b3) (a1,b2) (a1,b1), so each split's registers fall free as
    // early as possible and are refilled for the NEXT slab right then: every ds_read has 12-20
    // MFMAs (>= 384 cycles) to land and no extra registers are needed.  Left to itself the
    // compiler sinks the reads to their first use (one exposed LDS round trip per slab), hence
    // the full scheduling barriers.  The 2^-16 products still precede the 2^-8 ones of the same
    // A split; the sum already holds earlier slabs, so the order inside a slab is immaterial.
#pragma unroll
    for (int j = 0; j < NK16; j++) {
      if (WIDE && j == JMID) {
        // mid-stream barrier = the other group's end-of-tile barrier
        if (!AASR_DBG(16)) __builtin_amdgcn_s_barrier();
        if (group == 0 && t + 2 < t_end && !AASR_DBG(256))
          issue_tile_copy_raw(apf + (size_t)(t + 2) * kTileFloats, abuf0 + bnn * kTileFloats, kTileFloats, wave, lane, NW);
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int kNoSet = 0;
      const int cur = NAB == 2 ? (j & 1) : kNoSet;
      if (NAB == 2 && j + 1 < NK16 && !AASR_DBG(2)) {
#pragma unroll
        for (int sp = NS - 1; sp >= 0; sp--) {
          afr[cur ^ 1][sp][0] = afrag[(((j + 1) * NS + sp) * 2 + 0) * 64];
          afr[cur ^ 1][sp][1] = afrag[(((j + 1) * NS + sp) * 2 + 1) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int grp = 0; grp < NS; grp++) {
        const int sp = NS - 1 - grp;   // A split used by this group: a3, a2, a1 (f16x2: a2, a1)
        const int nprod = grp + 1;     // paired with b1 | b2 b1 | b3 b2 b1
#pragma unroll
        for (int c = 0; c < nprod; c++) {
          const int sb = nprod - 1 - c;
          c00 = mfma_split<NS>(afr[cur][sp][0], bq[j][sb][0], c00);
          c01 = mfma_split<NS>(afr[cur][sp][0], bq[j][sb][1], c01);
          c10 = mfma_split<NS>(afr[cur][sp][1], bq[j][sb][0], c10);
          c11 = mfma_split<NS>(afr[cur][sp][1], bq[j][sb][1], c11);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the aligned word holding tile t+1's bits (the array has a spare element); a 16-bit load
        // would need a zero-extension, which the compiler places -- with its vmcnt wait -- right here
        if (j == (NK16 > 1 ? 1 : 0) && grp == 0) mask_v = ((const uint32_t *)close_mask)[(t + 1) >> 1];
        if (NAB == 1 && j + 1 < NK16 && !AASR_DBG(2)) {
          afr[kNoSet][sp][0] = afrag[(((j + 1) * NS + sp) * 2 + 0) * 64];
          afr[kNoSet][sp][1] = afrag[(((j + 1) * NS + sp) * 2 + 1) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    if (WIDE && JMID >= NK16) {
      __builtin_amdgcn_s_barrier();
      if (group == 0 && t + 2 < t_end)
        issue_tile_copy_raw(apf + (size_t)(t + 2) * kTileFloats, abuf0 + bnn * kTileFloats, kTileFloats, wave, lane, NW);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(mask_v) : : "memory");
    if (!AASR_DBG(16)) __builtin_amdgcn_s_barrier();
    mask16_next = (unsigned)__builtin_amdgcn_readfirstlane((int)mask_v);
    mask16_next = ((t + 1) & 1) ? mask16_next >> 16 : mask16_next & 0xffffu;
    if (t + 1 < t_end) {
      // slab 0 of the next tile: in flight while the epilogue runs
      const u32x4 *nfrag = (const u32x4 *)anext + lane;
#pragma unroll
      for (int sp = NS - 1; sp >= 0; sp--) {
        afr[0][sp][0] = nfrag[(sp * 2 + 0) * 64];
        afr[0][sp][1] = nfrag[(sp * 2 + 1) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    if (AASR_DBG(1)) {
      asm volatile("" ::"v"(c00), "v"(c01), "v"(c10), "v"(c11));
      continue;
    }
    if (AASR_DBG(32)) __builtin_amdgcn_s_setprio(0);   // experiment: the epilogue yields to the partner's matrix stream

#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
      const f32x16 &ca = mb ? c10 : c00;
      const f32x16 &cb = mb ? c11 : c01;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          va[e] = ca[4 * q + e];
          vb[e] = cb[4 * q + e];
          if (CL) {
            va[e] = mask_select(va[e], bits, 8 * q + 4 * mb + e);        // k_cluster_expand's bit layout
            vb[e] = mask_select(vb[e], bits, 32 + 8 * q + 4 * mb + e);
          }
        }
        float e0 = __builtin_amdgcn_exp2f(va[0]) + __builtin_amdgcn_exp2f(va[1]);
        float e1 = __builtin_amdgcn_exp2f(va[2]) + __builtin_amdgcn_exp2f(va[3]);
        float g0 = __builtin_amdgcn_exp2f(vb[0]) + __builtin_amdgcn_exp2f(vb[1]);
        float g1 = __builtin_amdgcn_exp2f(vb[2]) + __builtin_amdgcn_exp2f(vb[3]);
        s0 += e0 + e1;
        s1 += g0 + g1;
        if ((mask >> (mb * 4 + q)) & 1) {
          float l0 = fmaf(__builtin_amdgcn_logf(s0), LN2_F, -ref_ln);
          float l1 = fmaf(__builtin_amdgcn_logf(s1), LN2_F, -ref_ln);
          l0 = fmaxf(l0, floor_val);
          l1 = fmaxf(l1, floor_val);
          s0 = 0.0f;
          s1 = 0.0f;
          closes++;
          if (!GROUPED) {
            if (ok0) orow0[next_sid] = l0;
            if (ok1) orow1[next_sid] = l1;
            next_sid = my_sid[closes];
          } else {
            const int pairs_closed = closes;
            const int slot = ((2 * (pairs_closed - 1)) & (OG - 1)) + h;
            ost[n * kOS + slot] = l0;
            ost[(32 + n) * kOS + slot] = l1;
            const int64_t closed = 2 * (int64_t)pairs_closed < S ? 2 * (int64_t)pairs_closed : S;
            if (((2 * pairs_closed) & (OG - 1)) == 0 || 2 * (int64_t)pairs_closed >= S) {
              const int64_t s_base = ((closed - 1) / OG) * OG;
              const int cnt = (int)(closed - s_base);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              if (OG == 32 && cnt == OG && f0 + FRAMES_PER_WAVE <= F) {
                // 8 lanes x 16 B cover the 32-state group; 8 frame rows per instruction
                const int k4 = lane & 7, r8 = lane >> 3;
                float *op = out + (f0 + r8) * pitch + s_base + 4 * k4;
                const float *ip = ost + r8 * kOS + 4 * k4;  // stride 34: 8-byte aligned
#pragma unroll
                for (int i = 0; i < FRAMES_PER_WAVE / 8; i++) {
                  const f32x2 lo = *(const f32x2 *)(ip + i * 8 * kOS);
                  const f32x2 hi = *(const f32x2 *)(ip + i * 8 * kOS + 2);
                  const f32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                  *(f32x4u *)(op + (int64_t)i * 8 * pitch) = v;
                }
              } else if (cnt == OG && f0 + FRAMES_PER_WAVE <= F) {
                // 4 lanes x 16 B cover the 16-state group; 16 frame rows per instruction
                const int k4 = lane & 3, r16 = lane >> 2;
                float *op = out + (f0 + r16) * pitch + s_base + 4 * k4;
                const float *ip = ost + r16 * kOS + 4 * k4;
#pragma unroll
                for (int i = 0; i < FRAMES_PER_WAVE / 16; i++) {
                  const f32x4 v = *(const f32x4 *)(ip + i * 16 * kOS);
                  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                  *(f32x4u *)(op + (int64_t)i * 16 * pitch) = v;
                }
              } else {
                constexpr int RPI = 64 / OG;
                const int k = lane & (OG - 1);
#pragma unroll 4
                for (int i = 0; i < FRAMES_PER_WAVE / RPI; i++) {
                  const int row = i * RPI + lane / OG;
                  const float v = ost[row * kOS + k];
                  if (k < cnt && f0 + row < F) out[(f0 + row) * pitch + s_base + k] = v;
                }
              }
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
      }
    }
  }
  if (WIDE && group == 0) __builtin_amdgcn_s_barrier();  // pairs with the lagging group's last end-of-tile barrier
}


// ---------------------------------------------------------------------------
// Software-pipelined form of the split-operand kernel (the f16x2 arithmetic runs on it).
//
// With three fp16 products per K slab a wave-tile is 60 MFMAs = 1 920 matrix cycles, and its epilogue -- 64
// v_exp_f32 at quarter rate, the adds, the close logic -- is ~1 600 VALU cycles: no longer the small part.  The
// phase-shifted wave groups of the kernel above only hide an epilogue under the PARTNER wave's matrix stream, and
// measured that hides about half of it (rocprofv3: 37.9 M cycles per 10^6-frame launch for 22.9 M of matrix work,
// VALU co-executing under 41 % of the MFMA cycles; s_setprio either way changes nothing).  What does hide is VALU
// placed between a wave's OWN MFMAs: an MFMA occupies the matrix pipe for 32 cycles, the in-order wave issues its
// next instructions meanwhile.  So the tile is processed as two half tiles (its two 32-row blocks), and the matrix
// stream of one block carries the exponentials of the other:
//
//     H0(t): 30 MFMAs into block 0 of tile t     ||  2^x and quad sums of block 1 of tile t-1
//            close logic of block 1, tile t-1          (branches, log, staging, stores: not interleaved)
//     H1(t): 30 MFMAs into block 1 of tile t     ||  2^x and quad sums of block 0 of tile t
//            s_waitcnt vmcnt(0); s_barrier; close logic of block 0, tile t
//
// Same 64 accumulator registers (a block is consumed before it is accumulated into again), two accumulator
// chains per phase instead of four (dependent MFMAs 64 cycles apart), one barrier per tile, two tile buffers, all
// waves of a workgroup in step -- no wave groups.  A fragments of the next slab are requested one slab (6 MFMAs)
// ahead into a second register set.  Everything else (operand layout, track epilogue, output groups, row cuts,
// selection masks) is the kernel above; results are bit-identical between the 4- and 8-wave forms.
// ---------------------------------------------------------------------------
// Work decomposition of a pipelined-kernel launch (see pick_cut_plan).
struct CutPlan {
  int n_main = 0;        // workgroups of the coarse part: blocks_main frame blocks x r_main cuts
  int blocks_main = 1;
  int blocks_rem = 1;    // frame blocks of the fine part
  int r_main = 1, r_rem = 0;
  const int32_t *split_rem = nullptr;   // cut table row of the fine part
};

// Pivot groups of a launch (nullptr colend: one pivot, the model's)
// log(exp(a) + exp(b)) for a state's two shares (matrix rows / outlier components, both with the 1e-50 floor, which the
// result keeps; a share AT the floor holds nothing).  The hardware's 2^x and log2 (1 ulp): 2e-7 on the result -- the
// library's expf / log1pf cost ~120 instructions per value, a seventh of the scoring kernel's time where 10 % of the
// states take this path in its close logic (k_gmm_diag_score_pl<..., HYB>); k_outlier_merge uses the same expression.
__device__ __forceinline__ float merge_floored_shares(float a, float b) {
  const float hi = fmaxf(a, b), lo = fminf(a, b);
  float r = hi;
  if (lo > LOG_TINY_F) r = fmaf(__builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f((lo - hi) * LOG2E_F)), LN2_F, hi);
  return fmaxf(r, LOG_TINY_F);
}

struct PivotGroups {
  const int32_t *colend = nullptr;   // [groups] one past the group's last output column
  int64_t fop_stride = 0;            // u32x4 elements between the groups' frame-operand images
  // PGF instances (the workgroup forms its group's operand in its prologue): the groups' pivots [groups][dim], their
  // column scales and clamps [groups][3 KH], the slab-constant flag of the layout
  const float *pivots = nullptr;
  const float *tabs = nullptr;
  int sc = 0;
  // HYB instances (outlier routing fused into the close logic, below): hyb_tab[s] = the next state >= s of s's track
  // parity that has outlier components (low 16 bits; 0xffff: none) and its record in the partial sums (high 16 bits);
  // the partial sums [records][pitch] (natural log, state-major, one row of frames per record:
  // k_gmm_diag_score_centred), log|det| of an in-place transform
  const uint32_t *hyb_tab = nullptr;
  const float *hyb_part = nullptr;
  int64_t hyb_pitch = 0;
  float hyb_bias = 0.0f;
};

template <int NK16, bool GROUPED, bool WIDE, int NS>
struct PlSmem {
  static constexpr int kTileBytes = NK16 * NS * 2 * 64 * 16;
  static constexpr int NBUF = WIDE ? 3 : 2;   // tile buffers (the 8-wave form's lagging group needs the third)
  static constexpr bool kBig = WIDE && GROUPED && NBUF * kTileBytes + 8 * FRAMES_PER_WAVE * 34 * 4 <= 160 * 1024;
  static constexpr int OG = kBig ? 32 : 16;
  static constexpr int kOutStride = kBig ? 34 : 20;
  static constexpr int kOutFloatsPerWave = GROUPED ? FRAMES_PER_WAVE * kOutStride : 0;
  static constexpr int kMapBytes = NBUF * 16 * 4;   // MAPPED: the tiles' pair entries, [buffer][track][quad position]
  static constexpr int kBytes = NBUF * kTileBytes + (WIDE ? 8 : 4) * kOutFloatsPerWave * 4 + kMapBytes;
};

// MAPPED (GROUPED only; NOT INSTANTIATED since round 6: it served section 0 of round 4's mixed layout, which the engine
// parts replaced; the branches stay in the kernel's source because the headline instance's code is better left untouched):
// the section's states are a SUBSET of
// the model's, so a pair's output columns come from a table instead of its ordinal: per tile, track and quad position
// `sid` holds column | flags of the pair that closes there (16 words per tile; they ride into LDS with the tile's rows,
// so the close logic never waits for global memory -- a per-close vector load is waited for in issue order, i.e. behind
// the tile copy requested just before it: measured +9 %).  Pairs are formed inside groups of 16 output columns, a
// group is staged and flushed as WHOLE lines exactly as in the unmapped kernel -- the columns of states the section does
// not hold go out with whatever the staging area holds, and the other section's launch, which comes second, stores its
// values over them.  (Both other forms were built and measured on configs[2] with 1 % of the states routed away: masked
// flushes that leave the foreign columns alone +5 %, direct 4-byte stores for the shared groups +12 % -- a partial
// write of a line costs a fill of that line, and every frame row has such a line wherever a state is missing.)
constexpr int kMapCol = 0xffffff;      // column field
constexpr int kMapEmpty = 1 << 28;     // this track holds no state in this pair (column: the partner's)
constexpr int kMapFlush16 = 1 << 29;   // last pair of its group of 16 columns
constexpr int kMapFlush32 = 1 << 30;   // ... of its group of 32

// AASR_PL_TRACE (experiment builds only, tools/pl_trace.py): where one workgroup's waves spend their cycles.  Every wave of
// workgroup AASR_PL_TRACE_BLOCK reads the shader clock (s_memtime) at the phase boundaries of its tile loop and sums the
// intervals: [0] H0 matrix phase, [1] close logic behind H0, [2] H1 matrix phase, [3] the tile barrier (wait + the next
// tile's copy issue; the lagging group passes it inside H0: its time is taken out of [0]), [4] fragment prefetch + close
// logic behind H1, [5] the part of [3] spent in s_barrier, [6] the part of [3] spent in s_waitcnt vmcnt(0), [7] whole kernel, [8] tiles, [9] / [10] of interval 4: the fragment prefetch, the close logic of block 0.  Reading the clock waits for every outstanding scalar and LDS
// operation, so the traced launch runs slower than the product kernel (the tool reports by how much).
#ifdef AASR_PL_TRACE
__device__ unsigned long long g_pl_trace[8][12];
#ifndef AASR_PL_TRACE_BLOCK
#define AASR_PL_TRACE_BLOCK 300
#endif
#define PL_TRACE_DECL unsigned long long tr_tiles = 0, tr_sub[3] = {0, 0, 0}, tr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_prev = __builtin_readcyclecounter(), tr_t0 = tr_prev, tr_bar = 0, tr_vm = 0
#define PL_TRACE(k) do { const unsigned long long tr_now = __builtin_readcyclecounter(); tr_acc[k] += tr_now - tr_prev; tr_prev = tr_now; } while (0)
#else
#define PL_TRACE_DECL
#define PL_TRACE(k)
#endif
#ifndef AASR_PL_PRIO
#define AASR_PL_PRIO 1   // issue priority of a wave of k_gmm_diag_score_pl inside its matrix phases (0: left alone)
#endif
// Priorities of the 8-wave form's two wave groups (AASR_PL_PRIO_SCHEME, round 5; found with the phase trace below).  The
// two waves of a SIMD share the matrix pipe whenever their matrix phases overlap, and at equal priority the arbiter gives
// the older wave -- the leading group's -- two thirds of it: the leading wave ran ahead through its H1 and then waited
// ~1 200 cycles per tile at the barrier for the lagging wave, whose H1 had crawled along beside it, and in that wait (the
// partner in its close logic, nobody in a matrix phase) the pipe idled 20 % of the time.  Scheme 1: the leading group
// takes the higher priority in H0 and the lower in H1, the lagging group the reverse -- in the long H1 / H1 overlap the
// lagging wave now wins, both groups reach the barrier together (waits 560 / 490 cycles in the traced build instead of
// 1 230 / 450), a tile takes 5 000 cycles instead of 5 450: configs[1] 18.77 -> 18.14 ms (-3.3 %), three alternating
// runs on one box.  Measured against it: the reverse assignment (scheme 2) 18.60, the lagging group higher throughout
// (the roles swap: 19.3), priorities 3 / 1 the same as 2 / 1, 1 / 0 18.44, the lagging group's barrier one or two slabs
// into H0 18.36 / 18.60.  0: every wave AASR_PL_PRIO in its matrix phases (round 4).
#ifndef AASR_PL_PRIO_SCHEME
#define AASR_PL_PRIO_SCHEME 1
#endif
#ifndef AASR_PL_PRIO_HI
#define AASR_PL_PRIO_HI 2
#endif
#ifndef AASR_PL_PRIO_LO
#define AASR_PL_PRIO_LO 1
#endif
// HYB (GROUPED, two terms, one pivot, unmasked; round 6): outlier routing without a merge pass.  The Gaussians the matrix
// layout left out (null rows) are summed per state by k_gmm_diag_score_centred BEFORE this launch, into a state-major
// buffer; a lane that closes such a state adds the buffer's value for its frame -- the arithmetic of k_outlier_merge,
// the same bits -- in front of the store.  The values are fetched a state ahead: per track parity a table says which state
// comes next and where its sums are; a state's two values (frames n, 32 + n) and the table entry of the state after it are
// requested when the previous one is consumed, so the close logic waits for global memory only where such states follow
// each other within a tile's time (the launcher leaves models where they are dense to the engine parts).  (k_outlier_merge's read-modify-write of one column of the score matrix
// touches a line per frame: 20 us per state and 449 280 frames, more than the gather of a model with engine parts from
// ~100 states on.)
template <int NK16, bool GROUPED, bool CL, bool WIDE, int NS, bool MAPPED = false, bool PGF = false, bool HYB = false>
__global__ __launch_bounds__(WIDE ? 512 : 256, WIDE ? 1 : 2) void k_gmm_diag_score_pl(
    const float *__restrict__ frames, int64_t F, int dim, const float *__restrict__ pivot,
    const uint16_t *__restrict__ apack, const int32_t *__restrict__ split_row,
    const uint16_t *__restrict__ close_mask, const int32_t *__restrict__ sid, int sid_stride,
    float *__restrict__ out, int64_t S, int64_t pitch, float ref_ln, int dbg, ClusterArgs cl,
    const u32x4 *__restrict__ fop, CutPlan plan, PivotGroups pg) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef PlSmem<NK16, GROUPED, WIDE, NS> SM;
  // work item -> (frame block, row cut): the first n_main workgroups take the coarse cuts of the frame blocks that fill
  // whole rounds of the chip, the rest the fine cuts of the remaining blocks (pick_cut_plan); within either part the cut
  // is the slow index, so the workgroups resident at one time stream the same rows
  int blk, cut;
  if ((int)blockIdx.x < plan.n_main) {
    cut = (int)blockIdx.x / plan.blocks_main;
    blk = (int)blockIdx.x - cut * plan.blocks_main;
  } else {
    const int b = (int)blockIdx.x - plan.n_main;
    cut = b / plan.blocks_rem;
    blk = plan.blocks_main + (b - cut * plan.blocks_rem);
    split_row = plan.split_rem;
  }
  blk = __builtin_amdgcn_readfirstlane(blk);
  cut = __builtin_amdgcn_readfirstlane(cut);
  PL_TRACE_DECL;
  constexpr int OG = SM::OG;
  constexpr int kTileFloats = SM::kTileBytes / 4;
  constexpr int kOS = SM::kOutStride;
  constexpr int NW = WIDE ? 8 : 4;
  constexpr int NPROD = NS * (NS + 1) / 2;       // products kept per slab: 3 (f16x2), 6 (bf16x3)
  constexpr int MPH = NK16 * NPROD * 2;          // MFMAs per phase (one 32-row block, two frame blocks)
  float *abuf0 = (float *)smem_raw;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  constexpr int NBUF = SM::NBUF;
  float *ost = abuf0 + NBUF * kTileFloats + wave * SM::kOutFloatsPerWave;
  int *emap = (int *)(abuf0 + NBUF * kTileFloats + NW * SM::kOutFloatsPerWave);   // MAPPED: [NBUF][2 tracks][8 positions]
  // a tile's rows and (MAPPED) its 16 pair entries into tile buffer `b`
  // The tile copy in the scalar-base form of the LDS-DMA instruction: the tile's address is wave-uniform, so the base goes
  // in a scalar register pair (two scalar additions per instruction) and the lanes carry ONE constant 32-bit offset,
  // 16 * lane, for the whole launch -- no 64-bit per-lane address to form and to send to the address unit per instruction
  // (AASR_PL_SADDR_COPY, default 1; 0: the generic per-lane pointers of issue_tile_copy_raw).  configs[1] 17.94 -> 17.84 ms,
  // configs[2] 10.42 -> 10.38 ms per step, alternating runs on one box; two registers fewer.  (Round 5 also spread the copy
  // instructions over the slabs of the H0 that follows the barrier instead of issuing them behind it -- the barrier interval
  // of the phase trace fell from ~550 to ~260 cycles and H0 grew by as much: an LDS-DMA instruction costs the issuing wave
  // 100-150 cycles wherever it stands; 1 % slower with twelve more registers, removed.  The whole copy issued by the
  // leading group alone, whose close logic follows the barrier: +0.7 %, removed.)
#ifndef AASR_PL_SADDR_COPY
#define AASR_PL_SADDR_COPY 1
#endif
  const unsigned lane_off16 = (unsigned)lane * 16u;
  auto issue_tile = [&](int64_t tile, int b) {
    if (AASR_PL_SADDR_COPY) {
      constexpr int kChunks = kTileFloats / 4 / 64;   // 1 KB instructions per tile
      const char *gbase = (const char *)((const float *)apack + (size_t)tile * kTileFloats);
      const unsigned lbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(abuf0 + b * kTileFloats));
      // a wave takes CONSECUTIVE 1 KB pieces: one scalar base, one M0, the pieces told apart by the instruction's immediate
      // offset (it moves the global and the LDS address alike) -- AASR_PL_COPY_IMM, default 1; 0: piece c = wave + k * NW,
      // a base and an M0 per instruction.  configs[1] 17.91 -> 17.82 ms, three alternating runs on one box.
#ifndef AASR_PL_COPY_IMM
#define AASR_PL_COPY_IMM 1
#endif
      constexpr int kRounds = (kChunks + NW - 1) / NW;
      if (AASR_PL_COPY_IMM && kRounds <= 4) {
        const int w = __builtin_amdgcn_readfirstlane(wave);
        const int c0 = w * kRounds;   // pieces c0 .. c0 + kRounds - 1 (the last waves may run past the tile: guarded)
        const unsigned long long sb = (unsigned long long)(uintptr_t)gbase + (unsigned long long)c0 * 1024ull;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb >> 32));
        const unsigned long long sbase = (unsigned long long)lo | ((unsigned long long)hi << 32);
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lbase + (unsigned)c0 * 1024u));
        const int cnt = kChunks - c0 < kRounds ? kChunks - c0 : kRounds;   // wave-uniform
        // (one statement per count: M0 must hold between the instructions)
        if (cnt >= 4)
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                       "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
                       : : "s"(dst), "v"(lane_off16), "s"(sbase) : "memory", "m0");
        else if (cnt == 3)
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                       "global_load_lds_dwordx4 %1, %2 offset:2048"
                       : : "s"(dst), "v"(lane_off16), "s"(sbase) : "memory", "m0");
        else if (cnt == 2)
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                       : : "s"(dst), "v"(lane_off16), "s"(sbase) : "memory", "m0");
        else if (cnt == 1)
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(lane_off16), "s"(sbase) : "memory", "m0");
      } else {
#pragma unroll
      for (int k = 0; k < (kChunks + NW - 1) / NW; k++) {
        const int c = __builtin_amdgcn_readfirstlane(wave) + k * NW;   // wave-uniform
        if (c < kChunks) {
          const unsigned long long sb = (unsigned long long)(uintptr_t)gbase + (unsigned long long)c * 1024ull;
          const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb);
          const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb >> 32));
          const unsigned long long sbase = (unsigned long long)lo | ((unsigned long long)hi << 32);
          const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lbase + (unsigned)c * 1024u));
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(lane_off16), "s"(sbase) : "memory", "m0");
        }
      }
      }
    } else {
      issue_tile_copy_raw((const float *)apack + (size_t)tile * kTileFloats, abuf0 + b * kTileFloats, kTileFloats, wave, lane, NW);
    }
    if (MAPPED && wave == NW - 1 && lane < 16) {
      const int32_t *src = sid + tile * 16 + lane;
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(emap + b * 16));
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" : : "s"(dst), "v"(src) : "memory", "m0");
    }
  };
  // 8-wave form: waves 4-7 pass the tile's barrier in the MIDDLE of their H0 instead of at the end of H1, so they run
  // three quarters of a tile behind waves 0-3 -- the two waves of a SIMD then never sit in their close logic (or at
  // the barrier) at the same time, one of them always has MFMAs to issue.  Three tile buffers make the lag legal: the
  // copy of tile t + 2 is issued by every wave right behind its barrier t (all waves are past tile t - 1 there) and
  // has landed at barrier t + 1, before the lagging group's first read of it.
#ifdef AASR_PL_NOLAG
  const int group = 0;
#else
  const int group = WIDE ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;
#endif
  auto matrix_prio = [&](int phase) {
    if (AASR_PL_PRIO_SCHEME >= 1 && WIDE) {
      const bool hi = AASR_PL_PRIO_SCHEME == 1 ? ((group == 1) == (phase == 1)) : ((group == 1) != (phase == 1));
      if (hi) __builtin_amdgcn_s_setprio(AASR_PL_PRIO_HI);
      else __builtin_amdgcn_s_setprio(AASR_PL_PRIO_LO);
    } else if (AASR_PL_PRIO > 0) {
      __builtin_amdgcn_s_setprio(AASR_PL_PRIO);
    }
  };
  // slab of H0 in front of which the lagging group's barrier sits: early in the phase, so that the lagging waves run
  // nearly a whole tile behind (configs[2], NK16 = 5, ms of the scoring stage on one box: slab 0 8.45, slab 1 8.43,
  // slab 2 -- the middle, rounds 3's choice -- 8.51, slab 3 8.61; with the issue priority of the matrix phases, below:
  // slab 0 8.47, slab 1 8.55, slab 2 8.53 -- the barrier in front of the phase)
#ifdef AASR_PL_JB
  constexpr int JB = AASR_PL_JB < NK16 ? AASR_PL_JB : 0;
#else
  constexpr int JB = 0;
#endif
  const int n = lane & 31;
  const int h = lane >> 5;  // K half of a slab held by this lane AND its row track
  const int64_t f0 = (int64_t)blk * (NW * FRAMES_PER_WAVE) + wave * FRAMES_PER_WAVE;

  // The first two tiles are requested before anything else: they land while the frame operand is being built.
  const int64_t t_begin = split_row[4 * cut];
  const int64_t t_end = split_row[4 * cut + 4];
  if (t_begin < t_end) issue_tile(t_begin, 0);
  if (t_begin + 1 < t_end) issue_tile(t_begin + 1, 1);
  // pivot groups (multi-pivot layouts, gmm.h TrackLayout::n_pg): a row cut lies inside ONE group -- its rows are expanded
  // around that group's pivot, so the workgroup takes that group's image of the frame operand, and the group's columns end
  // at its own limit (its last line goes out partly filled, the next group starts on a whole line)
  int pgi = 0;
  if (pg.colend) {
    pgi = split_row[4 * cut + 3];
    if (!PGF) fop += (size_t)pgi * pg.fop_stride;
    S = pg.colend[pgi];
  }

  // ---- frame operand: lane (n, h) holds k = 16*j + 8*h + i, i < 8, of slab j -- split into its terms ONCE per launch by
  // k_frame_operand (below the kernel) and fetched here with 16-byte loads, 64 lanes x 16 B contiguous per instruction.
  // Built in place (one 4-byte load per K slot at a lane-dependent address, ~2 000 instructions) it cost every workgroup
  // ~20 us -- six tiles' time in front of every row cut, paid R times per frame.
  u32x4 bq[NK16][NS][2];
  if constexpr (PGF && NS == 2) {
    // multi-pivot layouts: the group's image is formed here, around the group's pivot (fop_unit_f16 = k_frame_operand's
    // arithmetic: the same bits), instead of being fetched -- one image per group and launch through HBM cost more than
    // the ~800 instructions a row cut pays for it
    const float *pv = pg.pivots + (size_t)pgi * dim;
    const float *tab = pg.tabs + (size_t)pgi * (3 * 8 * NK16);
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
      int64_t f = f0 + nb * 32 + n;
      if (f > F - 1) f = F - 1;
      const float *xr = frames + f * dim;
#pragma unroll
      for (int j = 0; j < NK16; j++) {
        unsigned w1[4], w2[4];
        fop_unit_f16(xr, dim, pv, tab, 8 * NK16, j, h, pg.sc, w1, w2);
        bq[j][0][nb] = u32x4{w1[0], w1[1], w1[2], w1[3]};
        bq[j][1][nb] = u32x4{w2[0], w2[1], w2[2], w2[3]};
      }
    }
  } else {
    const u32x4 *bw = fop + ((size_t)blk * NW + wave) * (NK16 * NS * 2 * 64) + lane;
#pragma unroll
    for (int j = 0; j < NK16; j++)
#pragma unroll
      for (int sp = 0; sp < NS; sp++)
#pragma unroll
        for (int nb = 0; nb < 2; nb++) bq[j][sp][nb] = bw[((j * NS + sp) * 2 + nb) * 64];
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  PL_TRACE(6);
  float s0 = 0.0f, s1 = 0.0f;
  int closes = split_row[4 * cut + 1 + (GROUPED ? 0 : h)];
  // HYB: the next state of this lane's track that has outlier components, and its two values
  int hyb_st = 0x7fffffff;
  float hyb_v0 = LOG_TINY_F, hyb_v1 = LOG_TINY_F;
  uint32_t hyb_e_next = 0xffffu;   // the table entry of the state AFTER hyb_st (requested together with hyb_st's values)
  // ... whose values are requested BEHIND the next tile barrier, not where hyb_st is consumed: the barrier waits for every
  // outstanding vector-memory operation of the wave (the tile copy's), and a request issued in the close logic in front
  // of it made all eight waves wait for its latency (+13 % with such a state in every tenth column)
  uint32_t hyb_pend = 0xffffu;
  auto hyb_issue = [&](uint32_t e) {   // e: table entry of the state to take next (0xffff in the low half: none)
    hyb_st = 0x7fffffff;
    hyb_e_next = 0xffffu;
    if ((e & 0xffffu) != 0xffffu) {
      hyb_st = (int)(e & 0xffffu);
      const float *pr = pg.hyb_part + (int64_t)(e >> 16) * pg.hyb_pitch;
      const int64_t fa = f0 + n < F ? f0 + n : F - 1, fb = f0 + 32 + n < F ? f0 + 32 + n : F - 1;   // (never stored past F)
      hyb_v0 = pr[fa];
      hyb_v1 = pr[fb];
      if (hyb_st + 2 < (int)S) hyb_e_next = pg.hyb_tab[hyb_st + 2];
    }
  };
  if constexpr (HYB) {
    const int from = 2 * closes + h;   // the first state of this lane's track (parity h) in this row cut
    hyb_issue(from < (int)S ? pg.hyb_tab[from] : 0xffffu);
  }
  const int32_t *my_sid = sid + h * sid_stride;
  int next_sid = GROUPED ? 0 : my_sid[closes];
  float *orow0 = out + (f0 + n) * pitch;  // pitch: row stride of `out` in floats (>= S)
  float *orow1 = out + (f0 + 32 + n) * pitch;
  const bool ok0 = f0 + n < F, ok1 = f0 + 32 + n < F;
  const float floor_val = CL ? cl.floor_val : LOG_TINY_F;
  const unsigned long long *mrow =
      CL ? cl.maskrow + (size_t)(f0 >> 6) * cl.rows_padded + lane : nullptr;

  // a staged group of `cnt` (<= OG) columns from s_base on goes out: whole 16-byte pieces, 128 (64) bytes per frame row
  auto flush_group = [&](const int64_t s_base, const int cnt) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (OG == 32 && cnt == OG && f0 + FRAMES_PER_WAVE <= F) {
      // 8 lanes x 16 B cover the 32-state group; 8 frame rows per instruction
      const int k4 = lane & 7, r8 = lane >> 3;
      float *op = out + (f0 + r8) * pitch + s_base + 4 * k4;
      const float *ip = ost + r8 * kOS + 4 * k4;  // stride 34: 8-byte aligned
#pragma unroll
      for (int i = 0; i < FRAMES_PER_WAVE / 8; i++) {
        const f32x2 lo = *(const f32x2 *)(ip + i * 8 * kOS);
        const f32x2 hi = *(const f32x2 *)(ip + i * 8 * kOS + 2);
        const f32x4 v = {lo[0], lo[1], hi[0], hi[1]};
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        *(f32x4u *)(op + (int64_t)i * 8 * pitch) = v;
      }
    } else if (cnt == OG && f0 + FRAMES_PER_WAVE <= F) {
      // 4 lanes x 16 B cover the 16-state group; 16 frame rows per instruction
      const int k4 = lane & 3, r16 = lane >> 2;
      float *op = out + (f0 + r16) * pitch + s_base + 4 * k4;
      const float *ip = ost + r16 * kOS + 4 * k4;
#pragma unroll
      for (int i = 0; i < FRAMES_PER_WAVE / 16; i++) {
        const f32x4 v = *(const f32x4 *)(ip + i * 16 * kOS);
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        *(f32x4u *)(op + (int64_t)i * 16 * pitch) = v;
      }
    } else {
      constexpr int RPI = 64 / OG;
      const int k = lane & (OG - 1);
#pragma unroll 4
      for (int i = 0; i < FRAMES_PER_WAVE / RPI; i++) {
        const int row = i * RPI + lane / OG;
        const float v = ost[row * kOS + k];
        if (k < cnt && f0 + row < F) out[(f0 + row) * pitch + s_base + k] = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  // close logic of one 32-row block: P[nb][q] = this lane's sum of 2^x over quad q for frame block nb
  // e0 / e1 (MAPPED): the pair entries of the block's four quad positions on track 0 / 1 (wave-uniform: scalar registers)
  auto commit = [&](const float (&P)[2][4], unsigned nib, const int (&e0)[4], const int (&e1)[4]) {
    if (AASR_DBG(128)) {   // ablation: no close logic
      asm volatile("" ::"v"(P[0][0]), "v"(P[0][1]), "v"(P[0][2]), "v"(P[0][3]), "v"(P[1][0]), "v"(P[1][1]), "v"(P[1][2]), "v"(P[1][3]));
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      s0 += P[0][q];
      s1 += P[1][q];
      if ((nib >> q) & 1) {
        float l0 = fmaf(__builtin_amdgcn_logf(s0), LN2_F, -ref_ln);
        float l1 = fmaf(__builtin_amdgcn_logf(s1), LN2_F, -ref_ln);
        l0 = fmaxf(l0, floor_val);
        l1 = fmaxf(l1, floor_val);
        s0 = 0.0f;
        s1 = 0.0f;
        closes++;
        if (!GROUPED) {
          if (ok0) orow0[next_sid] = l0;
          if (ok1) orow1[next_sid] = l1;
          next_sid = my_sid[closes];
        } else if (MAPPED) {
          const int ent = h ? e1[q] : e0[q];
          const int col = ent & kMapCol;
          const int slot = (ent & kMapEmpty) ? OG : (col & (OG - 1));   // an empty track stages into the spare slot
          ost[n * kOS + slot] = l0;
          ost[(32 + n) * kOS + slot] = l1;
          if (ent & (OG == 32 ? kMapFlush32 : kMapFlush16)) {   // the same on both tracks: wave-uniform
            const int64_t s_base = col & ~(OG - 1);
            flush_group(s_base, (int)(S - s_base < OG ? S - s_base : OG));
          }
        } else {
          const int pairs_closed = closes;
          if constexpr (HYB) {
            const int stc = 2 * (pairs_closed - 1) + h;
            if ((hyb_pend & 0xffffu) != 0xffffu && stc == (int)(hyb_pend & 0xffffu)) {   // (closes before the barrier came)
              hyb_issue(hyb_pend);
              hyb_pend = 0xffffu;
            }
            if (stc == hyb_st) {
              // k_outlier_merge's arithmetic: out = log(exp(out) + exp(part)); a part AT the floor holds nothing
              l0 = merge_floored_shares(l0, hyb_v0 > LOG_TINY_F ? hyb_v0 + pg.hyb_bias : hyb_v0);
              l1 = merge_floored_shares(l1, hyb_v1 > LOG_TINY_F ? hyb_v1 + pg.hyb_bias : hyb_v1);
              hyb_pend = hyb_e_next;
              hyb_st = 0x7fffffff;
            }
          }
          const int slot = ((2 * (pairs_closed - 1)) & (OG - 1)) + h;
          ost[n * kOS + slot] = l0;
          ost[(32 + n) * kOS + slot] = l1;
          const int64_t closed = 2 * (int64_t)pairs_closed < S ? 2 * (int64_t)pairs_closed : S;
          if (((2 * pairs_closed) & (OG - 1)) == 0 || 2 * (int64_t)pairs_closed >= S) {
            const int64_t s_base = ((closed - 1) / OG) * OG;
            flush_group(s_base, (int)(closed - s_base));
          }
        }
      }
    }
  };

  // element `k` (0..31) of a block's exponentials: frame block nb = k / 16, quad q, element e; the quad's four
  // values are summed pairwise as the kernel above does, (x0 + x1) + (x2 + x3)
  float t0 = 0.0f, t1 = 0.0f;
  auto epi_step = [&](int k, int mb, const f32x16 &c0, const f32x16 &c1, unsigned long long bits, float (&P)[2][4]) {
    const int nb = k >> 4, q = (k >> 2) & 3, e = k & 3;
    float v = nb ? c1[4 * q + e] : c0[4 * q + e];
    if (CL) v = mask_select(v, bits, 32 * nb + 8 * q + 4 * mb + e);  // k_cluster_expand's bit layout
    // pinned where it is written: left as a builtin the compiler sinks all 32 exponentials of a phase into the
    // close logic that consumes P, i.e. out from under the matrix stream (s_nop: a VALU read of a transcendental's
    // result needs one wait state, and the hazard recogniser does not look inside assembly)
    float x;
    if (AASR_DBG(64)) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(v));   // ablation: no transcendentals
    else asm volatile("v_exp_f32 %0, %1\n\ts_nop 0" : "=v"(x) : "v"(v));
    // the pair sums ride in the stream as well (left to the compiler they gather behind the phase's last MFMA).
    // (Round 4: the additions run one element behind the exponentials, so that no instruction reads a transcendental's
    // result right behind it and the s_nop can go -- measured 1 % SLOWER, 8.60 against 8.51 ms on configs[2]; kept as is.)
    if (e == 0) t0 = x;
    else if (e == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(t0) : "v"(t0), "v"(x));
    else if (e == 2) t1 = x;
    else {
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(t1), "v"(x));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(P[nb][q]) : "v"(t0), "v"(t1));
    }
  };

  // one phase: the MFMAs of 32-row block MB of the tile in `acur` into (n0, n1), carrying the exponentials of the
  // other block's accumulators (o0, o1, selection bits obits) into P
  u32x4 afr[2][NS];  // A fragments [register set][split] of the block being accumulated
  auto load_frags = [&](const float *tile, int j, int mb, int set) {
    const u32x4 *afrag = (const u32x4 *)tile + lane;  // [slab][split][mb][64 lanes]
#pragma unroll
    for (int sp = NS - 1; sp >= 0; sp--) afr[set][sp] = afrag[((j * NS + sp) * 2 + mb) * 64];
  };

  int lane_zero = 0;
  asm volatile("" : "+v"(lane_zero));   // a zero the compiler cannot see through
  f32x16 cA0 = {0}, cA1 = {0}, cB0 = {0}, cB1 = {0};
  unsigned long long bits_cur = 0, bits_prev = 0;
  unsigned mask_cur = 0, mask_prev = 0;
  unsigned mask_v = 0;
  if (t_begin < t_end) {
    mask_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)close_mask[t_begin]);
    if (CL) bits_cur = mrow[(size_t)t_begin * TILE_ROWS];
    load_frags(abuf0, 0, 0, 0);
  }
  int bi = 0;
  int ep0[4] = {0, 0, 0, 0}, ep1[4] = {0, 0, 0, 0};   // MAPPED: entries of block 1 of the previous tile
  for (int64_t t = t_begin; t < t_end; t++) {
    float *acur = abuf0 + bi * kTileFloats;
    const int bn = bi + 1 < NBUF ? bi + 1 : 0, bnn = bn + 1 < NBUF ? bn + 1 : 0;
    float *anext = abuf0 + bn * kTileFloats;
    // tile t + 2 goes where tile t - 1 was (three buffers), or into tile t's own buffer when every wave is done
    // with it at the barrier (two buffers, no lagging group)
    const int bcur = bi;
    bi = bn;
    // barrier t of this wave: its share of tile t + 1 has landed, and every wave is past tile t - 1
    auto tile_barrier = [&]() {
#ifdef AASR_PL_TRACE
      const unsigned long long tb0 = __builtin_readcyclecounter();
#endif
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(mask_v) : : "memory");
#ifdef AASR_PL_TRACE
      const unsigned long long tb1 = __builtin_readcyclecounter();
      tr_vm += tb1 - tb0;
#endif
      if (!AASR_DBG(16)) __builtin_amdgcn_s_barrier();
#ifdef AASR_PL_TRACE
      const unsigned long long tb2 = __builtin_readcyclecounter();
      tr_acc[5] += tb2 - tb1;   // (the s_barrier itself; the tile count moves to the host side)
#endif
      if (t + 2 < t_end) issue_tile(t + 2, bnn);
      if constexpr (HYB) {
        if ((hyb_pend & 0xffffu) != 0xffffu) {
          hyb_issue(hyb_pend);
          hyb_pend = 0xffffu;
        }
      }
#ifdef AASR_PL_TRACE
      tr_bar += __builtin_readcyclecounter() - tb0;
#endif
    };
    // close bits and selection bits of tile t+1: vector loads waited for by the vmcnt(0) in front of the barrier (an
    // aligned 32-bit word: the array has a spare element).  It has to stay a VECTOR load -- as a scalar load it would turn
    // every LDS wait of the stream into lgkmcnt(0) -- and it has to stay a load the COMPILER knows: the first version
    // issued it through inline assembly, and the compiler, for which the result was ready at the asm statement, copied
    // the register before the value had landed (one wave group's close bits were garbage in ~1 workgroup of 6 000 per
    // launch, found by the 10^6-frame test).  The opaque zero keeps the address a vector value.
    mask_v = ((const uint32_t *)close_mask)[((t + 1) >> 1) + lane_zero];
    unsigned long long bits_next = 0;
    if (CL && t + 1 < t_end) bits_next = mrow[(size_t)(t + 1) * TILE_ROWS];
    // MAPPED: this tile's pair entries (this lane's track) out of the tile buffer's side table, both blocks now -- the
    // buffer may be handed to tile t + 2 at this tile's barrier, in front of the commit of block 0
    // (one word per lane, then lane reads: the entries are wave-uniform and live in scalar registers)
    int en00[4] = {0, 0, 0, 0}, en01[4] = {0, 0, 0, 0}, en10[4] = {0, 0, 0, 0}, en11[4] = {0, 0, 0, 0};   // [block][track]
    if (MAPPED) {
      const int ev = emap[bcur * 16 + (lane & 15)];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        en00[q] = __builtin_amdgcn_readlane(ev, q);
        en10[q] = __builtin_amdgcn_readlane(ev, 4 + q);
        en01[q] = __builtin_amdgcn_readlane(ev, 8 + q);
        en11[q] = __builtin_amdgcn_readlane(ev, 12 + q);
      }
    }

    float P[2][4];
    // s_setprio: while a wave is in a matrix phase the SIMD's issue arbiter prefers it to the other wave's close logic
    // (vector, LDS and store instructions), so its matrix instructions do not queue behind them: -0.9 % on configs[2]
    // (8.49 -> 8.41 ms same box, priority 1 and 3 alike)
    PL_TRACE(4);   // (what ran since the end of the previous tile's H1: its barrier excluded below)
    matrix_prio(0);
    // ---------------- H0: block 0 of tile t  ||  exponentials of block 1 of tile t-1
    {
      int mi = 0;
#pragma unroll
      for (int j = 0; j < NK16; j++) {
        const int cur = j & 1;
        if (WIDE && j == JB) {
          if (group == 1) tile_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j + 1 < NK16) load_frags(acur, j + 1, 0, cur ^ 1);
        else load_frags(acur, 0, 1, cur ^ 1);     // slab 0 of block 1, for H1
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int grp = 0; grp < NS; grp++) {
          const int sp = NS - 1 - grp;
#pragma unroll
          for (int c = 0; c <= grp; c++) {
            const int sb = grp - c;
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
              if (j == 0 && grp == 0 && c == 0) {
                const f32x16 z = {0};
                if (nb == 0) cA0 = mfma_split<NS>(afr[cur][sp], bq[j][sb][0], z);
                else cA1 = mfma_split<NS>(afr[cur][sp], bq[j][sb][1], z);
              } else {
                if (nb == 0) cA0 = mfma_split<NS>(afr[cur][sp], bq[j][sb][0], cA0);
                else cA1 = mfma_split<NS>(afr[cur][sp], bq[j][sb][1], cA1);
              }
#pragma unroll
              for (int k = mi * 32 / MPH; k < (mi + 1) * 32 / MPH; k++) epi_step(k, 1, cB0, cB1, bits_prev, P);
              mi++;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
    }
    if (AASR_PL_PRIO > 0) __builtin_amdgcn_s_setprio(0);
    PL_TRACE(0);
    if (t > t_begin) commit(P, (GROUPED ? mask_prev : (h ? mask_prev >> 8 : mask_prev)) >> 4 & 0xfu, ep0, ep1);
    PL_TRACE(1);
    matrix_prio(1);
    // ---------------- H1: block 1 of tile t  ||  exponentials of block 0 of tile t
    {
      int mi = 0;
      constexpr int set0 = NK16 & 1;   // the register set H0 left slab 0 of block 1 in
#pragma unroll
      for (int j = 0; j < NK16; j++) {
        const int cur = (j + set0) & 1;
        if (j + 1 < NK16) {
          load_frags(acur, j + 1, 1, cur ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int grp = 0; grp < NS; grp++) {
          const int sp = NS - 1 - grp;
#pragma unroll
          for (int c = 0; c <= grp; c++) {
            const int sb = grp - c;
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
              if (j == 0 && grp == 0 && c == 0) {
                const f32x16 z = {0};
                if (nb == 0) cB0 = mfma_split<NS>(afr[cur][sp], bq[j][sb][0], z);
                else cB1 = mfma_split<NS>(afr[cur][sp], bq[j][sb][1], z);
              } else {
                if (nb == 0) cB0 = mfma_split<NS>(afr[cur][sp], bq[j][sb][0], cB0);
                else cB1 = mfma_split<NS>(afr[cur][sp], bq[j][sb][1], cB1);
              }
#pragma unroll
              for (int k = mi * 32 / MPH; k < (mi + 1) * 32 / MPH; k++) epi_step(k, 0, cA0, cA1, bits_cur, P);
              mi++;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
    }
    if (AASR_PL_PRIO > 0) __builtin_amdgcn_s_setprio(0);
    PL_TRACE(2);
#ifdef AASR_PL_TRACE
    tr_tiles++;
#endif
    // end of tile: the leading group's barrier
    if (!WIDE || group == 0) tile_barrier();
    else asm volatile("" : "+v"(mask_v));
#ifdef AASR_PL_TRACE
    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
    if (t + 1 < t_end) {
      load_frags(anext, 0, 0, 0);   // slab 0 of the next tile's block 0: in flight during the close logic
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef AASR_PL_TRACE
    const unsigned long long ts1 = __builtin_readcyclecounter();
    tr_sub[0] += ts1 - ts0;   // fragment prefetch (issue)
#endif
    commit(P, (GROUPED ? mask_cur : (h ? mask_cur >> 8 : mask_cur)) & 0xfu, en00, en01);
#ifdef AASR_PL_TRACE
    tr_sub[1] += __builtin_readcyclecounter() - ts1;   // close logic of block 0
#endif
#pragma unroll
    for (int q = 0; q < 4; q++) {
      ep0[q] = en10[q];
      ep1[q] = en11[q];
    }
    mask_prev = mask_cur;
    bits_prev = bits_cur;
    {
      const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)mask_v);
      mask_cur = ((t + 1) & 1) ? w >> 16 : w & 0xffffu;
    }
    bits_cur = bits_next;
  }
  // drain: block 1 of the last tile
  if (t_begin < t_end) {
    float P[2][4];
#pragma unroll
    for (int k = 0; k < 32; k++) epi_step(k, 1, cB0, cB1, bits_prev, P);
    commit(P, (GROUPED ? mask_prev : (h ? mask_prev >> 8 : mask_prev)) >> 4 & 0xfu, ep0, ep1);
  }
#ifdef AASR_PL_TRACE
  if ((int)blockIdx.x == AASR_PL_TRACE_BLOCK && lane == 0) {
    PL_TRACE(4);
    // the lagging group's barrier sits inside H0, the leading group's behind H1 (inside interval 4)
    if (WIDE && group == 1) tr_acc[0] -= tr_bar;
    else tr_acc[4] -= tr_bar;
    tr_acc[3] = tr_bar;
    tr_acc[6] = tr_vm;   // (of interval 3: the wait for the wave's own vector-memory operations, tile copy share and stores)
    tr_acc[7] = tr_prev - tr_t0;
    for (int k = 0; k < 8; k++) g_pl_trace[wave & 7][k] = tr_acc[k];
    g_pl_trace[wave & 7][8] = tr_tiles;
    g_pl_trace[wave & 7][9] = tr_sub[0];
    g_pl_trace[wave & 7][10] = tr_sub[1];
  }
#endif
}

// ---------------------------------------------------------------------------
// Frame operand of the split-term kernels, formed once per launch: for every block of 64 frames the K x 64 operand in
// the register layout of k_gmm_diag_score_pl -- [block][slab j][term][frame half nb][lane (n, h)] x 8 halves, K slot
// k = 16 j + 8 h + i -- so that a wave's prologue is NK16 * NS * 2 coalesced 16-byte loads.  Per value the arithmetic of
// the former in-kernel prologue: (x - pivot), the dimension's clamp and the column's power-of-two scale (f16x2), the
// square for odd k, 1 in the constant's slot(s), then the two fp16 / three bf16 terms.  Frames past the end repeat the
// last one (their results are never stored).  One thread per (frame, slab, K half).
// ---------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void k_frame_operand(const float *__restrict__ frames, int64_t F, int dim,
                                                       const float *__restrict__ pivot, const float *__restrict__ f16tab,
                                                       int nk16, u32x4 *__restrict__ out, int64_t n_units, int n_pg,
                                                       int64_t pg_stride, int sc) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(tid & 63);
  const int64_t unit = tid >> 6;   // (block of 64 frames, frame half, slab)
  if (unit >= n_units) return;
  const int j = (int)(unit % nk16);
  const int nb = (int)((unit / nk16) & 1);
  const int64_t blk = unit / (2 * nk16);
  const int n = lane & 31, h = lane >> 5;
  const int KH = 8 * nk16;
  int64_t f = blk * 64 + nb * 32 + n;
  if (f > F - 1) f = F - 1;
  const float *xr = frames + f * dim;
  // the thread's 8 K slots are four (linear, quadratic) pairs: pair u = k / 2 is the constant's two slots for u = 0 and
  // dimension u - 1 otherwise (the K order at the top of the split-term kernels), so a thread handles 4 consecutive
  // dimensions d0 .. d0 + 3 (three and the constant in the first slab's first half).  Where they all exist the frame
  // components, pivots and clamps come as one 16-byte load each, the column scales as two (rows are 4-byte aligned; the
  // tables' loads are the same for every lane of a K half)
  // Slab-constant layout (sc, TrackLayout::sc): slab j = its constant's two slots, then dimensions 7 j .. 7 j + 6 -- the
  // first K half holds the constant and three dimensions, the second four.
  const int k0 = 16 * j + 8 * h;
  const int d0 = sc ? 7 * j + (h ? 3 : -1) : (k0 >> 1) - 1;
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  const bool whole = !sc && d0 >= 0 && d0 + 4 <= dim;   // uniform per K half
  float x[4];
  if (whole) {
    const f32x4u a = *(const f32x4u *)(xr + d0);
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = a[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = xr[d0 + i >= 0 && d0 + i < dim ? d0 + i : 0];
  }
  u32x4 *o = out + ((size_t)(blk * nk16 + j) * NS * 2 + nb) * 64 + lane;   // + term * 2 * 64
  // one image per pivot group (multi-pivot layouts; n_pg = 1 otherwise): the frame is read once
  for (int g = 0; g < n_pg; g++, pivot += dim, f16tab += (NS == 2 ? 3 * KH : 0), o += pg_stride) {
    float v[8];
    if (whole) {
      const f32x4u b = *(const f32x4u *)(pivot + d0);
      f32x4u c = {0, 0, 0, 0}, e0 = {1, 1, 1, 1}, e1 = {1, 1, 1, 1};
      if (NS == 2) {
        c = *(const f32x4u *)(f16tab + 2 * KH + d0);
        e0 = *(const f32x4u *)(f16tab + k0);
        e1 = *(const f32x4u *)(f16tab + k0 + 4);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float xc = x[i] - b[i];
        float xq = xc;
        if (NS == 2) xq = fminf(fmaxf(xc, -c[i]), c[i]);   // fp16 range: the dimension's clamp (pack_f16x2)
        float lin = xq, quad = xq * xq;
        if (NS == 2) {   // the columns' power-of-two scales (the rows carry their inverses): exact
          lin *= i < 2 ? e0[2 * i] : e1[2 * i - 4];
          quad *= i < 2 ? e0[2 * i + 1] : e1[2 * i - 3];
        }
        v[2 * i] = lin;
        v[2 * i + 1] = quad;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int k = k0 + i;
        // (the constant's slots: the first two of the plain layout, the first two of every slab of the slab-constant one)
        const int d = (sc ? (h == 0 && i < 2) : k < 2) ? -1 : d0 + (i >> 1);
        const int dc = d >= 0 && d < dim ? d : 0;
        const float xc = x[i >> 1] - pivot[dc];
        float xq = xc;
        if (NS == 2) {  // fp16 range: the dimension's clamp (see the f16x2 note above and pack_f16x2)
          const float lim = f16tab[2 * KH + dc];
          xq = fminf(fmaxf(xc, -lim), lim);
        }
        float val = (k & 1) ? xq * xq : xq;
        if (d < 0) val = (k == 0 || NS == 2) ? 1.0f : 0.0f;   // the constant and (f16x2) its remainder
        else if (d >= dim) val = 0.0f;
        if (NS == 2) val *= f16tab[k];   // the column's power-of-two scale (the rows carry its inverse): exact
        v[i] = val;
      }
    }
    if constexpr (NS == 3) {
      unsigned w1[4], w2[4], w3[4];
#pragma unroll
      for (int i = 0; i < 4; i++) split3_pair(v[2 * i], v[2 * i + 1], w1[i], w2[i], w3[i]);
      o[0] = u32x4{w1[0], w1[1], w1[2], w1[3]};
      o[2 * 64] = u32x4{w2[0], w2[1], w2[2], w2[3]};
      o[4 * 64] = u32x4{w3[0], w3[1], w3[2], w3[3]};
    } else {
      // (the unit's arithmetic lives in fop_unit_f16, shared with the kernels that form their operand themselves; `v`
      // above is the three-term form's)
      unsigned w1[4], w2[4];
      fop_unit_f16(xr, dim, pivot, f16tab, KH, j, h, sc, w1, w2);
      o[0] = u32x4{w1[0], w1[1], w1[2], w1[3]};
      o[2 * 64] = u32x4{w2[0], w2[1], w2[2], w2[3]};
    }
  }
}

// frame operand of `blocks64` blocks of 64 frames into the handle's scratch (grown as needed)
template <int NS>
static const u32x4 *frame_operand(const aasr_gmm *g, const TrackLayout &L, const float *d_frames, int64_t F,
                                  int64_t blocks64, hipStream_t stream, int64_t *pg_stride) {
  const size_t per_block = (size_t)L.nk16 * NS * 2 * 64;   // u32x4 per 64 frames
  const int n_pg = L.n_pg > 1 ? L.n_pg : 1;
  const size_t image = (size_t)blocks64 * per_block;       // u32x4 per pivot group
  if (image * n_pg * 4 > g->fop_scratch.n) {
    AASR_HIP(hipDeviceSynchronize());   // growing frees the old buffer
    g->fop_scratch.ensure(image * n_pg * 4);
  }
  const int64_t n_units = blocks64 * 2 * L.nk16;
  hipLaunchKernelGGL(k_frame_operand<NS>, dim3((unsigned)((n_units * 64 + 255) / 256)), dim3(256), 0, stream, d_frames, F,
                     g->dim, L.n_pg > 1 ? L.pg_pivot.p : g->d_pivot.p,
                     NS == 2 ? (L.n_pg > 1 ? L.pg_tab.p : L.f16tab.p) : nullptr, L.nk16, (u32x4 *)g->fop_scratch.p, n_units,
                     n_pg, (int64_t)image, (NS == 2 && L.sc) ? 1 : 0);
  AASR_HIP(hipGetLastError());
  *pg_stride = (int64_t)image;
  return (const u32x4 *)g->fop_scratch.p;
}

// Row cuts of a launch: `blocks` frame blocks x R cuts are dealt to `slots` resident workgroups in rounds; a workgroup
// costs its tiles plus a fixed part (launch, frame operand, first tile's latency, drain), `overhead` in units of one
// tile's time.  Measured on configs[2] (878 blocks of 512 frames, 782 tiles, 256 slots; ms of the scoring kernel at
// R = 2 / 4 / 8 / 16: 8.51 / 8.68 / 8.94 / 9.20): the fixed part was 6.6 tiles with the operand built in the kernel.
// The former rule -- the R whose last round is fullest -- took R = 9 there (8.86 ms).
static int pick_row_cuts(int64_t blocks, double slots, int64_t tiles, int max_splits, double overhead) {
  static const int force_r = AASR_EXPERIMENT_ENV("AASR_SPLITS") ? atoi(AASR_EXPERIMENT_ENV("AASR_SPLITS")) : 0;
  static const double force_c = AASR_EXPERIMENT_ENV("AASR_CUT_OVERHEAD") ? atof(AASR_EXPERIMENT_ENV("AASR_CUT_OVERHEAD")) : -1.0;
  if (force_r >= 1 && force_r <= max_splits) return force_r;
  if (force_c >= 0) overhead = force_c;
  int R = 1;
  double best = 1e300;
  for (int r = 1; r <= max_splits; r++) {
    const double cost = std::ceil((double)blocks * r / slots) * ((double)tiles / r + overhead);
    if (cost < best * 0.999) {
      best = cost;
      R = r;
    }
  }
  return R;
}

// Two-level plan: the workgroups of a launch run in rounds of `slots`, and a uniform R leaves the last round partly
// empty (configs[2]: 878 blocks x 2 cuts = 6.86 rounds of 256).  So the frame blocks that fill whole rounds at a coarse
// cut count go first, and the remaining blocks are cut finer so that THEIR last round is nearly full too: configs[2]
// 768 blocks x 2 cuts (6 rounds) + 110 blocks x 16 cuts (6.9 short rounds) instead of 7 long ones.  Same cost model as
// pick_row_cuts; falls back to the uniform plan when that is no better.
static CutPlan pick_cut_plan(int64_t blocks, double slots_d, int64_t tiles, int max_splits, double overhead,
                             const int32_t *splits_base, int min_splits = 1, int split_cap = TRACK_MAX_SPLITS) {
  static const int force_r = AASR_EXPERIMENT_ENV("AASR_SPLITS") ? atoi(AASR_EXPERIMENT_ENV("AASR_SPLITS")) : 0;
  static const double force_c = AASR_EXPERIMENT_ENV("AASR_CUT_OVERHEAD") ? atof(AASR_EXPERIMENT_ENV("AASR_CUT_OVERHEAD")) : -1.0;
  static const int two_level = AASR_EXPERIMENT_ENV("AASR_TWO_LEVEL") ? atoi(AASR_EXPERIMENT_ENV("AASR_TWO_LEVEL")) : 1;
  if (force_c >= 0) overhead = force_c;
  const int64_t slots = (int64_t)slots_d;
  CutPlan best;
  double best_cost = 1e300;
  auto row = [&](int r) { return splits_base + (size_t)(r - 1) * (split_cap + 1) * 4; };
  for (int r1 = min_splits; r1 <= max_splits; r1++) {   // (multi-pivot layouts: every pivot group is at least one cut)
    if (force_r >= min_splits && force_r <= max_splits && r1 != force_r) continue;
    // uniform
    const double uni = std::ceil((double)blocks * r1 / slots_d) * ((double)tiles / r1 + overhead);
    if (uni < best_cost * 0.999) {
      best_cost = uni;
      best = CutPlan();
      best.n_main = (int)(blocks * r1);
      best.blocks_main = (int)blocks;
      best.r_main = r1;
    }
    if (!two_level || (force_r >= 1 && force_r <= max_splits)) continue;
    // whole rounds at r1, the rest at r2
    const int64_t rounds = blocks * r1 / slots;
    if (rounds < 1 || (rounds * slots) % r1 != 0) continue;
    const int64_t bm = rounds * slots / r1;
    const int64_t rem = blocks - bm;
    if (rem <= 0) continue;
    for (int r2 = r1 + 1; r2 <= max_splits; r2++) {
      const double cost = (double)rounds * ((double)tiles / r1 + overhead) +
                          std::ceil((double)rem * r2 / slots_d) * ((double)tiles / r2 + overhead);
      if (cost < best_cost * 0.995) {
        best_cost = cost;
        best.n_main = (int)(bm * r1);
        best.blocks_main = (int)bm;
        best.blocks_rem = (int)rem;
        best.r_main = r1;
        best.r_rem = r2;
        best.split_rem = row(r2);
      }
    }
  }
  return best;
}

template <int NK16, bool GROUPED, bool CL, bool WIDE, int NS>
static void launch_bf16_t(const aasr_gmm *g, const TrackLayout &L, const float *d_frames, int64_t F,
                          float *d_out, hipStream_t stream, const ClusterArgs &cl, int64_t pitch) {
  constexpr int NW = WIDE ? 8 : 4;
  const int64_t blocks = (F + NW * FRAMES_PER_WAVE - 1) / (NW * FRAMES_PER_WAVE);
  const int smem = (WIDE ? 3 : 2) * Bf16Smem<NK16, GROUPED, WIDE, NS>::kTileBytes +
                   NW * Bf16Smem<NK16, GROUPED, WIDE, NS>::kOutFloatsPerWave * 4;
  static const int dbg = AASR_EXPERIMENT_ENV("AASR_DBG") ? atoi(AASR_EXPERIMENT_ENV("AASR_DBG")) : 0;
  static bool attr_set[64] = {false};
  auto kern = k_gmm_diag_score_bf16x3<NK16, GROUPED, CL, WIDE, NS>;
  if (!attr_set[g->device & 63]) {
    AASR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[g->device & 63] = true;
  }
  const int R = pick_row_cuts(blocks, (WIDE ? 1.0 : 2.0) * (g->num_cus > 0 ? g->num_cus : 256),
                              L.rows_padded / TILE_ROWS, L.max_splits, 6.0);
  const int32_t *split_row = L.splits.p + (size_t)(R - 1) * (L.split_cap + 1) * 4;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)R), dim3(NW * 64), smem, stream, d_frames, F,
                     g->dim, g->d_pivot.p, NS == 3 ? L.a16.p : L.a16h.p, split_row, L.close.p, L.sid.p, L.sid_stride,
                     d_out, g->S, pitch, L.ref_ln - (float)g->out_bias_ln, dbg, cl);
  AASR_HIP(hipGetLastError());
}

// The three-term bf16 arithmetic on the pipelined kernel as well, up to 39 dimensions (five slabs: with six the 8-wave
// instance spills).  Round 3 measured it SLOWER there than on the wave-group kernel (33.2 against 32.4 ms per 10^6 frames);
// with the wave groups' priorities crossed per phase (AASR_PL_PRIO_SCHEME) it is the faster one: 31.08 against 31.65 ms,
// two alternating runs on one box.  0: k_gmm_diag_score_bf16x3 for every one-pivot three-term layout.
#ifndef AASR_PL_BF16X3
#define AASR_PL_BF16X3 1
#endif
template <int NK16, bool GROUPED, bool CL, bool WIDE, int NS>
static void launch_pl_t(const aasr_gmm *g, const TrackLayout &L, const float *d_frames, int64_t F,
                        float *d_out, hipStream_t stream, const ClusterArgs &cl, int64_t pitch) {
  constexpr int NW = WIDE ? 8 : 4;
  const int64_t blocks = (F + NW * FRAMES_PER_WAVE - 1) / (NW * FRAMES_PER_WAVE);
  const int smem = PlSmem<NK16, GROUPED, WIDE, NS>::kBytes;
  static const int dbg = AASR_EXPERIMENT_ENV("AASR_DBG") ? atoi(AASR_EXPERIMENT_ENV("AASR_DBG")) : 0;
  static bool attr_set[64] = {false};
  auto kern = k_gmm_diag_score_pl<NK16, GROUPED, CL, WIDE, NS>;
  if (!attr_set[g->device & 63]) {
    AASR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[g->device & 63] = true;
  }
  const int32_t *splits_base = L.splits.p;
  const bool multi = L.n_pg > 1;   // pivot groups: every group at least one cut, its own frame operand
  const int cap = L.split_cap;   // rows of the cut table
  const CutPlan plan = pick_cut_plan(blocks, (WIDE ? 1.0 : 2.0) * (g->num_cus > 0 ? g->num_cus : 256),
                                     L.rows_padded / TILE_ROWS, L.max_splits, 3.0, splits_base, multi ? L.n_pg : 1, cap);
  const int32_t *split_row = splits_base + (size_t)(plan.r_main - 1) * (cap + 1) * 4;
  PivotGroups pg;
  const unsigned n_items = (unsigned)(plan.n_main + (plan.r_rem ? plan.blocks_rem * plan.r_rem : 0));
  if constexpr (GROUPED && NS == 2) {
    // multi-pivot layouts: the workgroups form their group's frame operand themselves (no k_frame_operand launch)
    static const int pgf_env = AASR_EXPERIMENT_ENV("AASR_PGF") ? atoi(AASR_EXPERIMENT_ENV("AASR_PGF")) : 1;   // EXPERIMENT: 0 = images through HBM
    if (multi && pgf_env && L.pg_tab.p) {
      auto kern_pgf = k_gmm_diag_score_pl<NK16, GROUPED, CL, WIDE, NS, false, true>;
      static bool attr_set_pgf[64] = {false};
      if (!attr_set_pgf[g->device & 63]) {
        AASR_HIP(hipFuncSetAttribute((const void *)kern_pgf, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set_pgf[g->device & 63] = true;
      }
      pg.colend = L.pg_colend.p;
      pg.pivots = L.pg_pivot.p;
      pg.tabs = L.pg_tab.p;
      pg.sc = L.sc ? 1 : 0;
      hipLaunchKernelGGL(kern_pgf, dim3(n_items), dim3(NW * 64), smem, stream, d_frames, F,
                         g->dim, g->d_pivot.p, L.a16h.p, split_row, L.close.p, L.sid.p, L.sid_stride,
                         d_out, g->S, pitch, L.ref_ln - (float)g->out_bias_ln, dbg, cl, (const u32x4 *)nullptr, plan, pg);
      AASR_HIP(hipGetLastError());
      return;
    }
  }
  const u32x4 *fop = frame_operand<NS>(g, L, d_frames, F, blocks * NW, stream, &pg.fop_stride);
  if (multi) pg.colend = L.pg_colend.p;
  if constexpr (GROUPED && NS == 2 && !CL) {
    // outlier routing with the merge in the close logic: the launcher has put the outliers' partial sums on the handle
    if (!multi && g->hyb_fuse.part && g->hyb_tab.p) {
      auto kern_hyb = k_gmm_diag_score_pl<NK16, GROUPED, CL, WIDE, NS, false, false, true>;
      static bool attr_set_hyb[64] = {false};
      if (!attr_set_hyb[g->device & 63]) {
        AASR_HIP(hipFuncSetAttribute((const void *)kern_hyb, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set_hyb[g->device & 63] = true;
      }
      pg.hyb_tab = g->hyb_tab.p;
      pg.hyb_part = g->hyb_fuse.part;
      pg.hyb_pitch = g->hyb_fuse.pitch;
      pg.hyb_bias = (float)g->out_bias_ln;
      hipLaunchKernelGGL(kern_hyb, dim3(n_items), dim3(NW * 64), smem, stream, d_frames, F,
                         g->dim, g->d_pivot.p, L.a16h.p, split_row, L.close.p, L.sid.p, L.sid_stride,
                         d_out, g->S, pitch, L.ref_ln - (float)g->out_bias_ln, dbg, cl, fop, plan, pg);
      AASR_HIP(hipGetLastError());
      return;
    }
  }
  hipLaunchKernelGGL(kern, dim3(n_items), dim3(NW * 64), smem, stream, d_frames, F,
                     g->dim, g->d_pivot.p, NS == 3 ? L.a16.p : L.a16h.p, split_row, L.close.p,
                     L.sid.p, L.sid_stride,
                     d_out, g->S, pitch, L.ref_ln - (float)g->out_bias_ln, dbg, cl, fop, plan, pg);
  AASR_HIP(hipGetLastError());
}

// the 8-wave form needs three tile buffers + eight staging areas in 160 KB of LDS
template <int N, int NS>
static constexpr bool wide_ok() {
  if (NS == 2) return PlSmem<N, true, true, NS>::kBytes <= 160 * 1024;
  // (three terms: the wave-group kernel, or -- multi-pivot layouts -- the pipelined one: room for either)
  return 3 * Bf16Smem<N, true, true, NS>::kTileBytes + 8 * Bf16Smem<N, true, true, NS>::kOutFloatsPerWave * 4 <= 160 * 1024 &&
         PlSmem<N, true, true, NS>::kBytes <= 160 * 1024;
}

// NS = 3: three bf16 terms (AASR_PREC_BF16X3) on the wave-group kernel; NS = 2: two fp16 terms (AASR_PREC_F16X2) on
// the software-pipelined kernel
template <int NS>
static bool launch_split(const aasr_gmm *g, const TrackLayout &L, const float *d_frames, int64_t F,
                         float *d_out, hipStream_t stream, const ClusterArgs *cl = nullptr, int64_t pitch = 0) {
  if (pitch <= 0) pitch = g->S;
  if (NS == 3 ? !L.a16.p : !L.a16h.p) return false;
  const bool grouped = L.grouped;
  const ClusterArgs none;
  // AASR_BF16_WIDE=0 selects the 4-wave workgroups
  static const int wide_env = AASR_EXPERIMENT_ENV("AASR_BF16_WIDE") ? atoi(AASR_EXPERIMENT_ENV("AASR_BF16_WIDE")) : -1;
  // small batches (a decoder's per-utterance blocks) fill the chip better with 256-frame workgroups
  const int wide = wide_env >= 0 ? wide_env : (F >= 8192 ? 1 : 0);
  switch (L.nk16) {
#define AASR_LAUNCH(N, GR, CLF, WD, CLA)                                                   \
  do {                                                                                     \
    if constexpr (NS == 2 || (AASR_PL_BF16X3 && N <= 5))                                   \
      launch_pl_t<N, GR, CLF, WD, NS>(g, L, d_frames, F, d_out, stream, CLA, pitch);        \
    else if (L.n_pg > 1) {                                                                  \
      /* three bf16 terms on a multi-pivot layout: the pipelined kernel takes the groups' operand images */ \
      if constexpr (GR) launch_pl_t<N, true, CLF, WD, NS>(g, L, d_frames, F, d_out, stream, CLA, pitch); \
    } else                                                                                 \
      launch_bf16_t<N, GR, CLF, WD, NS>(g, L, d_frames, F, d_out, stream, CLA, pitch);      \
  } while (0)
#define AASR_CASE(N)                                                                       \
  case N:                                                                                  \
    if (cl && NS == 2 && wide && wide_ok<N, NS>()) {                                       \
      /* masked (clustered) runs: the bf16x3 8-wave form with masks needs 254 VGPRs + spills and was */ \
      /* measured slower, so it keeps 4-wave workgroups; the f16x2 kernel has the registers          */ \
      if constexpr (NS == 2) {                                                             \
        if (grouped) AASR_LAUNCH(N, true, true, true, *cl);                                \
        else AASR_LAUNCH(N, false, true, true, *cl);                                       \
      }                                                                                    \
    } else if (cl) {                                                                       \
      if (grouped) AASR_LAUNCH(N, true, true, false, *cl);                                 \
      else AASR_LAUNCH(N, false, true, false, *cl);                                        \
    } else if (wide && wide_ok<N, NS>()) {                                                 \
      if (grouped) AASR_LAUNCH(N, true, false, true, none);                                \
      else AASR_LAUNCH(N, false, false, true, none);                                       \
    } else {                                                                               \
      if (grouped) AASR_LAUNCH(N, true, false, false, none);                               \
      else AASR_LAUNCH(N, false, false, false, none);                                      \
    }                                                                                      \
    return true;
    AASR_CASE(1) AASR_CASE(2) AASR_CASE(3) AASR_CASE(4) AASR_CASE(5) AASR_CASE(6) AASR_CASE(8)
#undef AASR_CASE
#undef AASR_LAUNCH
    default:
      return false;
  }
}

// the split-operand kernel the handle's precision asks for (f16x2 only where the layout is eligible)
static bool launch_bf16(const aasr_gmm *g, const TrackLayout &L, const float *d_frames, int64_t F,
                        float *d_out, hipStream_t stream, const ClusterArgs *cl = nullptr, int64_t pitch = 0) {
  if (g->precision == AASR_PREC_F16X2 && L.a16h.p && launch_split<2>(g, L, d_frames, F, d_out, stream, cl, pitch))
    return true;
  return launch_split<3>(g, L, d_frames, F, d_out, stream, cl, pitch);
}

// Verdicts of the probe, keyed by what it depends on (the model's arrays, pivots, the rows' eligibility, the round and the
// tolerance): a rebuild of the same content -- a second handle on the same files, a transform that is taken off again, a
// sub-model rebuilt for a speaker whose classes did not change -- takes the verdict instead of scoring the probe frames
// again.  Process-wide, bounded (256 entries), under a mutex.
namespace {
struct ProbeCache {
  std::mutex mu;
  std::unordered_map<uint64_t, std::vector<int32_t>> verdicts;   // key -> states rejected in that round
  int64_t runs = 0, hits = 0;
};
ProbeCache &probe_cache() {
  static ProbeCache c;
  return c;
}
inline uint64_t fnv1a(const void *p, size_t n, uint64_t h) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) {
    h ^= b[i];
    h *= 0x100000001b3ull;
  }
  return h;
}
template <typename T>
inline uint64_t fnv_vec(const std::vector<T> &v, uint64_t h) {
  const uint64_t n = v.size();
  h = fnv1a(&n, sizeof n, h);
  return v.empty() ? h : fnv1a(v.data(), v.size() * sizeof(T), h);
}
}  // namespace
// Diagnostic (tests): device runs of the probe and verdicts taken from the cache since the library was loaded
extern "C" void aasr_debug_probe_counts(int64_t *runs, int64_t *hits) {
  ProbeCache &c = probe_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  if (runs) *runs = c.runs;
  if (hits) *hits = c.hits;
}

// ---------------------------------------------------------------------------
// Load-time guard of the two-term fp16 form.  Which states get it is decided by conditioning limits that were set
// from sweeps (gmm.h, KAPPA_LIMIT_F16): a bound in the statistical sense, not a proof.  So every model that got fp16
// rows is probed once when it is built: frames placed on its own Gaussians -- 0.5 to 2.5 sigma out in every dimension,
// and one dimension at a time pushed to +-6 sigma -- are scored by the f16x2 path on the device and in double on the
// host (the formula of aku/Distributions.cc:1040-1062, 2078-2086 over the same components), and a state whose visible
// values differ by more than 5e-5 (half the 1e-4 contract) loses the fp16 rows: it moves to the three-term section of
// the mixed layout (per-state precision routing), the rest of the model keeps the fast form.  The number of probe
// frames is sized so that the host side stays at a few million frame x component pairs (24 ... 192 frames).
// AASR_F16_PROBE=0 switches the guard off.
// ---------------------------------------------------------------------------
void gmm_probe_f16x2(aasr_gmm *g) {
  static const int probe_env = AASR_EXPERIMENT_ENV("AASR_F16_PROBE") ? atoi(AASR_EXPERIMENT_ENV("AASR_F16_PROBE")) : 1;
  static const float probe_tol = getenv("AASR_F16_PROBE_TOL") ? (float)atof(getenv("AASR_F16_PROBE_TOL")) : 5.0e-5f;  // test hook
  g->f16_probe_moved = 0;
  if (!probe_env || !g->dim_parts.empty() || g->class_routing || g->host.factor_path() || g->ill_conditioned) return;
  const HostModel &m = g->host;
  const int D = m.dim;
  const int64_t S = m.S, K = (int64_t)m.mix_idx.size();
  if (K == 0) return;
  const int P = (int)std::max<int64_t>(24, std::min<int64_t>(192, 4000000 / K));
  // per mixture component: constant + log weight (natural log), as pack_rows forms them; components routed to the
  // centred kernel (outliers) are not part of the matrix path's sum
  std::vector<double> cst((size_t)K);
  std::vector<uint8_t> live((size_t)K, 1);
  for (int64_t k = 0; k < K; k++) {
    const int64_t gi = m.mix_idx[(size_t)k];
    if (!g->outlier.empty() && g->outlier[(size_t)gi]) live[(size_t)k] = 0;
    double prod = 1;
    for (int d = 0; d < D; d++) {
      const double v = m.var[(size_t)gi * D + d];
      prod *= v > 0 ? 1 / v : 0;
    }
    cst[(size_t)k] = ((prod > 0) ? std::log(std::sqrt(prod)) : prod) + m.logw((size_t)k);
    if (!std::isfinite(cst[(size_t)k])) live[(size_t)k] = 0;
  }
  for (int round = 0; round < 2; round++) {
    TrackLayout &L0 = g->paired.ok ? g->paired : g->tracks;
    if (!L0.ok) return;
    // (a three-term engine part -- gmm_plan_engine_parts admits states to it beyond the one-pivot forms' limits -- is
    // probed the same way on its own rows)
    const bool pg3 = m.n_pg() > 0 && m.pg_arith == 3;
    const TrackLayout *LF = pg3 ? (L0.a16.p ? &L0 : nullptr) : (L0.a16h.p ? &L0 : nullptr);
    if (!LF) return;
    // probe frames (deterministic): frame i sits on mixture component (i * step) % K
    std::vector<float> fr((size_t)P * D);
    uint64_t st = 0x9e3779b97f4a7c15ull + (uint64_t)round * 77;
    auto uni = [&]() {   // xorshift64*, uniform in (0, 1)
      st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
      return ((st * 0x2545f4914f6cdd1dull) >> 11) * (1.0 / 9007199254740992.0) + 1e-17;
    };
    const int64_t step = std::max<int64_t>(1, K / P) | 1;
    for (int i = 0; i < P; i++) {
      const int64_t k = ((int64_t)i * step + round) % K;
      const int64_t gi = m.mix_idx[(size_t)k];
      static const double amps[4] = {0.5, 1.0, 1.5, 2.5};
      const double amp = i < P * 2 / 3 ? amps[i & 3] : 1.0;
      const int far_d = i < P * 2 / 3 ? -1 : (int)(uni() * D) % D;
      // every other far frame is far along SEVERAL dimensions at once (round 6: where tools/fuzz_fitted.py found the
      // round-5 limits wanting): ~ a sixth of the dimensions at +-3.5 .. 5 sigma, 8-12 sigma in all at 39 dimensions
      const bool multi = far_d >= 0 && ((i >> 1) & 1);
      for (int d = 0; d < D; d++) {
        const double v = m.var[(size_t)gi * D + d];
        const double sd = v > 0 ? std::sqrt(v) : 0.0;
        // a normal deviate from twelve uniforms is plenty here
        double z = -6.0;
        for (int u = 0; u < 12; u++) z += uni();
        double x = m.mean[(size_t)gi * D + d] + amp * sd * z;
        if (d == far_d) x = m.mean[(size_t)gi * D + d] + ((i & 1) ? 6.0 : -6.0) * sd;
        else if (multi && uni() < 1.0 / 6.0) x = m.mean[(size_t)gi * D + d] + (uni() < 0.5 ? -1.0 : 1.0) * (3.5 + 1.5 * uni()) * sd;
        fr[(size_t)i * D + d] = (float)x;
      }
    }
    // the verdict of this round, if the same content has been probed before
    uint64_t key = 0xcbf29ce484222325ull;
    {
      const int64_t hdr[8] = {D, S, K, P, round, pg3 ? 1 : 0, 0, m.pg_arith};
      key = fnv1a(hdr, sizeof hdr, key);
      const double tol_d = (double)probe_tol, bias_d = g->out_bias_ln, lwb = m.logw_bias;
      key = fnv1a(&tol_d, sizeof tol_d, key);
      key = fnv1a(&bias_d, sizeof bias_d, key);
      key = fnv1a(&lwb, sizeof lwb, key);
      key = fnv_vec(m.mean, key);
      key = fnv_vec(m.var, key);
      key = fnv_vec(m.mix_off, key);
      key = fnv_vec(m.mix_idx, key);
      key = fnv_vec(m.mix_w, key);
      key = fnv_vec(m.pg_pivot, key);
      key = fnv_vec(m.pg_begin, key);
      key = fnv_vec(m.pg_real_end, key);
      key = fnv_vec(g->pivot, key);
      key = fnv_vec(g->f16_state_ok, key);
      key = fnv_vec(g->outlier, key);
    }
    std::vector<uint8_t> bad((size_t)S, 0);
    int64_t n_bad = 0;
    bool cached = false;
    {
      ProbeCache &pc = probe_cache();
      std::lock_guard<std::mutex> lk(pc.mu);
      auto it = pc.verdicts.find(key);
      if (it != pc.verdicts.end()) {
        cached = true;
        pc.hits++;
        for (int32_t s2 : it->second)
          if (s2 >= 0 && s2 < S && !bad[(size_t)s2]) { bad[(size_t)s2] = 1; n_bad++; }
      }
    }
    if (!cached) {
    DevBuf<float> d_fr, d_a;
    d_fr.upload(fr.data(), fr.size());
    d_a.alloc((size_t)P * S);
    const int prec = g->precision;
    const bool use = g->use_bf16x3;
    g->precision = AASR_PREC_F16X2;
    g->use_bf16x3 = true;
    const bool ok_a = pg3 ? launch_split<3>(g, *LF, d_fr.p, P, d_a.p, nullptr) : launch_bf16(g, *LF, d_fr.p, P, d_a.p, nullptr);
    g->precision = prec;
    g->use_bf16x3 = use;
    if (!ok_a) return;
    std::vector<float> a((size_t)P * S);
    AASR_HIP(hipMemcpy(a.data(), d_a.p, a.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> ph((size_t)D);
    for (int i = 0; i < P; i++) {
      const float *x = &fr[(size_t)i * D];
      for (int64_t s2 = 0; s2 < S; s2++) {
        if (!g->f16_state_ok[(size_t)s2] || bad[(size_t)s2]) continue;
        double sum = 0;
        for (int32_t k = m.mix_off[s2]; k < m.mix_off[s2 + 1]; k++) {
          if (!live[(size_t)k]) continue;
          const int64_t gi = m.mix_idx[(size_t)k];
          const double *mu = &m.mean[(size_t)gi * D], *var = &m.var[(size_t)gi * D];
          double q = 0;
          for (int d = 0; d < D; d++) {
            const double t = (double)x[d] - mu[d];
            q += var[d] > 0 ? t * t / var[d] : 0.0;
          }
          sum += std::exp(cst[(size_t)k] - 0.5 * q);
        }
        const double ref = std::log(std::max(sum, 1e-50)) + g->out_bias_ln;
        // (three-term part: 3/4 of the contract -- its operands carry 24 bits, what the probe sees there is the f32
        // accumulators' granularity at |log2 value| ~ 100, 5-7e-5 on the probe's 6-sigma frames, not conditioning)
        // (slab-constant part: admitted by limits calibrated off the data up to 7.5e-5, tools/exp_calib.py)
        if (ref > -103.0 && !(std::fabs((double)a[(size_t)i * S + s2] - ref) <= (double)probe_tol * ((pg3 || m.pg_sc()) ? 1.5 : 1.0))) {
          bad[(size_t)s2] = 1;
          n_bad++;
        }
      }
    }
    {
      ProbeCache &pc = probe_cache();
      std::lock_guard<std::mutex> lk(pc.mu);
      pc.runs++;
      if (pc.verdicts.size() >= 256) pc.verdicts.clear();
      std::vector<int32_t> &v = pc.verdicts[key];
      v.clear();
      for (int64_t s2 = 0; s2 < S; s2++)
        if (bad[(size_t)s2]) v.push_back((int32_t)s2);
    }
    }   // !cached
    if (n_bad == 0) return;
    g->f16_probe_moved += n_bad;
    for (int64_t s2 = 0; s2 < S; s2++)
      if (bad[(size_t)s2]) g->f16_state_ok[(size_t)s2] = 0;
    if (m.n_pg() > 0) return;   // a multi-pivot engine part: the planner takes the marked states out and builds it again
    // the whole-model fp16 rows are gone: the model keeps its three-term rows, and the planner of the engine parts (which
    // runs next and starts from f16_state_ok) gives the states that still qualify their two-term rows back
    g->f16_whole_rejected = true;   // (a layout built later -- aasr_debug_set_layouts -- must not pack them again)
    g->paired.a16h = DevBuf<uint16_t>();
    g->paired.states_f16 = 0;
    g->tracks.a16h = DevBuf<uint16_t>();
    g->tracks.states_f16 = 0;
    return;
  }
}

// ---------------------------------------------------------------------------
// Full-covariance kernel (see gmm_build_fullcov()).
//
// Same frame-stationary skeleton; the streamed rows are the rows of
// sqrt(log2e/2) * R^-1 of every mixture component (Sigma = R R^T), K = dim + 1.
// The accumulators hold y = R^-1 (x - mu); the epilogue squares and sums them
// per component (one FMA per value), turns each finished component into
// 2^(C_g - |y|^2 + ref) and adds it to its state's running sum.  Two
// independent row tracks, results stored per state.
// ---------------------------------------------------------------------------
// CL (Gaussian clustering over a full-covariance pool, gmm_cluster.hip): a component counts for a frame only where
// the selection bit of its rows is set (all rows of a component belong to one cluster; the bit of its last row is
// tested where the component closes), and the result carries no 1e-50 floor (k_cluster_merge applies it).
template <int NKK, bool CL>
__global__ __launch_bounds__(256, 2) void k_gmm_full_score(
    const float *__restrict__ frames, int64_t F, int dim, const float *__restrict__ pivot,
    const float *__restrict__ apack, const int32_t *__restrict__ split_row,
    const uint32_t *__restrict__ close_mask, const float *__restrict__ gconst, int g_stride,
    const int32_t *__restrict__ sid, int s_stride, float *__restrict__ out, int64_t S, float ref_ln,
    ClusterArgs cl) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *smem = (float *)smem_raw;
  constexpr int kTileFloats = (NKK / 2) * 64 * 4;
  float *abuf0 = smem;
  float *abuf1 = smem + kTileFloats;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int n = lane & 31;
  const int h = lane >> 5;
  const int64_t f0 = (int64_t)blockIdx.x * FRAMES_PER_BLOCK + wave * FRAMES_PER_WAVE;

  // B[kk][nb]: K index k = 2*kk + h -> x'_k (k < dim), 1 (k == dim), 0 beyond
  float bf[NKK][2];
#pragma unroll
  for (int nb = 0; nb < 2; nb++) {
    int64_t f = f0 + nb * 32 + n;
    if (f > F - 1) f = F - 1;
    const float *xr = frames + f * dim;
#pragma unroll
    for (int kk = 0; kk < NKK; kk++) {
      const int k = 2 * kk + h;
      const int kc = k < dim ? k : 0;
      float v = xr[kc] - pivot[kc];
      if (k == dim) v = 1.0f;
      if (k > dim) v = 0.0f;
      bf[kk][nb] = v;
    }
  }

  const int64_t t_begin = split_row[8 * blockIdx.y];
  const int64_t t_end = split_row[8 * blockIdx.y + 8];
  issue_tile_copy(apack + (size_t)t_begin * kTileFloats, abuf0, kTileFloats, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float q0 = 0.0f, q1 = 0.0f;  // |y|^2 of the open component, frames n / 32+n
  float s0 = 0.0f, s1 = 0.0f;  // sum over finished components of the open state
  int ks = split_row[8 * blockIdx.y + 1 + h];
  int kg = split_row[8 * blockIdx.y + 3 + h];
  const int32_t *my_sid = sid + h * s_stride;
  const float *my_gc = gconst + h * g_stride;
  int next_sid = my_sid[ks];
  float next_gc = my_gc[kg];
  float *orow0 = out + (f0 + n) * S;
  float *orow1 = out + (f0 + 32 + n) * S;
  const bool ok0 = f0 + n < F, ok1 = f0 + 32 + n < F;
  const float floor_val = CL ? cl.floor_val : LOG_TINY_F;
  const unsigned long long *mrow = CL ? cl.maskrow + (size_t)(f0 >> 6) * cl.rows_padded + lane : nullptr;
  const int etest = (dim - 1) & 3;   // element of a component's last quad that holds its last row

  for (int64_t t = t_begin; t < t_end; t++) {
    const int par = (int)((t - t_begin) & 1);
    float *acur = par ? abuf1 : abuf0;
    float *anext = par ? abuf0 : abuf1;
    if (t + 1 < t_end)
      issue_tile_copy(apack + (size_t)(t + 1) * kTileFloats, anext, kTileFloats, wave, lane);
    unsigned long long bits = 0;
    if (CL) bits = mrow[(size_t)t * TILE_ROWS];   // k_cluster_expand's per-lane word of this tile
    const unsigned m32 = sload_close32(close_mask, t);
    const unsigned gmask = h ? ((m32 >> 8) & 0xffu) : (m32 & 0xffu);
    const unsigned smask = h ? ((m32 >> 24) & 0xffu) : ((m32 >> 16) & 0xffu);

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    const f32x4 *afrag = (const f32x4 *)acur + lane;
    f32x4 a0 = afrag[0];
    f32x4 a1 = afrag[(NKK / 2 > 1 ? 1 : 0) * 64];
#pragma unroll
    for (int q = 0; q < NKK / 2; q++) {
      const int qn = (q + 2 < NKK / 2) ? q + 2 : NKK / 2 - 1;
      f32x4 a2 = afrag[qn * 64];
      const f32x4 av = a0;
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[2 * q][0], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bf[2 * q][1], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[2 * q][0], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bf[2 * q][1], c11, 0, 0, 0);
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[2 * q + 1][0], c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bf[2 * q + 1][1], c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[2 * q + 1][0], c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bf[2 * q + 1][1], c11, 0, 0, 0);
      a0 = a1;
      a1 = a2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
      const f32x16 &ca = mb ? c10 : c00;
      const f32x16 &cb = mb ? c11 : c01;
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          q0 = fmaf(ca[4 * q + e], ca[4 * q + e], q0);
          q1 = fmaf(cb[4 * q + e], cb[4 * q + e], q1);
        }
        if ((gmask >> (mb * 4 + q)) & 1) {
          float e0 = __builtin_amdgcn_exp2f(next_gc - q0);
          float e1 = __builtin_amdgcn_exp2f(next_gc - q1);
          if (CL) {
            e0 = ((bits >> (8 * q + 4 * mb + etest)) & 1ull) ? e0 : 0.0f;
            e1 = ((bits >> (32 + 8 * q + 4 * mb + etest)) & 1ull) ? e1 : 0.0f;
          }
          s0 += e0;
          s1 += e1;
          q0 = 0.0f;
          q1 = 0.0f;
          kg++;
          next_gc = my_gc[kg];
          if ((smask >> (mb * 4 + q)) & 1) {
            float l0 = fmaf(__builtin_amdgcn_logf(s0), LN2_F, -ref_ln);
            float l1 = fmaf(__builtin_amdgcn_logf(s1), LN2_F, -ref_ln);
            l0 = fmaxf(l0, floor_val);
            l1 = fmaxf(l1, floor_val);
            if (ok0) orow0[next_sid] = l0;
            if (ok1) orow1[next_sid] = l1;
            s0 = 0.0f;
            s1 = 0.0f;
            ks++;
            next_sid = my_sid[ks];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// The same kernel on the bf16 matrix pipe (AASR_PREC_BF16X3): rows and frames as
// three bf16 terms, six products per slab accumulated in f32 -- the scheme of
// k_gmm_diag_score_bf16x3 (rolling A-fragment prefetch ordered by split, tile
// copy through inline assembly, close bits requested mid-stream one tile ahead).
// K = dim + 1 padded to a multiple of 16, K index = column of R^-1 | bias.
// ---------------------------------------------------------------------------
// NS = 2 (AASR_PREC_F16X2): rows and frames as two fp16 terms, three products per slab -- half the matrix
// instructions.  State-level error ~2x the three-term form's at the same conditioning (tools/exp_fullcov_f16.py), so
// a pool takes it only below FULL_KAPPA_LIMIT_F16 (gmm.h); the frame operand is clamped to +-kFullF16Clamp.
// CL (Gaussian clustering over a full-covariance pool): a component counts for a frame only where its cluster's bit of
// the tile's per-lane word (k_cluster_expand) is set; no floor on the states -- the merge adds the centres.
template <int NK16, int NS, bool CL = false>
__global__ __launch_bounds__(256, 2) void k_gmm_full_score_bf16x3(
    const float *__restrict__ frames, int64_t F, int dim, const float *__restrict__ pivot,
    const uint16_t *__restrict__ apack, const int32_t *__restrict__ split_row,
    const uint32_t *__restrict__ close_mask, const float *__restrict__ gc_tile,
    const int32_t *__restrict__ sid_tile, float *__restrict__ out, int64_t S, float ref_ln, ClusterArgs cl,
    const float *__restrict__ f16scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int kTileFloats = NK16 * NS * 2 * 64 * 16 / 4;
  float *abuf0 = (float *)smem_raw;
  float *abuf1 = abuf0 + kTileFloats;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int n = lane & 31;
  const int h = lane >> 5;
  const int64_t f0 = (int64_t)blockIdx.x * FRAMES_PER_BLOCK + wave * FRAMES_PER_WAVE;

  // frame operand: lane (n, h) holds k = 16*j + 8*h + i, i < 8: x'_k (k < dim), 1 (k == dim), 0 beyond
  u32x4 bq[NK16][NS][2];
#pragma unroll
  for (int nb = 0; nb < 2; nb++) {
    int64_t f = f0 + nb * 32 + n;
    if (f > F - 1) f = F - 1;
    const float *xr = frames + f * dim;
#pragma unroll
    for (int j = 0; j < NK16; j++) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int k = 16 * j + 8 * h + i;
        const int kc = k < dim ? k : 0;
        float val = xr[kc] - pivot[kc];
        // two fp16 terms: the column's power-of-two scale (the factor rows carry its inverse: exact), then the fp16 range
        if (NS == 2) val = fminf(fmaxf(val * f16scale[kc], -kFullF16Clamp), kFullF16Clamp);
        if (k == dim) val = 1.0f;
        if (k > dim) val = 0.0f;
        v[i] = val;
      }
      unsigned w1[4], w2[4], w3[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if constexpr (NS == 3) split3_pair(v[2 * i], v[2 * i + 1], w1[i], w2[i], w3[i]);
        else split2_pair(v[2 * i], v[2 * i + 1], w1[i], w2[i]);
      }
      bq[j][0][nb] = u32x4{w1[0], w1[1], w1[2], w1[3]};
      bq[j][1][nb] = u32x4{w2[0], w2[1], w2[2], w2[3]};
      if constexpr (NS == 3) bq[j][2][nb] = u32x4{w3[0], w3[1], w3[2], w3[3]};
    }
  }

  const int64_t t_begin = split_row[8 * blockIdx.y];
  const int64_t t_end = split_row[8 * blockIdx.y + 8];
  const float *apf = (const float *)apack;
  issue_tile_copy_raw(apf + (size_t)t_begin * kTileFloats, abuf0, kTileFloats, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float q0 = 0.0f, q1 = 0.0f;  // |y|^2 of the open component, frames n / 32+n
  float s0 = 0.0f, s1 = 0.0f;  // sum over finished components of the open state
  // this track's closing constants / state indices of a tile, by quad position
  const f32x4 *gct = (const f32x4 *)(gc_tile + h * 8);
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 *sdt = (const i32x4 *)(sid_tile + h * 8);
  float *orow0 = out + (f0 + n) * S;
  float *orow1 = out + (f0 + 32 + n) * S;
  const bool ok0 = f0 + n < F, ok1 = f0 + 32 + n < F;
  const float floor_val = CL ? cl.floor_val : LOG_TINY_F;
  const unsigned long long *mrow = CL ? cl.maskrow + (size_t)(f0 >> 6) * cl.rows_padded + lane : nullptr;
  const int etest = (dim - 1) & 3;   // element of a component's last quad that holds its last row

  unsigned m32_next = t_begin < t_end ? (unsigned)__builtin_amdgcn_readfirstlane((int)close_mask[t_begin]) : 0u;
  unsigned mask_v = 0;
  u32x4 afr[NS][2];
  if (t_begin < t_end) {
#pragma unroll
    for (int sp = NS - 1; sp >= 0; sp--) {
      afr[sp][0] = ((const u32x4 *)abuf0 + lane)[(sp * 2 + 0) * 64];
      afr[sp][1] = ((const u32x4 *)abuf0 + lane)[(sp * 2 + 1) * 64];
    }
  }
  for (int64_t t = t_begin; t < t_end; t++) {
    const int par = (int)((t - t_begin) & 1);
    float *acur = par ? abuf1 : abuf0;
    float *anext = par ? abuf0 : abuf1;
    if (t + 1 < t_end)
      issue_tile_copy_raw(apf + (size_t)(t + 1) * kTileFloats, anext, kTileFloats, wave, lane);
    // fetched with the tile: they land under the matrix stream, the epilogue never waits on memory
    const f32x4 gca = gct[4 * t], gcb = gct[4 * t + 1];
    const i32x4 sda = sdt[4 * t], sdb = sdt[4 * t + 1];
    const float gcv[8] = {gca.x, gca.y, gca.z, gca.w, gcb.x, gcb.y, gcb.z, gcb.w};
    const int sdv[8] = {sda.x, sda.y, sda.z, sda.w, sdb.x, sdb.y, sdb.z, sdb.w};
    unsigned long long bits = 0;
    if (CL) bits = mrow[(size_t)t * TILE_ROWS];   // k_cluster_expand's per-lane word of this tile
    const unsigned m32 = m32_next;
    const unsigned gmask = h ? ((m32 >> 8) & 0xffu) : (m32 & 0xffu);
    const unsigned smask = h ? ((m32 >> 24) & 0xffu) : ((m32 >> 16) & 0xffu);

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    const u32x4 *afrag = (const u32x4 *)acur + lane;  // [slab][split][mb][64 lanes]
#pragma unroll
    for (int j = 0; j < NK16; j++) {
#pragma unroll
      for (int grp = 0; grp < NS; grp++) {
        const int sp = NS - 1 - grp;  // a3 | a2 | a1
        const int nprod = grp + 1;    // b1 | b2 b1 | b3 b2 b1
#pragma unroll
        for (int c = 0; c < nprod; c++) {
          const int sb = nprod - 1 - c;
          c00 = mfma_split<NS>(afr[sp][0], bq[j][sb][0], c00);
          c01 = mfma_split<NS>(afr[sp][0], bq[j][sb][1], c01);
          c10 = mfma_split<NS>(afr[sp][1], bq[j][sb][0], c10);
          c11 = mfma_split<NS>(afr[sp][1], bq[j][sb][1], c11);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j == 0 && grp == 0) mask_v = close_mask[t + 1];  // the array has one spare element
        if (j + 1 < NK16) {
          afr[sp][0] = afrag[(((j + 1) * NS + sp) * 2 + 0) * 64];
          afr[sp][1] = afrag[(((j + 1) * NS + sp) * 2 + 1) * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(mask_v) : : "memory");
    __builtin_amdgcn_s_barrier();
    m32_next = (unsigned)__builtin_amdgcn_readfirstlane((int)mask_v);
    if (t + 1 < t_end) {
      const u32x4 *nfrag = (const u32x4 *)anext + lane;
#pragma unroll
      for (int sp = NS - 1; sp >= 0; sp--) {
        afr[sp][0] = nfrag[(sp * 2 + 0) * 64];
        afr[sp][1] = nfrag[(sp * 2 + 1) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }

#pragma unroll
    for (int mb = 0; mb < 2; mb++) {
      const f32x16 &ca = mb ? c10 : c00;
      const f32x16 &cb = mb ? c11 : c01;
#pragma unroll
      for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          q0 = fmaf(ca[4 * q + e], ca[4 * q + e], q0);
          q1 = fmaf(cb[4 * q + e], cb[4 * q + e], q1);
        }
        if ((gmask >> (mb * 4 + q)) & 1) {
          float e0 = __builtin_amdgcn_exp2f(gcv[mb * 4 + q] - q0);
          float e1 = __builtin_amdgcn_exp2f(gcv[mb * 4 + q] - q1);
          if (CL) {
            e0 = ((bits >> (8 * q + 4 * mb + etest)) & 1ull) ? e0 : 0.0f;
            e1 = ((bits >> (32 + 8 * q + 4 * mb + etest)) & 1ull) ? e1 : 0.0f;
          }
          s0 += e0;
          s1 += e1;
          q0 = 0.0f;
          q1 = 0.0f;
          if ((smask >> (mb * 4 + q)) & 1) {
            float l0 = fmaf(__builtin_amdgcn_logf(s0), LN2_F, -ref_ln);
            float l1 = fmaf(__builtin_amdgcn_logf(s1), LN2_F, -ref_ln);
            l0 = fmaxf(l0, floor_val);
            l1 = fmaxf(l1, floor_val);
            if (ok0) orow0[sdv[mb * 4 + q]] = l0;
            if (ok1) orow1[sdv[mb * 4 + q]] = l1;
            s0 = 0.0f;
            s1 = 0.0f;
          }
        }
      }
    }
  }
}

template <int NK16, int NS = 3, bool CL = false>
static void launch_full_bf16_t(const aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                               hipStream_t stream, const ClusterArgs &cl = ClusterArgs()) {
  const FullLayout &L = g->full;
  const int64_t blocks = (F + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
  const int smem = 2 * NK16 * NS * 2 * 64 * 16;
  const double slots = 2.0 * (g->num_cus > 0 ? g->num_cus : 256);
  int R = 1;
  double best_eff = 0;
  for (int r = 1; r <= L.max_splits; r++) {
    double x = (double)blocks * r / slots;
    double eff = x / std::ceil(x);
    if (x < 1.0) eff = x;
    if (eff > best_eff + 0.005) {
      best_eff = eff;
      R = r;
    }
  }
  const int32_t *split_row = L.splits.p + (size_t)(R - 1) * (TRACK_MAX_SPLITS + 1) * 8;
  hipLaunchKernelGGL((k_gmm_full_score_bf16x3<NK16, NS, CL>), dim3((unsigned)blocks, (unsigned)R), dim3(256), smem,
                     stream, d_frames, F, g->dim, g->d_pivot.p, NS == 2 ? L.a16h.p : L.a16.p, split_row, L.close.p,
                     L.gc_tile.p, L.sid_tile.p, d_out, g->S, L.ref_ln, cl, NS == 2 ? L.f16scale.p : nullptr);
  AASR_HIP(hipGetLastError());
}

template <int NKK, bool CL = false>
static void launch_full_t(const aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                          hipStream_t stream, const ClusterArgs &cl = ClusterArgs()) {
  const FullLayout &L = g->full;
  const int64_t blocks = (F + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
  const int smem = 2 * (NKK / 2) * 64 * 4 * 4;
  const double slots = 2.0 * (g->num_cus > 0 ? g->num_cus : 256);
  int R = 1;
  double best_eff = 0;
  for (int r = 1; r <= L.max_splits; r++) {
    double x = (double)blocks * r / slots;
    double eff = x / std::ceil(x);
    if (x < 1.0) eff = x;
    if (eff > best_eff + 0.005) {
      best_eff = eff;
      R = r;
    }
  }
  const int32_t *split_row = L.splits.p + (size_t)(R - 1) * (TRACK_MAX_SPLITS + 1) * 8;
  hipLaunchKernelGGL((k_gmm_full_score<NKK, CL>), dim3((unsigned)blocks, (unsigned)R), dim3(256), smem, stream,
                     d_frames, F, g->dim, g->d_pivot.p, L.rows.a.p, split_row, L.close.p, L.gconst.p,
                     L.g_stride, L.sid.p, L.s_stride, d_out, g->S, L.ref_ln, cl);
  AASR_HIP(hipGetLastError());
}

void gmm_full_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                     hipStream_t stream) {
  if (!g->full.ok) raise(AASR_ERR_UNSUPPORTED, "full-covariance layout was not built for this model");
  if (g->use_bf16x3 && g->precision == AASR_PREC_F16X2 && g->full.a16h.p) {
    switch (g->full.nk16) {
      case 1: launch_full_bf16_t<1, 2>(g, d_frames, F, d_out, stream); return;
      case 2: launch_full_bf16_t<2, 2>(g, d_frames, F, d_out, stream); return;
      case 3: launch_full_bf16_t<3, 2>(g, d_frames, F, d_out, stream); return;
      case 4: launch_full_bf16_t<4, 2>(g, d_frames, F, d_out, stream); return;
      default: break;
    }
  }
  if (g->use_bf16x3 && g->full.a16.p) {
    switch (g->full.nk16) {
      case 1: launch_full_bf16_t<1>(g, d_frames, F, d_out, stream); return;
      case 2: launch_full_bf16_t<2>(g, d_frames, F, d_out, stream); return;
      case 3: launch_full_bf16_t<3>(g, d_frames, F, d_out, stream); return;
      case 4: launch_full_bf16_t<4>(g, d_frames, F, d_out, stream); return;
      default: break;
    }
  }
  switch (g->full.rows.nkk) {
#define AASR_CASE(N)                                   \
  case N:                                              \
    launch_full_t<N>(g, d_frames, F, d_out, stream);   \
    return;
    AASR_CASE(8) AASR_CASE(14) AASR_CASE(20) AASR_CASE(26) AASR_CASE(32)
#undef AASR_CASE
    default:
      raise(AASR_ERR_UNSUPPORTED, "no full-covariance kernel instance for K/2 = %d", g->full.rows.nkk);
  }
}

// Gaussian clustering over a full-covariance pool: the exact part of every state on the f32 factor-row kernel with
// the selection masks (the fp16 / bf16 matrix forms where the rows are packed for them, else the f32 kernel), no floor -- the merge adds the centres.
void gmm_full_masked_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                            const unsigned long long *maskrow, hipStream_t stream) {
  if (!g->full.ok) raise(AASR_ERR_UNSUPPORTED, "full-covariance layout was not built for this model");
  ClusterArgs cl;
  cl.maskrow = maskrow;
  cl.rows_padded = g->full.rows_padded;
  cl.floor_val = NEG_BIG_F;
  // the matrix-pipe forms of the rows (two fp16 / three bf16 terms) with the masks, where they are packed
  if (g->use_bf16x3 && g->precision == AASR_PREC_F16X2 && g->full.a16h.p) {
    switch (g->full.nk16) {
      case 1: launch_full_bf16_t<1, 2, true>(g, d_frames, F, d_out, stream, cl); return;
      case 2: launch_full_bf16_t<2, 2, true>(g, d_frames, F, d_out, stream, cl); return;
      case 3: launch_full_bf16_t<3, 2, true>(g, d_frames, F, d_out, stream, cl); return;
      case 4: launch_full_bf16_t<4, 2, true>(g, d_frames, F, d_out, stream, cl); return;
      default: break;
    }
  }
  if (g->use_bf16x3 && g->full.a16.p) {
    switch (g->full.nk16) {
      case 1: launch_full_bf16_t<1, 3, true>(g, d_frames, F, d_out, stream, cl); return;
      case 2: launch_full_bf16_t<2, 3, true>(g, d_frames, F, d_out, stream, cl); return;
      case 3: launch_full_bf16_t<3, 3, true>(g, d_frames, F, d_out, stream, cl); return;
      case 4: launch_full_bf16_t<4, 3, true>(g, d_frames, F, d_out, stream, cl); return;
      default: break;
    }
  }
  switch (g->full.rows.nkk) {
#define AASR_CASE(N)                                                  \
  case N:                                                             \
    launch_full_t<N, true>(g, d_frames, F, d_out, stream, cl);        \
    return;
    AASR_CASE(8) AASR_CASE(14) AASR_CASE(20) AASR_CASE(26) AASR_CASE(32)
#undef AASR_CASE
    default:
      raise(AASR_ERR_UNSUPPORTED, "no full-covariance kernel instance for K/2 = %d", g->full.rows.nkk);
  }
}

// ---------------------------------------------------------------------------
// Centred-form kernel: the numerically safe path.
//
// The expanded (GEMM) form cancels when |mu - pivot| / sigma is large; models
// whose conditioning estimate kappa = max_g sum_d p_gd (mu_gd - v_d)^2 would push
// the f32 error past the 1e-4 budget are scored with the reference's own
// arithmetic shape instead: t = x - mu, acc += (p') t^2 per dimension (all terms
// of one sign, no cancellation), online (max, sum) over a state's components.
// One lane owns one frame (its x vector lives in VGPRs), the Gaussian
// parameters are wave-uniform and arrive through the scalar cache (s_load), so
// the inner loop is 3 VALU instructions per dimension.  The f32 MFMA runs on
// the same lanes as the VALU anyway, so this costs ~1.5x the matrix path, not
// 16x.  Also the fallback for any model the track layouts cannot hold.
// ---------------------------------------------------------------------------
// CL (Gaussian clustering, gmm_cluster.hip): record r belongs to cluster crow[r]; its value counts
// for a frame only where the frame's bit of maskw[word][cluster] is set (the cluster is evaluated
// exactly there), and the result carries no 1e-50 floor (k_cluster_merge applies it).
template <int DIMP, bool CL>
__global__ __launch_bounds__(256) void k_gmm_diag_score_centred(
    const float *__restrict__ frames, int64_t F, int dim, const float *__restrict__ recs,
    const int32_t *__restrict__ state_off, const int32_t *__restrict__ split_state,
    float *__restrict__ out, int64_t frame_stride, int64_t state_stride,
    const int32_t *__restrict__ crow, const unsigned long long *__restrict__ maskw, int c1, int64_t n_words,
    float floor_val, int tile_out) {
  // A record = DIMP / 4 groups of 16 floats, group q = [mu x 4][mu_lo x 4][p' x 4][C, pad x 3] of dimensions 4 q .. 4 q + 3
  // (the constant in group 0): the mean as a float pair, mu = mu_hi + mu_lo to 2^-48 -- a mean rounded to one float
  // costs p t ulp(mu)/2, 1e-4 at 14 sigma from a sigma = 0.01 Gaussian.  A group is ONE scalar load of 64 bytes and the
  // groups of consecutive records follow each other in memory, so the kernel walks one stream and fetches a group ahead
  // of the one it computes on (round 6: with [mu][p'][C][mu_lo] a dimension needed three loads from three places, none
  // could be issued early within the scalar registers, and the waves stood at s_waitcnt: 36 % of the vector rate).
  constexpr int NG = DIMP / 4;
  constexpr int REC = 16 * NG;
  typedef float f32x16u __attribute__((ext_vector_type(16), aligned(64)));
  // LDS: first the staging area of the prologue (128 frames x (dim | 1) floats), then -- tile_out -- the results of 16
  // consecutive states for the workgroup's 512 frames ([512][17]), written out as runs of 16 floats per frame row
  extern __shared__ float cen_smem[];
  // each lane owns TWO frames (f, f + 256): one scalar fetch of a Gaussian's
  // parameters feeds 128 frame x Gaussian pairs per wave
  const int tid = threadIdx.x;
  const int64_t f_base = (int64_t)blockIdx.x * 512;
  const int64_t fa = f_base + tid;
  const int64_t fb = fa + 256;
  // The two frames of a lane travel as one <2 x float>: t = x - mu, t*t, fma with p' are
  // v_pk_add / v_pk_mul / v_pk_fma_f32 (two frames per instruction, the scalar operand
  // broadcast) -- 1.5 VALU instructions per frame and dimension instead of 3, same roundings.
  f32x2 x2[DIMP];
  // Prologue: the workgroup's frames are one contiguous run of the frame matrix; it is copied through LDS in four quarters
  // (coalesced loads; a lane then reads its own row, rows an odd number of floats apart: no bank conflicts).  Lanes
  // beyond F take zeros (never stored).
  {
    const int dimo = dim | 1;
#pragma unroll
    for (int qt = 0; qt < 4; qt++) {
      const int64_t f0 = f_base + qt * 128;
      const int nfr = (int)max((int64_t)0, min((int64_t)128, F - f0));
      const int n = nfr * dim;
      const float *src = frames + f0 * dim;
      __syncthreads();
      for (int i = tid; i < n; i += 256) {
        const int fr = i / dim;
        cen_smem[fr * dimo + (i - fr * dim)] = src[i];
      }
      __syncthreads();
      const int row = (tid & 127) < nfr ? (tid & 127) : 0;
      const bool mine = (tid >> 7) == (qt & 1);   // frames f_base + tid (quarters 0, 1) and f_base + 256 + tid (2, 3)
      if (mine) {
        if (qt < 2) {
#pragma unroll
          for (int d = 0; d < DIMP; d++) x2[d].x = (d < dim && nfr > 0) ? cen_smem[row * dimo + d] : 0.0f;
        } else {
#pragma unroll
          for (int d = 0; d < DIMP; d++) x2[d].y = (d < dim && nfr > 0) ? cen_smem[row * dimo + d] : 0.0f;
        }
      }
    }
    __syncthreads();
  }
  const int s_begin = split_state[blockIdx.y], s_end = split_state[blockIdx.y + 1];
  // the 64-frame words of this wave's two frame groups (wave-uniform)
  const int64_t word_a = min((int64_t)blockIdx.x * 8 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_words - 1);
  const int64_t word_b = min(word_a + 4, n_words - 1);
  const int lane_bit = threadIdx.x & 63;
  // the stream of groups: from the first record of the state range on, RING - 1 groups ahead (RING divides the groups of a
  // record, so a group's ring slot is a compile-time constant: no copies between the scalar registers; 12 of them per
  // slot).  One group ahead left the waves waiting on records that come from L2 (whole-model runs: 32 MB of records,
  // 349 -> 296 ms per 10^6 frames x 50 k rows); four ahead covers it.
  constexpr int RING = NG % 5 == 0 ? 5 : NG % 4 == 0 ? 4 : NG % 3 == 0 ? 3 : 2;
  const f32x16u *gp = (const f32x16u *)(recs + (size_t)state_off[s_begin] * REC);
  f32x16u ring[RING];
#pragma unroll
  for (int i = 0; i < RING - 1; i++) ring[i] = gp[i];   // (short ranges: the spare records behind the last one)
  gp += RING - 1;
  for (int s = s_begin; s < s_end; s++) {
    const int r0 = state_off[s], r1 = state_off[s + 1];
    float ma = NEG_BIG_F, sa = 0.0f, mb = NEG_BIG_F, sb = 0.0f;
    for (int r = r0; r < r1; r++) {
      bool on_a = true, on_b = true;
      if (CL) {
        const int c = crow[r];
        on_a = (maskw[word_a * c1 + c] >> lane_bit) & 1ull;
        on_b = (maskw[word_b * c1 + c] >> lane_bit) & 1ull;
      }
      f32x2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};  // two chains per frame
      float c = 0.0f;
#pragma unroll
      for (int q = 0; q < NG; q++) {
        ring[(q + RING - 1) % RING] = *gp;   // (the spare records behind the last one keep this inside the buffer)
        gp++;
        const f32x16u cur = ring[q % RING];
        if (q == 0) c = cur[12];
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const float mu0 = cur[j], mu1 = cur[j + 1];
          const float ml0 = cur[4 + j], ml1 = cur[4 + j + 1];
          const float p0 = cur[8 + j], p1 = cur[8 + j + 1];
          const f32x2 t0 = (x2[4 * q + j] - (f32x2){mu0, mu0}) - (f32x2){ml0, ml0};
          const f32x2 t1 = (x2[4 * q + j + 1] - (f32x2){mu1, mu1}) - (f32x2){ml1, ml1};
          acc0 = __builtin_elementwise_fma(t0 * t0, (f32x2){p0, p0}, acc0);
          acc1 = __builtin_elementwise_fma(t1 * t1, (f32x2){p1, p1}, acc1);
        }
      }
      const float a0 = acc0.x, b0 = acc0.y, a1 = acc1.x, b1 = acc1.y;
      float la = c + (a0 + a1), lb = c + (b0 + b1);  // log2 units
      if (CL) {
        la = on_a ? la : NEG_BIG_F;
        lb = on_b ? lb : NEG_BIG_F;
      }
      const float na = fmaxf(ma, la), nb = fmaxf(mb, lb);
      sa = sa * __builtin_amdgcn_exp2f(ma - na) + __builtin_amdgcn_exp2f(la - na);
      sb = sb * __builtin_amdgcn_exp2f(mb - nb) + __builtin_amdgcn_exp2f(lb - nb);
      ma = na;
      mb = nb;
    }
    float lla = fmaf(ma, LN2_F, __builtin_amdgcn_logf(sa) * LN2_F);
    float llb = fmaf(mb, LN2_F, __builtin_amdgcn_logf(sb) * LN2_F);
    lla = fmaxf(lla, floor_val);
    llb = fmaxf(llb, floor_val);
    if (r1 <= r0) lla = llb = floor_val;
    if (!tile_out) {
      if (fa < F) out[fa * frame_stride + s * state_stride] = lla;
      if (fb < F) out[fb * frame_stride + s * state_stride] = llb;
      continue;
    }
    // frame-major output (state_stride == 1): a lane's values of 16 consecutive states are collected in LDS and leave as
    // runs of 16 floats per frame row -- whole 64-byte half lines where the caller's pitch is a multiple of 16 floats
    const int col = (s - s_begin) & 15;
    cen_smem[tid * 17 + col] = lla;
    cen_smem[(tid + 256) * 17 + col] = llb;
    if (col == 15 || s == s_end - 1) {
      __syncthreads();
      const int c = tid & 15;
      const int64_t s0 = s - col;
      if (c <= col) {
#pragma unroll 4
        for (int r = tid >> 4; r < 512; r += 16) {
          const int64_t f = f_base + r;
          if (f < F) out[f * frame_stride + s0 + c] = cen_smem[r * 17 + c];
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------
// AASR_PREC_F64: the reference's own arithmetic, operation by operation, in double
// (DiagonalGaussian::compute_log_likelihood, aku/Distributions.cc:1040-1062: ll += d*d*p over the
// dimensions, ll *= -0.5, ll += constant; compute_likelihood :1033-1037 = exp; Mixture::
// compute_likelihood :2078-2086: l += w * lik in component order; HmmSet's 1e-50 clamp
// :497-498).  The build has -ffp-contract=off, so every product and sum is rounded separately as
// in the reference's x86-64 build; what is left against the oracle is the device's exp() and
// log() (<= 1 ulp).  One lane per frame (its vector in VGPRs as doubles), the Gaussian records are
// wave-uniform and arrive through the scalar cache.  A verification / training-side mode:
// ~6 f64 operations per frame, Gaussian and dimension on the vector ALU.
// ---------------------------------------------------------------------------
// CL: Gaussian clustering -- component r belongs to cluster crow[r]; where the frame's bit of
// maskw[word][cluster] is clear the component takes its centre's likelihood, recovered from the
// ranking key the centre kernel stored (key == ll where exp(ll) is a normal double, else the
// denormal's integer multiple of 2^-1074, gmm_cluster.hip lin_key).
template <int DIMP, bool CL>
__global__ __launch_bounds__(256) void k_gmm_diag_score_f64(const double *__restrict__ frames, int64_t F, int dim,
                                                            const double *__restrict__ recs,
                                                            const int32_t *__restrict__ state_off, int64_t S,
                                                            double *__restrict__ out, int linear, double det,
                                                            const int32_t *__restrict__ crow,
                                                            const unsigned long long *__restrict__ maskw, int c1,
                                                            const double *__restrict__ ll64, int64_t Cs, int C) {
  constexpr int REC = 2 * DIMP + 2;  // [mean x DIMP][precision x DIMP][constant, weight]
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t fc = f < F ? f : F - 1;
  double x[DIMP];
#pragma unroll
  for (int d = 0; d < DIMP; d++) x[d] = d < dim ? frames[fc * dim + d] : 0.0;
  const int64_t s_per = (S + gridDim.y - 1) / gridDim.y;
  const int64_t s_begin = (int64_t)blockIdx.y * s_per, s_end = min(S, s_begin + s_per);
  for (int64_t s = s_begin; s < s_end; s++) {
    const int r0 = state_off[s], r1 = state_off[s + 1];
    double l = 0;
    for (int r = r0; r < r1; r++) {
      const double *rec = recs + (size_t)r * REC;
      double ll = 0;
#pragma unroll
      for (int d = 0; d < DIMP; d++) {
        const double t = x[d] - rec[d];
        ll += t * t * rec[DIMP + d];
      }
      ll *= -0.5;
      ll += rec[2 * DIMP];
      // AdaptedGaussian::compute_likelihood = g(A f + b) * |det| (aku/ModelModules.hh:172-173); det = 1 unadapted
      double lik = exp(ll) * det;
      if (CL) {
        const int c = crow[r];
        const bool on = (maskw[(fc >> 6) * c1 + c] >> (fc & 63)) & 1ull;
        if (!on) {  // c < C here: the "no cluster" column C is all ones
          const double key = ll64[fc * Cs + (c < C ? c : 0)];
          lik = key > -1000.0 ? exp(key) : ldexp((key + 2000.0) * 4398046511104.0, -1074);
        }
      }
      l += rec[2 * DIMP + 1] * lik;
    }
    if (l < 1e-50) l = 1e-50;  // also NaN-free: comparisons with NaN are false, as in the reference
    if (f < F) out[f * S + s] = linear ? l : log(l);
  }
}

__global__ void k_f32_to_f64(const float *__restrict__ in, double *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)in[i];
}
__global__ void k_f64_to_f32(const double *__restrict__ in, float *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

// o = b + A f in double, the reference's order (AdaptedFeatureVector::calculate_new_ada_vector,
// aku/ModelModules.hh:208-212)
__global__ void k_affine_frames_f64(const double *__restrict__ x, int64_t F, int dim, const double *__restrict__ A,
                                    const double *__restrict__ b, double *__restrict__ y) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= F * dim) return;
  const int64_t f = idx / dim;
  const int i = (int)(idx - f * dim);
  double acc = b[i];
  for (int j = 0; j < dim; j++) acc += A[(size_t)i * dim + j] * x[f * dim + j];
  y[idx] = acc;
}

// Per-class model transforms under AASR_PREC_F64 (regression classes: ConstrainedMllr, aku/ModelModules.cc:164-232).
// Every component is an AdaptedGaussian of its class: g(A_c f + b_c) |det_c|, summed in COMPONENT order as
// Mixture::compute_likelihood does -- so the frames of every class are laid out [class][dimension][frame] and a
// record reads its class's values straight from there (coalesced over the lanes, one load per dimension and record):
// a verification mode, an order of magnitude slower than the single-transform kernel.
__global__ void k_affine_frames_f64_classes(const double *__restrict__ x, int64_t F, int dim, int classes,
                                            const double *__restrict__ A, const double *__restrict__ b,
                                            double *__restrict__ y) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (class, i, f), f fastest
  if (idx >= (int64_t)classes * dim * F) return;
  const int64_t f = idx % F;
  const int64_t ci = idx / F;
  const int i = (int)(ci % dim), c = (int)(ci / dim);
  if (c == 0) {
    y[idx] = x[f * dim + i];
    return;
  }
  // o = b + A f in the reference's order (AdaptedFeatureVector::calculate_new_ada_vector, aku/ModelModules.hh:208-212)
  const double *Ac = A + ((size_t)c * dim + i) * dim;
  double acc = b[(size_t)c * dim + i];
  for (int j = 0; j < dim; j++) acc += Ac[j] * x[f * dim + j];
  y[idx] = acc;
}

template <int DIMP, bool CL>
__global__ __launch_bounds__(256) void k_gmm_diag_score_f64_classes(
    const double *__restrict__ xc, int64_t F, int dim, const double *__restrict__ recs,
    const int32_t *__restrict__ rec_class, const double *__restrict__ class_det,
    const int32_t *__restrict__ state_off, int64_t S, double *__restrict__ out, int linear,
    const int32_t *__restrict__ crow, const unsigned long long *__restrict__ maskw, int c1,
    const double *__restrict__ ll64, int64_t Cs, int C) {
  constexpr int REC = 2 * DIMP + 2;
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t fc = f < F ? f : F - 1;
  const int64_t s_per = (S + gridDim.y - 1) / gridDim.y;
  const int64_t s_begin = (int64_t)blockIdx.y * s_per, s_end = min(S, s_begin + s_per);
  for (int64_t s = s_begin; s < s_end; s++) {
    const int r0 = state_off[s], r1 = state_off[s + 1];
    double l = 0;
    for (int r = r0; r < r1; r++) {
      const double *rec = recs + (size_t)r * REC;
      const int c = rec_class[r];
      const double *x = xc + (size_t)c * dim * F + fc;
      double ll = 0;
      for (int d = 0; d < dim; d++) {
        const double t = x[(size_t)d * F] - rec[d];
        ll += t * t * rec[DIMP + d];
      }
      ll *= -0.5;
      ll += rec[2 * DIMP];
      double lik = exp(ll) * class_det[c];
      if (CL) {  // as k_gmm_diag_score_f64: an unselected cluster's members take the (plain) centre's likelihood
        const int cc = crow[r];
        const bool on = (maskw[(fc >> 6) * c1 + cc] >> (fc & 63)) & 1ull;
        if (!on) {
          const double key = ll64[fc * Cs + (cc < C ? cc : 0)];
          lik = key > -1000.0 ? exp(key) : ldexp((key + 2000.0) * 4398046511104.0, -1074);
        }
      }
      l += rec[2 * DIMP + 1] * lik;
    }
    if (l < 1e-50) l = 1e-50;
    if (f < F) out[f * S + s] = linear ? l : log(l);
  }
}

// one pass of at most `n` frames: class frames, then the kernel (masked when the selection tables are given)
static void f64_classes_pass(aasr_gmm *g, const double *d_frames, int64_t n, double *d_out, int linear,
                             const int32_t *crow, const unsigned long long *maskw, int c1, const double *ll64,
                             int64_t Cs, int C, hipStream_t stream) {
  const int nc = g->f64_classes;
  g->f64_class_x.ensure((size_t)nc * g->dim * (size_t)n);
  const int64_t nv = (int64_t)nc * g->dim * n;
  hipLaunchKernelGGL(k_affine_frames_f64_classes, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, d_frames, n,
                     g->dim, nc, g->f64_class_A.p, g->f64_class_b.p, g->f64_class_x.p);
  AASR_HIP(hipGetLastError());
  const int64_t blocks = (n + 255) / 256;
  int64_t cuts = std::max<int64_t>(1, std::min<int64_t>(g->S, (4 * (int64_t)(g->num_cus > 0 ? g->num_cus : 256) + blocks - 1) / blocks));
  if (cuts > 65535) cuts = 65535;
#define AASR_ARGS g->f64_class_x.p, n, g->dim, g->f64_recs.p, g->f64_rec_class.p, g->f64_class_det.p, \
                  g->f64_state_off.p, g->S, d_out, linear, crow, maskw, c1, ll64, Cs, C
#define AASR_CASE(N)                                                                                              \
  case N:                                                                                                         \
    if (maskw)                                                                                                    \
      hipLaunchKernelGGL((k_gmm_diag_score_f64_classes<N, true>), dim3((unsigned)blocks, (unsigned)cuts), dim3(256), 0, stream, AASR_ARGS); \
    else                                                                                                          \
      hipLaunchKernelGGL((k_gmm_diag_score_f64_classes<N, false>), dim3((unsigned)blocks, (unsigned)cuts), dim3(256), 0, stream, AASR_ARGS); \
    break;
  switch (g->f64_dimp) {
    AASR_CASE(8) AASR_CASE(16) AASR_CASE(24) AASR_CASE(32) AASR_CASE(40) AASR_CASE(48) AASR_CASE(64)
    default:
      raise(AASR_ERR_UNSUPPORTED, "no f64 kernel instance for dimension %d", g->dim);
  }
#undef AASR_CASE
#undef AASR_ARGS
  AASR_HIP(hipGetLastError());
}

static void score_f64_classes_launch(aasr_gmm *g, const double *d_frames, int64_t F, double *d_out, int linear,
                                     hipStream_t stream) {
  // passes of at most ~1 GB of class frames
  int64_t pass = std::max<int64_t>(256, (int64_t)(1.0e9 / ((double)g->f64_classes * g->dim * 8)));
  if (pass > F) pass = F;
  for (int64_t f0 = 0; f0 < F; f0 += pass) {
    const int64_t n = std::min(pass, F - f0);
    f64_classes_pass(g, d_frames + f0 * g->dim, n, d_out + f0 * g->S, linear, nullptr, nullptr, 0, nullptr, 0, 0, stream);
  }
}

// clustered sub-pass under per-class transforms (called by gmm_cluster_score_f64_launch with the RAW frames)
void gmm_f64_classes_masked_launch(aasr_gmm *g, const double *d_frames, int64_t n, double *d_out, int linear,
                                   const int32_t *crow, const unsigned long long *maskw, int c1, const double *ll64,
                                   int64_t Cs, int C, hipStream_t stream) {
  gmm_build_f64(g);
  f64_classes_pass(g, d_frames, n, d_out, linear, crow, maskw, c1, ll64, Cs, C, stream);
}

void gmm_score_f64_launch(aasr_gmm *g, const double *d_frames, int64_t F, double *d_out, int linear,
                          hipStream_t stream) {
  if (F <= 0) return;
  if (g->host.any_full()) raise(AASR_ERR_UNSUPPORTED, "AASR_PREC_F64 is built for diagonal pools");
  if (!g->dim_parts.empty() && (g->cl.enabled || (g->host.n_transforms > 0 && !g->host.global_xform())))
    raise(AASR_ERR_UNSUPPORTED, "AASR_PREC_F64 with clustering or regression classes is built for feature dimensions <= 63");
  gmm_build_f64(g);
  if (g->f64_classes > 0) {
    if (g->cl.enabled) gmm_cluster_score_f64_launch(g, d_frames, d_frames, F, d_out, linear, 1.0, stream);
    else score_f64_classes_launch(g, d_frames, F, d_out, linear, stream);
    return;
  }
  const double *d_raw = d_frames;
  double det = 1.0;
  if (g->host.n_transforms > 0) {  // one global transform: adapted frames, |prod diag A| on every Gaussian
    const int64_t nv = F * g->dim;
    g->f64_xframes.ensure((size_t)nv);
    hipLaunchKernelGGL(k_affine_frames_f64, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, d_frames, F,
                       g->dim, g->f64_A.p, g->f64_b.p, g->f64_xframes.p);
    AASR_HIP(hipGetLastError());
    d_frames = g->f64_xframes.p;
    det = g->f64_det;
  }
  if (g->cl.enabled) {
    gmm_cluster_score_f64_launch(g, d_raw, d_frames, F, d_out, linear, det, stream);
    return;
  }
  gmm_f64_masked_launch(g, d_frames, F, d_out, linear, det, nullptr, nullptr, 0, nullptr, 0, 0, stream);
}

void gmm_f64_masked_launch(aasr_gmm *g, const double *d_frames, int64_t F, double *d_out, int linear, double det,
                           const int32_t *crow, const unsigned long long *maskw, int c1, const double *ll64,
                           int64_t Cs, int C, hipStream_t stream) {
  gmm_build_f64(g);
  const int64_t blocks = (F + 255) / 256;
  // state-range cuts so that small batches still fill the chip
  int64_t cuts = std::max<int64_t>(1, std::min<int64_t>(g->S, (4 * (int64_t)(g->num_cus > 0 ? g->num_cus : 256) + blocks - 1) / blocks));
  if (cuts > 65535) cuts = 65535;
#define AASR_CASE(N)                                                                                              \
  case N:                                                                                                         \
    if (maskw)                                                                                                    \
      hipLaunchKernelGGL((k_gmm_diag_score_f64<N, true>), dim3((unsigned)blocks, (unsigned)cuts), dim3(256), 0,   \
                         stream, d_frames, F, g->dim, g->f64_recs.p, g->f64_state_off.p, g->S, d_out, linear, det, \
                         crow, maskw, c1, ll64, Cs, C);                                                           \
    else                                                                                                          \
      hipLaunchKernelGGL((k_gmm_diag_score_f64<N, false>), dim3((unsigned)blocks, (unsigned)cuts), dim3(256), 0,  \
                         stream, d_frames, F, g->dim, g->f64_recs.p, g->f64_state_off.p, g->S, d_out, linear, det, \
                         crow, maskw, c1, ll64, Cs, C);                                                           \
    break;
  // wide models (64 < dimension <= 192): the unmasked instance only
#define AASR_WIDE(N)                                                                                              \
  case N:                                                                                                         \
    if (maskw) raise(AASR_ERR_UNSUPPORTED, "AASR_PREC_F64 with clustering is built for feature dimensions <= 63"); \
    hipLaunchKernelGGL((k_gmm_diag_score_f64<N, false>), dim3((unsigned)blocks, (unsigned)cuts), dim3(256), 0,    \
                       stream, d_frames, F, g->dim, g->f64_recs.p, g->f64_state_off.p, g->S, d_out, linear, det,  \
                       crow, maskw, c1, ll64, Cs, C);                                                             \
    break;
  switch (g->f64_dimp) {
    AASR_CASE(8) AASR_CASE(16) AASR_CASE(24) AASR_CASE(32) AASR_CASE(40) AASR_CASE(48) AASR_CASE(64)
    AASR_WIDE(96) AASR_WIDE(128) AASR_WIDE(192)
    default:
      raise(AASR_ERR_UNSUPPORTED, "no f64 kernel instance for dimension %d", g->dim);
  }
#undef AASR_CASE
#undef AASR_WIDE
  AASR_HIP(hipGetLastError());
}

// float entry points under AASR_PREC_F64: frames widened, scores rounded once at the end
static void score_f64_for_f32_callers(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, hipStream_t stream) {
  const int64_t nx = F * g->dim, ns = F * g->S;
  g->f64_x.ensure((size_t)nx);
  g->f64_out.ensure((size_t)ns);
  hipLaunchKernelGGL(k_f32_to_f64, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, stream, d_frames, g->f64_x.p, nx);
  gmm_score_f64_launch(g, g->f64_x.p, F, g->f64_out.p, 0, stream);
  hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, g->f64_out.p, d_out, ns);
  AASR_HIP(hipGetLastError());
}

// scratch budget of the routed passes (outlier / class partial scores); tests shrink it to force
// many passes
static double g_pass_bytes = 1.0e9;
extern "C" void aasr_debug_set_pass_bytes(double bytes) { g_pass_bytes = bytes > 0 ? bytes : 1.0e9; }

// operand set of the centred kernel: the whole model, or the outlier components only
struct CentredOps {
  const float *recs;
  const int32_t *state_off, *splits;
  int max_splits;
  int64_t frame_stride, state_stride;  // out[f * frame_stride + s * state_stride]
  // Gaussian clustering: cluster of every record, selection bits [words][c1]; null = unmasked
  const int32_t *crow = nullptr;
  const unsigned long long *maskw = nullptr;
  int c1 = 0;
  int64_t n_words = 1;
  float floor_val = LOG_TINY_F;  // NEG_BIG_F: no floor (clustered passes, per-Gaussian view)
  int64_t n_recs = 0;            // records of the operand set (the launcher's cost model; 0: unknown)
};

template <int DIMP>
static void launch_centred_t(const aasr_gmm *g, const CentredOps &ops, const float *d_frames, int64_t F,
                             float *d_out, hipStream_t stream) {
  const int64_t blocks = (F + 511) / 512;
  // frame-major callers (state_stride == 1) get their values as runs of 32 states per frame row through LDS
  const int tile_out = ops.state_stride == 1 ? 1 : 0;
  const int smem = 4 * std::max(128 * (g->dim | 1), tile_out ? 512 * 17 : 0);
  // State-range cuts: a workgroup keeps its 512 frames in registers and walks the records of its state range, so a cut
  // costs every frame block its prologue again (frames through LDS, ~c records' time) -- R minimises
  // rounds x (records / R + c) over the workgroups the chip holds at once (LDS: two per CU with the output tile).
  const double slots = 4.0 * (g->num_cus > 0 ? g->num_cus : 256);
  const double c_fixed = 6.0;
  int R = 1;
  double best = 1e300;
  for (int r = 1; r <= ops.max_splits; r++) {
    const double rounds = std::ceil((double)blocks * r / slots);
    const double cost = rounds * ((double)std::max<int64_t>(1, ops.n_recs) / r + c_fixed);
    if (cost < best * 0.995) {
      best = cost;
      R = r;
    }
  }
  const int32_t *split = ops.splits + (size_t)(R - 1) * (CENTRED_MAX_SPLITS + 1);
  static bool attr_set[64][2] = {{false}};
  if (!attr_set[g->device & 63][ops.maskw ? 1 : 0]) {
    if (ops.maskw)
      AASR_HIP(hipFuncSetAttribute((const void *)k_gmm_diag_score_centred<DIMP, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 512 * 17));
    else
      AASR_HIP(hipFuncSetAttribute((const void *)k_gmm_diag_score_centred<DIMP, false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 512 * 17));
    attr_set[g->device & 63][ops.maskw ? 1 : 0] = true;
  }
  if (ops.maskw)
    hipLaunchKernelGGL((k_gmm_diag_score_centred<DIMP, true>), dim3((unsigned)blocks, (unsigned)R), dim3(256), smem,
                       stream, d_frames, F, g->dim, ops.recs, ops.state_off, split, d_out, ops.frame_stride,
                       ops.state_stride, ops.crow, ops.maskw, ops.c1, ops.n_words, NEG_BIG_F, tile_out);
  else
    hipLaunchKernelGGL((k_gmm_diag_score_centred<DIMP, false>), dim3((unsigned)blocks, (unsigned)R), dim3(256), smem,
                       stream, d_frames, F, g->dim, ops.recs, ops.state_off, split, d_out, ops.frame_stride,
                       ops.state_stride, (const int32_t *)nullptr, (const unsigned long long *)nullptr, 0, (int64_t)1,
                       ops.floor_val, tile_out);
  AASR_HIP(hipGetLastError());
}

static bool launch_centred_ops(const aasr_gmm *g, const CentredOps &ops, int dimp, const float *d_frames,
                               int64_t F, float *d_out, hipStream_t stream) {
  switch (dimp) {
#define AASR_CASE(N)                                             \
  case N:                                                        \
    launch_centred_t<N>(g, ops, d_frames, F, d_out, stream);     \
    return true;
    AASR_CASE(8) AASR_CASE(16) AASR_CASE(24) AASR_CASE(32) AASR_CASE(40) AASR_CASE(48) AASR_CASE(64)
#undef AASR_CASE
    default:
      return false;
  }
}

static bool launch_centred(const aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                           hipStream_t stream, int64_t pitch = 0) {
  CentredOps ops{g->centred_recs.p, g->centred_state_off.p, g->centred_splits.p, g->centred_max_splits,
                 pitch > 0 ? pitch : g->S, 1};
  ops.n_recs = (int64_t)g->host.mix_idx.size();
  return launch_centred_ops(g, ops, g->centred_dimp, d_frames, F, d_out, stream);
}

// out[f][map[j]] = log(exp(out[f][map[j]]) + exp(part[j][f])): the matrix path's sum over a
// state's well-conditioned components plus the centred sum over its outliers.  `part` is
// state-major ([Sb][pitch]: the centred kernel's lane = frame stores are coalesced that way); a
// workgroup moves a 64 x 64 tile through LDS so that the update of `out` walks along a frame row.
// Both inputs carry the 1e-50 floor, which the result keeps (floors = 1); in a clustered pass neither
// does (floors = 0).
__global__ __launch_bounds__(256) void k_outlier_merge(float *__restrict__ out, int64_t S,
                                                       const float *__restrict__ part, int64_t pitch,
                                                       int64_t Sb, const int32_t *__restrict__ map,
                                                       int64_t F, int floors, float part_bias) {
  __shared__ float tile[64][65];
  const int64_t f0 = (int64_t)blockIdx.x * 64;
  const int64_t j0 = (int64_t)blockIdx.y * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int jj = w; jj < 64; jj += 4) {
    const int64_t j = j0 + jj, f = f0 + lane;
    tile[jj][lane] = (j < Sb && f < F) ? part[j * pitch + f] : LOG_TINY_F;
  }
  __syncthreads();
  const int64_t j = j0 + lane;
  if (j >= Sb) return;
  const int col = map[j];
  for (int ff = w; ff < 64; ff += 4) {
    const int64_t f = f0 + ff;
    if (f >= F) break;
    float *o = out + f * S + col;
    // part_bias: log|det| of an in-place global transform -- the matrix path carries it at its output, the centred
    // records do not (a part AT the floor holds nothing and stays there)
    float b = tile[lane][ff];
    if (!floors || b > LOG_TINY_F) b += part_bias;
    const float a = *o;
    const float hi = fmaxf(a, b), lo = fminf(a, b);
    float r = hi;
    if (floors) {
      r = merge_floored_shares(a, b);  // a part AT the floor holds nothing
    } else {
      r = hi + log1pf(expf(lo - hi));  // clustered pass: no floors before k_cluster_merge
    }
    *o = r;
  }
}

// Outlier routing (gmm.h): the outlier components of the states that have any, in the centred
// form, merged into the scores the matrix path has already written.
static void score_outliers(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, hipStream_t stream,
                           const int32_t *crow = nullptr, const unsigned long long *maskw = nullptr, int c1 = 0,
                           int64_t n_words = 1, int64_t pitch = 0) {
  if (pitch <= 0) pitch = g->S;
  const int64_t Sb = g->hyb_states;
  if (Sb <= 0) return;
  // passes of at most ~1 GB of partial scores
  int64_t pass = std::max<int64_t>(512, ((int64_t)(g_pass_bytes / (double)(Sb * 4))) / 512 * 512);
  if (pass > F) pass = (F + 63) / 64 * 64;
  g->hyb_scratch.ensure((size_t)pass * (size_t)Sb);
  CentredOps ops{g->hyb_recs.p, g->hyb_state_off.p, g->hyb_splits.p, g->hyb_max_splits, 1, pass};
  ops.n_recs = g->hyb_rows;
  ops.crow = crow;
  ops.c1 = c1;
  for (int64_t f0 = 0; f0 < F; f0 += pass) {  // pass is a multiple of 512 frames: whole mask words
    const int64_t n = std::min(pass, F - f0);
    if (maskw) {
      ops.maskw = maskw + (f0 / 64) * c1;
      ops.n_words = n_words - f0 / 64;
    }
    if (!launch_centred_ops(g, ops, g->centred_dimp, d_frames + f0 * g->dim, n, g->hyb_scratch.p, stream))
      raise(AASR_ERR_UNSUPPORTED, "no centred kernel instance for dimension %d", g->dim);
    hipLaunchKernelGGL(k_outlier_merge, dim3((unsigned)((n + 63) / 64), (unsigned)((Sb + 63) / 64)), dim3(256), 0,
                       stream, d_out + f0 * pitch, pitch, g->hyb_scratch.p, pass, Sb, g->hyb_map.p, n, maskw ? 0 : 1,
                       (float)g->out_bias_ln);
    AASR_HIP(hipGetLastError());
  }
}

// Outlier routing with the merge inside the scoring kernel (k_gmm_diag_score_pl<..., HYB>): where the launch that follows
// is the grouped layout's two-term kernel, the outliers' partial sums of all F frames are formed first (state-major, the
// centred kernel's coalesced form) and put on the handle for the launcher; returns false where the merge pass has to do
// it (other layouts / precisions, clustering, more partial sums than a pass holds).
static bool hyb_fuse_begin(aasr_gmm *g, const TrackLayout &L, const float *d_frames, int64_t F, hipStream_t stream) {
  g->hyb_fuse = aasr_gmm::HybFuse();
  static const int fuse_env = AASR_EXPERIMENT_ENV("AASR_HYB_FUSE") ? atoi(AASR_EXPERIMENT_ENV("AASR_HYB_FUSE")) : 1;   // EXPERIMENT: 0 = merge pass
  const int64_t Sb = g->hyb_states;
  if (!fuse_env || !g->hyb_enabled || Sb <= 0 || !g->hyb_tab.p || g->cl.enabled || g->precision != AASR_PREC_F16X2 ||
      !g->use_bf16x3 || !L.ok || !L.grouped || !L.a16h.p || L.n_pg > 1 || !(g->layout_mask & 1) || L.nk16 <= 0)
    return false;
  const int64_t pass = (F + 63) / 64 * 64;
  if ((double)pass * (double)Sb * 4.0 > g_pass_bytes) return false;
  if ((size_t)pass * (size_t)Sb > g->hyb_scratch.n) {
    AASR_HIP(hipDeviceSynchronize());   // growing frees the old buffer
    g->hyb_scratch.ensure((size_t)pass * (size_t)Sb);
  }
  CentredOps ops{g->hyb_recs.p, g->hyb_state_off.p, g->hyb_splits.p, g->hyb_max_splits, 1, pass};
  ops.n_recs = g->hyb_rows;
  if (!launch_centred_ops(g, ops, g->centred_dimp, d_frames, F, g->hyb_scratch.p, stream)) return false;
  g->hyb_fuse.part = g->hyb_scratch.p;
  g->hyb_fuse.pitch = pass;
  return true;
}

template <int NKK, int MODE>
static void launch_t(const aasr_gmm *g, const PackedRows &pr, const float *d_frames,
                     int64_t F, float *d_out, int64_t out_cols, hipStream_t stream) {
  if (F <= 0) return;
  const int64_t blocks = (F + FRAMES_PER_BLOCK - 1) / FRAMES_PER_BLOCK;
  int smem = ScoreSmem<NKK>::kBytes;
  static const int dbg = AASR_EXPERIMENT_ENV("AASR_DBG") ? atoi(AASR_EXPERIMENT_ENV("AASR_DBG")) : 0;
  if (dbg & 2) smem = 100 * 1024;  // ablation: one workgroup per CU
  static bool attr_set[64] = {false};
  auto kern = k_gmm_diag_score<NKK, MODE>;
  if (!attr_set[g->device & 63]) {
    AASR_HIP(hipFuncSetAttribute((const void *)kern,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[g->device & 63] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, stream, d_frames, F,
                     g->dim, g->d_pivot.p, pr.a.p, pr.tiles, pr.chunk_seg_begin.p,
                     pr.seg_desc.p, pr.seg_out.p, d_out, out_cols, pr.rows, dbg);
  AASR_HIP(hipGetLastError());
}

template <int MODE>
static void launch(const aasr_gmm *g, const PackedRows &pr, const float *d_frames,
                   int64_t F, float *d_out, int64_t out_cols, hipStream_t stream) {
  switch (pr.nkk) {
#define AASR_CASE(N)                                                    \
  case N:                                                               \
    launch_t<N, MODE>(g, pr, d_frames, F, d_out, out_cols, stream);     \
    return;
    AASR_CASE(8) AASR_CASE(14) AASR_CASE(20) AASR_CASE(26) AASR_CASE(32) AASR_CASE(40)
    AASR_CASE(48) AASR_CASE(64)
#undef AASR_CASE
    default:
      break;
  }
  raise(AASR_ERR_UNSUPPORTED, "no kernel instance for K/2 = %d", pr.nkk);
}

__global__ void k_affine_frames(const float *__restrict__ x, int64_t F, int dim,
                                const double *__restrict__ A, const double *__restrict__ b,
                                float *__restrict__ y) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= F * dim) return;
  int64_t f = idx / dim;
  int i = (int)(idx - f * dim);
  double acc = b[i];
  for (int j = 0; j < dim; j++) acc += A[(size_t)i * dim + j] * (double)x[f * dim + j];
  y[idx] = (float)acc;
}

// Diagnostic (not part of the public ABI): restrict the kernels the scoring
// launcher may use (bit 0 grouped tracks, bit 1 independent tracks, bit 2 the
// centred-form kernel; with bits 0-1 clear and bit 2 set the centred kernel is
// forced, with all clear the general LDS-staged MFMA kernel).  Lets the tests
// cover every kernel.
extern "C" void aasr_debug_set_layouts(aasr_gmm *g, int mask) {
  if (!g) return;
  g->layout_mask = mask;
  if ((mask & 2) && !g->tracks.ok) gmm_build_tracks(g, false);
}
// which kernel a launch would use now: 4 centred, 1 grouped tracks,
// 2 independent tracks, 0 general LDS-staged MFMA kernel
extern "C" int aasr_debug_active_layout(const aasr_gmm *g) {
  if (!g) return -1;
  if ((g->layout_mask & 4) && (g->ill_conditioned || g->precision == AASR_PREC_F32_CENTRED ||
                               !(g->layout_mask & 3)) && g->centred_ok)
    return 4;
  if ((g->layout_mask & 1) && g->paired.ok) return 1;
  if ((g->layout_mask & 2) && g->tracks.ok) return 2;
  return 0;
}
extern "C" double aasr_debug_kappa(const aasr_gmm *g) { return g ? g->kappa : -1.0; }
// Diagnostic: the model's OWN one-pivot layout -- out[0] whole-model two-term rows packed, [1] outlier routing on, [2] outlier
// components, [3] states that hold them, [4] scored in the centred form as a whole, [5] states the probe moved
extern "C" void aasr_debug_own_layout(const aasr_gmm *g, int64_t *out) {
  if (!g || !out) return;
  const TrackLayout &L = g->paired.ok ? g->paired : g->tracks;
  out[0] = (L.ok && L.a16h.p) ? 1 : 0;
  out[1] = g->hyb_enabled ? 1 : 0;
  out[2] = g->hyb_rows;
  out[3] = g->hyb_states;
  out[4] = g->ill_conditioned ? 1 : 0;
  out[5] = g->f16_probe_moved;
}

// Diagnostic (tests, bench.py): the engine parts of a model (gmm_plan_engine_parts) -- out[0] parts, out[1] columns of an
// engine score row, then per part (up to three) {arithmetic (2 / 3 / 0: ordinary model), states, pivot groups, rows
// with the padding}; returns 0 when the model has none.
// Diagnostic: the column of every state in an engine score row ([S]) and, for engine part `part`, its pivot groups --
// first column / one past the last real column of every group (relative to the part's first column, which is returned in
// *col0) and the groups' pivots [groups][dim].  Returns the number of groups (0: an ordinary model), -1: no such part.
extern "C" int aasr_debug_engine_layout(const aasr_gmm *g, int part, int32_t *colmap, int32_t *begin, int32_t *real_end,
                                       float *pivots, int64_t *col0) {
  if (!g || part < 0 || (size_t)part >= g->engine_parts.size()) return -1;
  if (colmap) std::copy(g->engine_colmap_h.begin(), g->engine_colmap_h.end(), colmap);
  const auto &ep = g->engine_parts[(size_t)part];
  if (col0) *col0 = ep.col0;
  const aasr::HostModel &m = ep.model->host;
  const int P = m.n_pg();
  for (int p = 0; p < P; p++) {
    if (begin) begin[p] = m.pg_begin[(size_t)p];
    if (real_end) real_end[p] = m.pg_real_end[(size_t)p];
  }
  if (pivots) std::copy(m.pg_pivot.begin(), m.pg_pivot.end(), pivots);
  return P;
}
extern "C" const char *aasr_debug_engine_plan_note(const aasr_gmm *g) { return g ? g->engine_plan_note.c_str() : ""; }
extern "C" int aasr_debug_engine_parts(const aasr_gmm *g, int64_t *out, int n) {
  if (!g || !out || n < 14) return -1;
  for (int i = 0; i < n; i++) out[i] = 0;
  out[0] = (int64_t)g->engine_parts.size();
  out[1] = g->engine_cols;
  for (size_t i = 0; i < g->engine_parts.size() && i < 3; i++) {
    const auto &part = g->engine_parts[i];
    out[2 + 4 * i] = part.arith;
    out[3 + 4 * i] = part.states;
    out[4 + 4 * i] = part.arith ? part.model->paired.n_pg : 0;
    out[5 + 4 * i] = part.arith ? part.model->paired.rows_padded : (int64_t)part.model->host.mix_idx.size();
  }
  return (int)g->engine_parts.size();
}

// Diagnostic (bench.py): milliseconds of ONE k_frame_operand launch over F frames for the layout and arithmetic a scoring
// call would use now -- the launch that precedes k_gmm_diag_score_pl in every scoring call, so that the bench can price
// the scoring kernel on its own duration (HIP events around the call see both).  < 0: the current path forms its frame
// operand inside the kernel.
extern "C" double aasr_debug_frame_operand_ms(aasr_gmm *g, const float *d_frames, int64_t F, int reps, void *stream_v) {
  if (!g || F <= 0 || reps <= 0) return -1.0;
  hipStream_t stream = (hipStream_t)stream_v;
  const TrackLayout &L = g->paired.ok ? g->paired : g->tracks;
  if (!L.ok || !g->use_bf16x3 || g->precision != AASR_PREC_F16X2 || !L.a16h.p || g->cl.enabled) return -1.0;
  const int NW = F >= 8192 ? 8 : 4;
  const int64_t blocks64 = (F + NW * FRAMES_PER_WAVE - 1) / (NW * FRAMES_PER_WAVE) * NW;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  int64_t stride = 0;
  frame_operand<2>(g, L, d_frames, F, blocks64, stream, &stride);
  (void)hipEventRecord(e0, stream);
  for (int i = 0; i < reps; i++) frame_operand<2>(g, L, d_frames, F, blocks64, stream, &stride);
  (void)hipEventRecord(e1, stream);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return (double)ms / reps;
}

// Diagnostic (not part of the public ABI): resident workgroups per CU the
// runtime predicts for the NKK=40 scoring kernel.
extern "C" int aasr_debug_score_occupancy(void) {
  int nb = -1;
  auto kern = k_gmm_diag_score<40, 0>;
  (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                            ScoreSmem<40>::kBytes);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, 256,
                                                   ScoreSmem<40>::kBytes) != hipSuccess)
    return -1;
  return nb;
}

// The exact part of a clustered scoring pass: a track layout with the selection
// masks applied and no 1e-50 floor (k_cluster_merge adds the centre terms and
// floors).
// which == 0: grouped layout, 1: independent tracks (gmm_cluster_layout()).
// A f + b for every frame under the model's global transform (into the handle's scratch)
const float *gmm_adapted_frames(aasr_gmm *g, const float *d_frames, int64_t F, hipStream_t stream) {
  if (!g->xf_a.p) return d_frames;
  g->d_xframes.ensure((size_t)F * g->dim);
  const int64_t n = F * g->dim;
  hipLaunchKernelGGL(k_affine_frames, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_frames, F, g->dim,
                     g->xf_a.p, g->xf_b.p, g->d_xframes.p);
  AASR_HIP(hipGetLastError());
  return g->d_xframes.p;
}

void gmm_tracks_masked_launch(aasr_gmm *g, int which, const float *d_frames, int64_t F,
                              float *d_out, const unsigned long long *maskrow,
                              hipStream_t stream, int64_t pitch) {
  const TrackLayout &L = which == 0 ? g->paired : g->tracks;
  if (!L.ok) raise(AASR_ERR_UNSUPPORTED, "Gaussian clustering needs a track layout for this model");
  ClusterArgs cl;
  cl.maskrow = maskrow;
  cl.rows_padded = L.rows_padded;
  cl.floor_val = NEG_BIG_F;
  if (g->use_bf16x3 && launch_bf16(g, L, d_frames, F, d_out, stream, &cl, pitch)) return;
  if (!launch_tracks(g, L, d_frames, F, d_out, stream, &cl, pitch))
    raise(AASR_ERR_UNSUPPORTED, "no track kernel instance for this model");
}

// Clustered pass over a model with outlier-routed Gaussians: the outlier components in the centred
// form, masked by their clusters' selection bits, added to what the masked track kernel wrote.
void gmm_outliers_masked_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, const int32_t *crow,
                                const unsigned long long *maskw, int c1, int64_t n_words, hipStream_t stream) {
  score_outliers(g, d_frames, F, d_out, stream, crow, maskw, c1, n_words);
}

// Clustered pass over a model that is scored in the centred form as a whole.
void gmm_centred_masked_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, const int32_t *crow,
                               const unsigned long long *maskw, int c1, int64_t n_words, hipStream_t stream) {
  if (!g->centred_ok) raise(AASR_ERR_UNSUPPORTED, "no centred kernel instance for dimension %d", g->dim);
  CentredOps ops{g->centred_recs.p, g->centred_state_off.p, g->centred_splits.p, g->centred_max_splits, g->S, 1};
  ops.n_recs = (int64_t)g->host.mix_idx.size();
  ops.crow = crow;
  ops.maskw = maskw;
  ops.c1 = c1;
  ops.n_words = n_words;
  if (!launch_centred_ops(g, ops, g->centred_dimp, d_frames, F, d_out, stream))
    raise(AASR_ERR_UNSUPPORTED, "no centred kernel instance for dimension %d", g->dim);
}

// Class routing (gmm.h): out[i] = log(exp(out[i]) + exp(part[i] + logdet)); a value AT the floor
// holds nothing (a sub-model writes the floor for states it has no component of).
__global__ void k_class_merge(float *__restrict__ out, const float *__restrict__ part, float logdet,
                              int first, int64_t n, int floors) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!floors) {  // clustered pass: exact parts without floors, k_cluster_merge applies it
    const float b = part[i] + logdet;
    if (first) {
      out[i] = b;
    } else {
      const float a = out[i];
      const float hi = fmaxf(a, b), lo = fminf(a, b);
      out[i] = hi + log1pf(expf(lo - hi));
    }
    return;
  }
  const float a = first ? LOG_TINY_F : out[i];
  float b = part[i];
  float r = a;
  if (b > LOG_TINY_F) {
    b += logdet;
    if (a > LOG_TINY_F) {
      const float hi = fmaxf(a, b), lo = fminf(a, b);
      r = hi + log1pf(expf(lo - hi));
    } else {
      r = b;
    }
  }
  out[i] = fmaxf(r, LOG_TINY_F);
}

__global__ void k_fill_value(float *__restrict__ out, float v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

__global__ void k_fill_floor(float *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = LOG_TINY_F;
}

static void score_classes(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, hipStream_t stream) {
  const int64_t S = g->S;
  int64_t pass = std::max<int64_t>(64, (int64_t)(g_pass_bytes / (double)(S * 4)));
  if (pass > F) pass = F;
  g->class_scratch.ensure((size_t)pass * (size_t)S);
  g->class_xframes.ensure((size_t)pass * (size_t)g->dim);
  for (int64_t f0 = 0; f0 < F; f0 += pass) {
    const int64_t n = std::min(pass, F - f0);
    const float *fr = d_frames + f0 * g->dim;
    float *out = d_out + f0 * S;
    const int64_t total = n * S;
    bool first = true;
    for (size_t c = 0; c < g->class_models.size(); c++) {
      aasr_gmm *sub = g->class_models[c].get();
      if (!sub) continue;
      const double logdet = c == 0 ? 0.0 : g->class_logdet[c];
      if (!(logdet > -INFINITY)) continue;  // det == 0: the adapted Gaussians contribute nothing
      const float *xf = fr;
      if (c > 0) {
        // f' = A f + b (AdaptedFeatureVector::calculate_new_ada_vector, aku/ModelModules.hh:208-212)
        const int64_t nv = n * g->dim;
        hipLaunchKernelGGL(k_affine_frames, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, fr, n,
                           g->dim, g->class_a[c].p, g->class_b[c].p, g->class_xframes.p);
        AASR_HIP(hipGetLastError());
        xf = g->class_xframes.p;
      }
      sub->precision = g->precision;
      sub->use_bf16x3 = g->use_bf16x3;
      sub->layout_mask = g->layout_mask;
      gmm_score_launch(sub, xf, n, g->class_scratch.p, stream);
      hipLaunchKernelGGL(k_class_merge, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, out,
                         g->class_scratch.p, (float)logdet, first ? 1 : 0, total, 1);
      AASR_HIP(hipGetLastError());
      first = false;
    }
    if (first) {  // no class has anything: every state at the floor
      hipLaunchKernelGGL(k_fill_floor, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, out, total);
      AASR_HIP(hipGetLastError());
    }
  }
}

void gmm_classes_exact_launch(aasr_gmm *g, const float *d_frames, int64_t n, float *d_out,
                              const std::function<void(aasr_gmm *, size_t, const float *, float *)> &exact,
                              hipStream_t stream) {
  const int64_t S = g->S, total = n * S;
  g->class_scratch.ensure((size_t)total);
  g->class_xframes.ensure((size_t)n * (size_t)g->dim);
  bool first = true;
  for (size_t c = 0; c < g->class_models.size(); c++) {
    aasr_gmm *sub = g->class_models[c].get();
    if (!sub) continue;
    const double logdet = c == 0 ? 0.0 : g->class_logdet[c];
    if (!(logdet > -INFINITY)) continue;  // det == 0: the adapted Gaussians' exact values are 0
    const float *xf = d_frames;
    if (c > 0) {
      const int64_t nv = n * g->dim;
      hipLaunchKernelGGL(k_affine_frames, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, d_frames, n,
                         g->dim, g->class_a[c].p, g->class_b[c].p, g->class_xframes.p);
      AASR_HIP(hipGetLastError());
      xf = g->class_xframes.p;
    }
    exact(sub, c, xf, g->class_scratch.p);
    hipLaunchKernelGGL(k_class_merge, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_out,
                       g->class_scratch.p, (float)logdet, first ? 1 : 0, total, 0);
    AASR_HIP(hipGetLastError());
    first = false;
  }
  if (first) {  // no class has an exact part: nothing but the centres' share
    hipLaunchKernelGGL(k_fill_value, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_out, NEG_BIG_F, total);
    AASR_HIP(hipGetLastError());
  }
}

// log|det| of an in-place global transform for the kernels that do not take it at their output
// (diagnostic layouts only: the track kernels fold it into their reference exponent)
__global__ void k_add_bias(float *__restrict__ out, float bias, int64_t n, int64_t S, int64_t pitch) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t f = i / S;
  float *o = out + f * pitch + (i - f * S);
  *o = fmaxf(*o + bias, LOG_TINY_F);
}
__global__ void k_add_bias_nofloor(float *__restrict__ out, float bias, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += bias;
}
// (clustered passes: exact parts carry no floor before the merge)
void gmm_add_bias_nofloor(float *d_out, int64_t n, float bias, hipStream_t stream) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_add_bias_nofloor, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_out, bias, n);
  AASR_HIP(hipGetLastError());
}
static void add_output_bias(const aasr_gmm *g, float *d_out, int64_t F, hipStream_t stream, int64_t pitch = 0) {
  const int64_t n = F * g->S;
  hipLaunchKernelGGL(k_add_bias, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_out,
                     (float)g->out_bias_ln, n, g->S, pitch > 0 ? pitch : g->S);
  AASR_HIP(hipGetLastError());
}

// Whether scores can be written with a row pitch other than S: the bf16x3 track kernels can
// (rows padded to a multiple of 16 floats make every 64-byte output group a whole cache line).
static bool engine_parts_public(const aasr_gmm *g);
bool gmm_score_pitch_ok(const aasr_gmm *g) {
  if (!g->dim_parts.empty()) return false;
  if (engine_parts_public(g)) return true;
  if (g->cl.enabled && gmm_engine_parts_clustered(g)) return true;
  if (g->host.factor_path() || g->class_routing || g->precision == AASR_PREC_F64) return false;
  if (g->hyb_enabled && g->cl.enabled) return false;   // (the clustered pass over outlier routing writes dense rows)
  // a model scored in the centred form as a whole: the kernel writes runs of 32 states per frame row at any pitch
  if (g->ill_conditioned) return g->centred_ok && !g->cl.enabled && (g->layout_mask & 4);
  if (g->cl.enabled && !gmm_cluster_pitch_ok(g)) return false;
  if ((g->layout_mask & 3) != 3) return false;
  const TrackLayout &L = g->paired.ok ? g->paired : g->tracks;
  // both track kernels (f32 and bf16x3) take a row pitch; the centred kernel does not
  return (g->precision == AASR_PREC_F32 || g->precision == AASR_PREC_BF16X3 || g->precision == AASR_PREC_F16X2) && L.ok;
}

// ---------------------------------------------------------------------------
// The engine's own score layout (recipe driver, aasr_run_utterance, aasr_gmm_score_lna_dev: scores that only the LNA pass
// reads): rows padded to whole lines; a model with engine parts: every part in its own column range, read through a
// column map.
// ---------------------------------------------------------------------------
// Engine parts (gmm_plan_engine_parts): the model as internal multi-pivot models over disjoint sets of its states.  They
// carry the default arithmetic only -- the other precisions are verification modes on the model's own layouts -- and
// nothing that merges by state column (clustering, class routing).
bool gmm_engine_parts_active(const aasr_gmm *g) {
  return !g->engine_parts.empty() && g->precision == AASR_PREC_F16X2 && g->use_bf16x3 && !g->cl.enabled &&
         !g->class_routing && g->dim_parts.empty() && (g->layout_mask & 3) == 3;
}

bool gmm_engine_parts_clustered(const aasr_gmm *g) {
  return !g->engine_parts.empty() && g->cl.enabled && g->precision == AASR_PREC_F16X2 && g->use_bf16x3 && !g->class_routing &&
         g->dim_parts.empty() && (g->layout_mask & 3) == 3;
}

// ... and public-layout calls (column = state) go through them too, with the columns gathered back: the parts are planned
// only for models whose own one-pivot layouts cannot put every state on two fp16 terms, and what those layouts do with the
// rest -- three bf16 terms up to their limits, the centred form beyond -- is slower than the parts + a gather of the columns.
// The exception: a model whose own layout HAS every state on plain two-term rows and only a few Gaussians off the matrix
// path (outlier routing) -- the public layout then costs those few rows in the centred form and their states' merge, less
// than the gather of the whole matrix (measured per 449 280 frames: 1.6 us per centred row, ~20 us per merged state -- its
// column's read-modify-write touches a line per frame -- as a pass of its own, nothing where the scoring kernel merges in
// its close logic (k_gmm_diag_score_pl<..., HYB>), 2.4 ms for the gather): one far-out Gaussian in 1 % / 10 % / 40 % of the
// states of configs[2]: see DESIGN 4.2.
static bool engine_parts_public(const aasr_gmm *g) {
  if (!gmm_engine_parts_active(g)) return false;
  static const int force_sc = AASR_EXPERIMENT_ENV("AASR_EXP_FORCE_SC") ? atoi(AASR_EXPERIMENT_ENV("AASR_EXP_FORCE_SC")) : 0;   // EXPERIMENT
  if (force_sc) return true;
  const TrackLayout &L = g->paired.ok ? g->paired : g->tracks;
  if (L.ok && L.a16h.p && g->hyb_enabled && !g->ill_conditioned && !g->cl.enabled) {
    // (the scoring kernel merges the outliers' sums in its close logic where it can: hyb_fuse_begin)
    // (... while such states are sparse: where they follow each other within a tile's time the close logic waits for its
    // values -- 40 % of the states of configs[2]: 15 ms against 11 through the parts)
    const bool fusable = g->hyb_tab.p && L.grouped && (g->layout_mask & 1) && 8 * g->hyb_states <= g->S;
    if (fusable ? 1.7 * (double)g->hyb_rows < 2400.0
                : 1.7 * (double)g->hyb_rows + 20.0 * (double)g->hyb_states < 1500.0)
      return false;
  }
  return true;
}

int64_t gmm_engine_pitch(const aasr_gmm *g) {
  // the parts' pitch only while the parts are what a scoring call runs: under another precision (verification modes on
  // the model's own layouts) the model's own rule holds (the parts' pitch there sent outlier-routed and
  // centred models into a pitched launch they do not have).  Scratch is sized with gmm_engine_pitch_max.
  if (gmm_engine_parts_active(g) || gmm_engine_parts_clustered(g)) return std::max(g->engine_cols, (g->S + 31) / 32 * 32);
  if (!gmm_score_pitch_ok(g)) return g->S;
  return (g->S + 31) / 32 * 32;
}

// the largest row pitch gmm_engine_pitch() can return for this model whatever the precision, clustering or transform
// state: what a caller sizes its scratch with (aasr_gmm_score_scratch_floats)
int64_t gmm_engine_pitch_max(const aasr_gmm *g) {
  return std::max((g->S + 31) / 32 * 32, g->engine_cols);
}

const int32_t *gmm_engine_colmap(const aasr_gmm *g) {
  return gmm_engine_parts_active(g) ? g->engine_colmap.p : nullptr;
}

__global__ void k_scatter_columns(const float *__restrict__ in, int64_t F, int64_t n, float *__restrict__ out, int64_t pitch) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * n) return;
  const int64_t f = i / n;
  out[f * pitch + (i - f * n)] = in[i];
}

// out[f][s] = rows[f][colmap[s]]: the engine's score rows back in the public layout.  A workgroup walks frames; a thread
// keeps the columns of its states (s = tid, tid + 256, ...) in registers, so a frame costs one gather of its row (L2: the
// row was written a moment ago) and coalesced stores.
template <int VPT>
__global__ __launch_bounds__(256) void k_gather_columns(const float *__restrict__ rows, int64_t F, int64_t in_pitch,
                                                        const int32_t *__restrict__ colmap, int64_t S,
                                                        float *__restrict__ out, int64_t out_pitch) {
  int col[VPT];
#pragma unroll
  for (int j = 0; j < VPT; j++) {
    const int64_t s = threadIdx.x + 256 * j;
    col[j] = s < S ? colmap[s] : -1;
  }
  for (int64_t f = blockIdx.x; f < F; f += gridDim.x) {
    const float *r = rows + f * in_pitch;
    float *o = out + f * out_pitch;
    float v[VPT];
#pragma unroll
    for (int j = 0; j < VPT; j++) v[j] = col[j] >= 0 ? r[col[j]] : 0.0f;
#pragma unroll
    for (int j = 0; j < VPT; j++)
      if (col[j] >= 0) o[threadIdx.x + 256 * j] = v[j];
  }
}

__global__ __launch_bounds__(256) void k_gather_columns_any(const float *__restrict__ rows, int64_t F, int64_t in_pitch,
                                                            const int32_t *__restrict__ colmap, int64_t S,
                                                            float *__restrict__ out, int64_t out_pitch) {
  for (int64_t f = blockIdx.x; f < F; f += gridDim.x)
    for (int64_t s = threadIdx.x; s < S; s += 256) out[f * out_pitch + s] = rows[f * in_pitch + colmap[s]];
}

static void launch_gather_columns(const aasr_gmm *g, const float *rows, int64_t F, int64_t in_pitch, float *out,
                                  int64_t out_pitch, hipStream_t stream) {
  const unsigned blocks = (unsigned)std::min<int64_t>(F, (int64_t)(g->num_cus > 0 ? g->num_cus : 256) * 16);
  if (g->S <= 256 * 4)
    hipLaunchKernelGGL(k_gather_columns<4>, dim3(blocks), dim3(256), 0, stream, rows, F, in_pitch, g->engine_colmap.p, g->S, out, out_pitch);
  else if (g->S <= 256 * 16)
    hipLaunchKernelGGL(k_gather_columns<16>, dim3(blocks), dim3(256), 0, stream, rows, F, in_pitch, g->engine_colmap.p, g->S, out, out_pitch);
  else
    hipLaunchKernelGGL(k_gather_columns_any, dim3(blocks), dim3(256), 0, stream, rows, F, in_pitch, g->engine_colmap.p, g->S, out, out_pitch);
  AASR_HIP(hipGetLastError());
}

void gmm_scatter_columns(const float *dense, int64_t F, int64_t n, float *out, int64_t pitch, hipStream_t stream) {
  const int64_t tot = F * n;
  if (tot <= 0) return;
  hipLaunchKernelGGL(k_scatter_columns, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, dense, F, n, out, pitch);
  AASR_HIP(hipGetLastError());
}
void gmm_gather_engine_columns(const aasr_gmm *g, const float *rows, int64_t F, int64_t in_pitch, float *out, int64_t out_pitch,
                               hipStream_t stream) {
  launch_gather_columns(g, rows, F, in_pitch, out, out_pitch, stream);
}

static void launch_engine_parts(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, int64_t pitch,
                                hipStream_t stream) {
  if (g->xf_a.p) {   // one constrained-MLLR transform for the pool: the frames transformed once, log|det| at the output
    if ((size_t)F * g->dim > g->d_xframes.n) AASR_HIP(hipDeviceSynchronize());
    g->d_xframes.ensure((size_t)F * g->dim);
    const int64_t n = F * g->dim;
    hipLaunchKernelGGL(k_affine_frames, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_frames, F, g->dim,
                       g->xf_a.p, g->xf_b.p, g->d_xframes.p);
    AASR_HIP(hipGetLastError());
    d_frames = g->d_xframes.p;
  }
  for (auto &part : g->engine_parts) {
    aasr_gmm *sub = part.model.get();
    sub->out_bias_ln = g->out_bias_ln;
    float *o = d_out + part.col0;
    bool done = false;
    if (part.arith == 2 || part.arith == 4) done = launch_split<2>(sub, sub->paired, d_frames, F, o, stream, nullptr, pitch);
    else if (part.arith == 3) done = launch_split<3>(sub, sub->paired, d_frames, F, o, stream, nullptr, pitch);
    else {
      sub->precision = g->precision;
      sub->use_bf16x3 = g->use_bf16x3;
      if (gmm_score_pitch_ok(sub)) {
        gmm_score_launch_pitched(sub, d_frames, F, o, pitch, stream);
      } else {   // (an ill-conditioned remainder: its kernels write dense rows)
        if ((size_t)F * sub->S > g->engine_part_scratch.n) AASR_HIP(hipDeviceSynchronize());
        g->engine_part_scratch.ensure((size_t)F * sub->S);
        gmm_score_launch(sub, d_frames, F, g->engine_part_scratch.p, stream);
        const int64_t n = F * sub->S;
        hipLaunchKernelGGL(k_scatter_columns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                           g->engine_part_scratch.p, F, sub->S, o, pitch);
        AASR_HIP(hipGetLastError());
      }
      done = true;
    }
    if (!done) raise(AASR_ERR_UNSUPPORTED, "no kernel instance for an engine part of the model");
  }
}

// public layout through the engine parts: chunks of frames into the handle's scratch, columns gathered back
static void launch_engine_parts_public(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, int64_t pitch,
                                       hipStream_t stream) {
  const int64_t ep = gmm_engine_pitch(g);
  const int64_t chunk = std::max<int64_t>(512, std::min<int64_t>(F, (int64_t)(1.0e9 / (4.0 * (double)ep))) / 512 * 512);
  if ((size_t)std::min(chunk, F) * ep > g->engine_scratch.n) {
    AASR_HIP(hipDeviceSynchronize());
    g->engine_scratch.ensure((size_t)std::min(chunk, F) * ep);
  }
  for (int64_t f0 = 0; f0 < F; f0 += chunk) {
    const int64_t fc = std::min(chunk, F - f0);
    launch_engine_parts(g, d_frames + f0 * g->dim, fc, g->engine_scratch.p, ep, stream);
    launch_gather_columns(g, g->engine_scratch.p, fc, ep, d_out + f0 * pitch, pitch, stream);
  }
}

void gmm_score_launch_engine(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, int64_t pitch,
                             hipStream_t stream) {
  if (F <= 0) return;
  if (gmm_engine_parts_active(g) && pitch >= g->engine_cols) {
    launch_engine_parts(g, d_frames, F, d_out, pitch, stream);
    return;
  }
  gmm_score_launch_pitched(g, d_frames, F, d_out, pitch, stream);
}

void gmm_score_launch_pitched(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, int64_t pitch,
                              hipStream_t stream) {
  if (F <= 0) return;
  if (pitch == g->S) {
    gmm_score_launch(g, d_frames, F, d_out, stream);
    return;
  }
  if (pitch > g->S && engine_parts_public(g)) {
    launch_engine_parts_public(g, d_frames, F, d_out, pitch, stream);
    return;
  }
  if (pitch < g->S || !gmm_score_pitch_ok(g))
    raise(AASR_ERR_UNSUPPORTED, "a row pitch other than the state count needs the track kernels");
  if (g->cl.enabled) {
    gmm_cluster_score_launch(g, d_frames, F, d_out, stream, pitch);
    return;
  }
  if (g->xf_a.p) {
    g->d_xframes.ensure((size_t)F * g->dim);
    const int64_t n = F * g->dim;
    hipLaunchKernelGGL(k_affine_frames, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       d_frames, F, g->dim, g->xf_a.p, g->xf_b.p, g->d_xframes.p);
    AASR_HIP(hipGetLastError());
    d_frames = g->d_xframes.p;
  }
  if (g->ill_conditioned) {
    if (!launch_centred(g, d_frames, F, d_out, stream, pitch))
      raise(AASR_ERR_UNSUPPORTED, "no centred kernel instance for dimension %d", g->dim);
    if (g->out_bias_ln != 0) add_output_bias(g, d_out, F, stream, pitch);
    return;
  }
  const TrackLayout &L = g->paired.ok ? g->paired : g->tracks;
  const bool fused = hyb_fuse_begin(g, L, d_frames, F, stream);
  const bool done = (g->use_bf16x3 && launch_bf16(g, L, d_frames, F, d_out, stream, nullptr, pitch)) ||
                    launch_tracks(g, L, d_frames, F, d_out, stream, nullptr, pitch);
  g->hyb_fuse = aasr_gmm::HybFuse();
  if (!done) raise(AASR_ERR_UNSUPPORTED, "no track kernel instance for this model");
  // the Gaussians the matrix layouts left out (null rows): centred form, merged per state into the padded rows (where
  // the scoring kernel has not merged them itself)
  if (g->hyb_enabled && !fused) score_outliers(g, d_frames, F, d_out, stream, nullptr, nullptr, 0, 1, pitch);
}

// ---------------------------------------------------------------------------
// Feature dimension > 63: the model as parts of <= 63 dimensions (aasr_gmm::dim_parts).  ll_g(x) = sum over the
// parts of the part's own diagonal Gaussian (constants included: log sqrt prod p factorises too), so every part
// scores its column range per Gaussian -- no floor, natural log -- and k_dim_split_combine adds them per mixture
// component and forms log max(sum_k w_k e^ll, 1e-50) with a running maximum.  Frames are taken in chunks so that the
// per-Gaussian scores of all parts stay below ~1.5 GB.
// ---------------------------------------------------------------------------
__global__ void k_slice_columns(const float *__restrict__ x, int64_t F, int dim, int d0, int dp, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * dp) return;
  const int64_t f = i / dp;
  const int d = (int)(i - f * dp);
  out[i] = x[f * dim + d0 + d];
}

__global__ __launch_bounds__(256) void k_dim_split_combine(const float *__restrict__ part_ll, int parts, int64_t Fc,
                                                           int64_t G, const int32_t *__restrict__ mix_off,
                                                           const int32_t *__restrict__ mix_idx,
                                                           const float *__restrict__ mix_logw, int64_t S,
                                                           float *__restrict__ out, int per_gaussian, float bias_ln,
                                                           const unsigned long long *__restrict__ maskw, int c1,
                                                           const int32_t *__restrict__ gclus, int64_t f_first) {
  const int64_t n = per_gaussian ? G : S;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Fc * n) return;
  const int64_t f = i / n, s = i - f * n;
  if (per_gaussian) {
    float v = 0.0f;
    for (int p = 0; p < parts; p++) v += part_ll[((int64_t)p * Fc + f) * G + s];
    out[i] = v;
    return;
  }
  // the parts' log-likelihoods are added in double: at several hundred dimensions |ll| ~ 10^3, where a float sum's
  // rounding alone (6e-5 per addition) reaches the 1e-4 bar; only differences to the running maximum go through expf
  double m = -3.0e38, sum = 0.0;
  for (int32_t k = mix_off[s]; k < mix_off[s + 1]; k++) {
    const float lw = mix_logw[k];
    if (!(lw > NEG_BIG_F)) continue;   // zero weight
    double v = (double)lw + (double)bias_ln;   // log |det| of the pool's one constrained-MLLR transform (AdaptedGaussian), else 0
    const int64_t gi = mix_idx[k];
    if (maskw) {   // Gaussian clustering: only the members of clusters selected for this frame are evaluated exactly
      const int64_t fa = f_first + f;
      if (!((maskw[(fa >> 6) * c1 + gclus[gi]] >> (fa & 63)) & 1ull)) continue;
    }
    for (int p = 0; p < parts; p++) v += (double)part_ll[((int64_t)p * Fc + f) * G + gi];
    const double mn = fmax(m, v);
    sum = sum * (double)__expf((float)(m - mn)) + (double)__expf((float)(v - mn));
    m = mn;
  }
  if (maskw) {   // the exact part alone, no floor: the merge adds the centres' share
    out[i] = sum > 0.0 ? (float)(m + (double)__logf((float)sum)) : NEG_BIG_F;
    return;
  }
  const float ll = sum > 0.0 ? (float)(m + (double)__logf((float)sum)) : LOG_TINY_F;
  out[i] = fmaxf(ll, LOG_TINY_F);
}

void gmm_dim_split_score(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, bool per_gaussian,
                         hipStream_t stream, const unsigned long long *maskw, int c1, const int32_t *gclus,
                         bool frames_adapted) {
  const int parts = (int)g->dim_parts.size();
  const int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(F, (int64_t)(1.5e9 / (4.0 * (double)g->G * parts))));
  int max_dp = 0;
  for (int p = 0; p < parts; p++) max_dp = std::max(max_dp, g->dim_part_off[(size_t)p + 1] - g->dim_part_off[(size_t)p]);
  if ((size_t)chunk * max_dp > g->dim_part_x.n || (size_t)parts * chunk * g->G > g->dim_part_ll.n)
    AASR_HIP(hipDeviceSynchronize());   // growing frees the old buffers
  g->dim_part_x.ensure((size_t)chunk * max_dp);
  g->dim_part_ll.ensure((size_t)parts * chunk * g->G);
  const int64_t n_out = per_gaussian ? g->G : g->S;
  if (g->xf_a.p && !frames_adapted) {
    // one transform for the whole pool: the Gaussians are evaluated on A f + b (k_affine_frames handles any dimension)
    if (per_gaussian) raise(AASR_ERR_UNSUPPORTED, "per-Gaussian log-likelihoods are not built for adapted pools");
    g->d_xframes.ensure((size_t)F * g->dim);
    const int64_t n = F * g->dim;
    hipLaunchKernelGGL(k_affine_frames, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_frames, F, g->dim,
                       g->xf_a.p, g->xf_b.p, g->d_xframes.p);
    AASR_HIP(hipGetLastError());
    d_frames = g->d_xframes.p;
  }
  for (int64_t f0 = 0; f0 < F; f0 += chunk) {
    const int64_t fc = std::min(chunk, F - f0);
    for (int p = 0; p < parts; p++) {
      const int d0 = g->dim_part_off[(size_t)p], dp = g->dim_part_off[(size_t)p + 1] - d0;
      const int64_t n = fc * dp;
      hipLaunchKernelGGL(k_slice_columns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_frames + f0 * g->dim,
                         fc, g->dim, d0, dp, g->dim_part_x.p);
      AASR_HIP(hipGetLastError());
      g->dim_parts[(size_t)p]->precision = g->precision == AASR_PREC_F64 ? AASR_PREC_F32 : g->precision;
      gmm_gauss_launch(g->dim_parts[(size_t)p].get(), g->dim_part_x.p, fc, g->dim_part_ll.p + (size_t)p * fc * g->G, stream);
    }
    const int64_t n = fc * n_out;
    hipLaunchKernelGGL(k_dim_split_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g->dim_part_ll.p, parts,
                       fc, g->G, g->dim_mix_off.p, g->dim_mix_idx.p, g->dim_mix_logw.p, g->S, d_out + f0 * n_out,
                       per_gaussian ? 1 : 0, g->xf_a.p ? (float)g->out_bias_ln : 0.0f, maskw, c1, gclus, f0);
    AASR_HIP(hipGetLastError());
  }
}

void gmm_score_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                      hipStream_t stream) {
  if (F <= 0) return;
  if (!g->dim_parts.empty()) {
    if (g->cl.enabled) {
      if (g->precision == AASR_PREC_F64)
        raise(AASR_ERR_UNSUPPORTED, "Gaussian clustering on a model of more than 63 dimensions: float arithmetic only");
      gmm_cluster_score_launch(g, d_frames, F, d_out, stream);
      return;
    }
    if (g->precision == AASR_PREC_F64) {   // the reference's arithmetic in double: instances up to 192 dimensions
      score_f64_for_f32_callers(g, d_frames, F, d_out, stream);
      return;
    }
    gmm_dim_split_score(g, d_frames, F, d_out, false, stream);
    return;
  }
  if (g->precision == AASR_PREC_F64) {
    score_f64_for_f32_callers(g, d_frames, F, d_out, stream);
    return;
  }
  if (g->cl.enabled) {
    gmm_cluster_score_launch(g, d_frames, F, d_out, stream);
    return;
  }
  if (g->class_routing) {
    score_classes(g, d_frames, F, d_out, stream);
    return;
  }
  if (g->host.factor_path()) {
    gmm_full_launch(g, d_frames, F, d_out, stream);
    return;
  }
  if (engine_parts_public(g)) {
    launch_engine_parts_public(g, d_frames, F, d_out, g->S, stream);
    return;
  }
  if (g->xf_a.p) {
    // global constrained-MLLR transform: f' = A f + b once per frame
    // (AdaptedFeatureVector::calculate_new_ada_vector, aku/ModelModules.hh:208-212)
    g->d_xframes.ensure((size_t)F * g->dim);
    const int64_t n = F * g->dim;
    hipLaunchKernelGGL(k_affine_frames, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       d_frames, F, g->dim, g->xf_a.p, g->xf_b.p, g->d_xframes.p);
    AASR_HIP(hipGetLastError());
    d_frames = g->d_xframes.p;
  }
  // numerically safe path first when the model needs it (or it is forced)
  if ((g->layout_mask & 4) && (g->ill_conditioned || g->precision == AASR_PREC_F32_CENTRED ||
                               !(g->layout_mask & 3) ) && g->centred_ok)
    if (launch_centred(g, d_frames, F, d_out, stream)) {
      if (g->out_bias_ln != 0) add_output_bias(g, d_out, F, stream);
      return;
    }
  // layout choice: grouped tracks > independent tracks > general (LDS-staged)
  bool done = false;
  bool fused = false;
  if (!done && (g->layout_mask & 1) && g->paired.ok) {
    fused = hyb_fuse_begin(g, g->paired, d_frames, F, stream);
    done = (g->use_bf16x3 && launch_bf16(g, g->paired, d_frames, F, d_out, stream)) ||
           launch_tracks(g, g->paired, d_frames, F, d_out, stream);
    g->hyb_fuse = aasr_gmm::HybFuse();
  }
  if (!done && (g->layout_mask & 2) && g->tracks.ok)
    done = (g->use_bf16x3 && launch_bf16(g, g->tracks, d_frames, F, d_out, stream)) ||
           launch_tracks(g, g->tracks, d_frames, F, d_out, stream);
  if (!done) {
    launch<0>(g, g->mix, d_frames, F, d_out, g->S, stream);
    if (g->out_bias_ln != 0) add_output_bias(g, d_out, F, stream);  // this kernel has no output bias
  }
  // the Gaussians the matrix layouts left out (null rows): centred form, merged per state
  if (g->hyb_enabled && !fused) score_outliers(g, d_frames, F, d_out, stream);
}

void gmm_gauss_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                      hipStream_t stream) {
  if (!g->dim_parts.empty()) {
    gmm_dim_split_score(g, d_frames, F, d_out, true, stream);
    return;
  }
  if (g->xf_a.p || g->class_routing || g->host.n_transforms > 0)
    raise(AASR_ERR_UNSUPPORTED, "per-Gaussian log-likelihoods are not built for adapted pools");
  if (g->host.any_full()) {
    // full-covariance pools: every Gaussian as a one-component state of an internal model, scored
    // by the factor-row kernel (values below the 1e-50 floor come out at the floor)
    if (!g->pool_view) {
      HostModel pm = g->host;
      pm.S = pm.G;
      pm.mix_off.resize((size_t)pm.G + 1);
      pm.mix_idx.resize((size_t)pm.G);
      pm.mix_w.assign((size_t)pm.G, 1.0);
      for (int64_t i = 0; i <= pm.G; i++) pm.mix_off[(size_t)i] = (int32_t)i;
      for (int64_t i = 0; i < pm.G; i++) pm.mix_idx[(size_t)i] = (int32_t)i;
      pm.weights_normalized = true;
      pm.hmm_label.clear();
      pm.hmm_states.clear();
      auto sub = std::make_unique<aasr_gmm>();
      sub->device = g->device;
      gmm_build(sub.get(), pm);
      g->pool_view = std::move(sub);
    }
    g->pool_view->precision = g->precision;
    g->pool_view->use_bf16x3 = g->use_bf16x3;
    gmm_score_launch(g->pool_view.get(), d_frames, F, d_out, stream);
    return;
  }
  if (g->ill_conditioned || g->hyb_enabled) {
    // variance-floored Gaussians: the expanded form loses eps * kappa, so the whole pool is
    // evaluated in the centred form (one-record states, no floor)
    gmm_build_pool_centred(g);
    CentredOps ops{g->poolc_recs.p, g->poolc_state_off.p, g->poolc_splits.p, g->poolc_max_splits, g->G, 1};
    ops.n_recs = g->G;
    ops.floor_val = NEG_BIG_F;
    if (!launch_centred_ops(g, ops, g->centred_dimp, d_frames, F, d_out, stream))
      raise(AASR_ERR_UNSUPPORTED, "no centred kernel instance for dimension %d", g->dim);
    return;
  }
  gmm_build_pool(g);
  launch<1>(g, g->pool, d_frames, F, d_out, g->G, stream);
}

}  // namespace aasr

#ifdef AASR_PL_TRACE
// Diagnostic of experiment builds (tools/pl_trace.py): the phase sums of the last traced launch, [wave][interval]
extern "C" int aasr_debug_pl_trace(unsigned long long *out) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(aasr::g_pl_trace), sizeof(unsigned long long) * 96) == hipSuccess ? 0 : -1;
}
#endif
