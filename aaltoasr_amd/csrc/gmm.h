// gmm.h -- device-resident diagonal-GMM acoustic model (aasr_gmm).
//
// Replaces the scoring side of aku::HmmSet / PDFPool / Mixture /
// DiagonalGaussian (aku/HmmSet.cc:484-501, aku/Distributions.cc:1040-1062,
// 2078-2086, 2647-2682).  See DESIGN.md "GMM scoring kernel" for the layout.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "common.h"

namespace aasr {

// Host-side parsed model, double precision as the reference stores it.
struct HostModel {
  int dim = 0;
  int64_t G = 0;
  std::vector<double> mean, var;        // [G x dim]
  int64_t S = 0;
  std::vector<int32_t> mix_off;         // [S+1]
  std::vector<int32_t> mix_idx;         // [K]
  std::vector<double> mix_w;            // [K] (as read; normalised in build)
};

HostModel read_model_files(const char *gk, const char *mc, const char *ph);

// Rows of the streamed operand are processed in tiles of TILE_ROWS; the
// epilogue reduces them in chunks of CHUNK_ROWS (one 32x32 MFMA block).
constexpr int TILE_ROWS = 64;
constexpr int CHUNK_ROWS = 32;
constexpr int FRAMES_PER_WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int FRAMES_PER_BLOCK = FRAMES_PER_WAVE * WAVES_PER_BLOCK;
constexpr int PAIRED_OUT_GROUP = 32;   // states written per frame row at a time
constexpr int PAIRED_MAX_SPLITS = 16;  // row-range cuts available to the launcher

// Kernel instances exist for these K/2 values; a model uses the smallest one
// that holds dim+1 (zero-padded beyond).
inline int pick_nkk(int dim) {
  static const int sizes[] = {8, 14, 20, 26, 32, 40, 48, 64};
  for (int s : sizes)
    if (dim + 1 <= s) return s;
  return -1;
}

// One packed operand: row-expanded Gaussians (one row per mixture component,
// or one row per pool Gaussian for the raw pool view).
struct PackedRows {
  int nkk = 0;             // K/2: dim+1 rounded up to even (one kk = one MFMA)
  int64_t rows = 0;        // real rows
  int64_t tiles = 0;       // ceil(rows / TILE_ROWS)
  DevBuf<float> a;         // [tiles][nkk/2][64 lanes][4]
  // segmented-reduce metadata (mixture view only)
  DevBuf<int32_t> chunk_seg_begin;  // [2*tiles + 1]
  DevBuf<uint32_t> seg_desc;        // row_begin | row_end<<8 | cont<<16 | open<<17
  DevBuf<int32_t> seg_out;          // output column (state index)
  // f64 mirror for AASR_PREC_F64 (built lazily)
  DevBuf<double> a64;
};

}  // namespace aasr

struct aasr_gmm {
  int device = 0;
  int dim = 0;
  int64_t G = 0, S = 0;
  int precision = AASR_PREC_F32;
  aasr::HostModel host;             // kept for lazy f64 build / adapters
  std::vector<float> pivot;         // per-dimension centring pivot
  aasr::DevBuf<float> d_pivot;
  aasr::PackedRows mix;             // component-expanded, per-state reduce
  aasr::PackedRows pool;            // pool Gaussians, raw log-likelihoods
  bool pool_built = false;
  // paired-track layout for the in-register epilogue (built when eligible)
  aasr::PackedRows paired;
  aasr::DevBuf<uint8_t> paired_close;  // per tile: bit p = a state pair closes after quad p
  int64_t paired_rows_padded = 0;
  float paired_ref_ln = 0.0f;          // reference exponent * ln 2
  aasr::DevBuf<int32_t> paired_splits; // [MAX_SPLITS][MAX_SPLITS+1][2]: tile, pairs closed
  int paired_max_splits = 1;
  int num_cus = 0;
  bool paired_ok = false;
  // staging for the host-buffer entry points
  aasr::DevBuf<float> d_frames, d_out;
};

namespace aasr {
void gmm_build(aasr_gmm *g, const HostModel &m);
void gmm_build_pool(aasr_gmm *g);
void gmm_score_launch(aasr_gmm *g, const float *d_frames, int64_t F,
                      float *d_out, hipStream_t stream);
void gmm_gauss_launch(aasr_gmm *g, const float *d_frames, int64_t F,
                      float *d_out, hipStream_t stream);
}  // namespace aasr
