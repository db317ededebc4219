// gmm.h -- device-resident diagonal-GMM acoustic model (aasr_gmm).
//
// Replaces the scoring side of aku::HmmSet / PDFPool / Mixture /
// DiagonalGaussian (aku/HmmSet.cc:484-501, aku/Distributions.cc:1040-1062,
// 2078-2086, 2647-2682).  See DESIGN.md "GMM scoring kernel" for the layout.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <functional>
#include <memory>
#include <vector>

#include "common.h"

namespace aasr {

// Host-side parsed model, double precision as the reference stores it.
struct HostModel {
  int dim = 0;
  int64_t G = 0;
  std::vector<double> mean, var;        // [G x dim]
  // full-covariance Gaussians (FullCovarianceGaussian, aku/Distributions.cc
  // :1466-1488): cov is [G x dim x dim] row-major, is_full[g] marks which
  // entries use it; both empty for purely diagonal pools
  std::vector<double> cov;
  std::vector<uint8_t> is_full;
  bool any_full() const { return !is_full.empty(); }
  // per-Gaussian offset added to the log-likelihood constant (natural log; empty = none).
  // Subspace-constrained Gaussians (SCGMM) are expanded at load time into full-covariance
  // Gaussians; the reference's constant for them (aku/Distributions.cc:1905-1914: log det P -
  // psi^T P^-1 psi - d log(2*3.1416), no halves) is not that of the normalised density, the
  // difference rides here.
  std::vector<double> gauss_bias;
  // model-side constrained MLLR (ConstrainedMllr / AdaptedGaussian,
  // aku/ModelModules.hh:128-212): Gaussian g scores A_t f + b_t instead of f and
  // its likelihood is multiplied by |prod diag A_t| (the reference's
  // full_matrix_determinant, aku/LinearAlgebra.cc:73-86).  xform is
  // [n][dim][dim+1] with column 0 = b (the reference's W layout).
  int n_transforms = 0;
  std::vector<int32_t> g2t;   // [G] transform index or -1
  std::vector<double> xform;
  double logw_bias = 0;       // log|det| of a global transform, folded into every weight
  double logw(size_t k) const {
    return mix_w[k] > 0 ? std::log(mix_w[k]) + logw_bias : -INFINITY;
  }
  bool global_xform() const {
    if (n_transforms != 1) return false;
    for (int32_t t : g2t)
      if (t != 0) return false;
    return true;
  }
  bool factor_path() const { return any_full() || (n_transforms > 0 && !global_xform()); }
  // HMM inventory of the .ph file (label, emission pdf of every state); only
  // models created from files have it.  Used by UNIT_PHONE regression classes.
  std::vector<std::string> hmm_label;
  std::vector<std::vector<int32_t>> hmm_states;
  int64_t S = 0;
  std::vector<int32_t> mix_off;         // [S+1]
  std::vector<int32_t> mix_idx;         // [K]
  std::vector<double> mix_w;            // [K] (as read; normalised by the first build)
  bool weights_normalized = false;      // Mixture::normalize_weights already applied
  // fingerprint of the text files the model was parsed from: (size, FNV-1a of the content) of
  // .gk, .mc, .ph (zeros for an absent .ph); carried into the binary cache so that a cache is
  // only trusted for the files it was written from
  bool has_src_fp = false;
  uint64_t src_fp[6] = {0, 0, 0, 0, 0, 0};
  // Pivot groups -- engine-internal models only (gmm_plan_engine_parts): the states are sorted into groups, the rows of
  // group p are expanded around pg_pivot[p] instead of the pool's one pivot.  Group p holds the states
  // [pg_begin[p], pg_real_end[p]); the states up to pg_begin[p + 1] -- a multiple of 32, so that every group starts on a
  // whole line of the score matrix -- are padding columns without components, never scored, never read.
  std::vector<int32_t> pg_begin, pg_real_end;   // [P + 1], [P]
  std::vector<float> pg_pivot;                  // [P][dim]
  int pg_arith = 0;                             // 2: two fp16 terms, 3: three bf16 terms, 4: two fp16 terms with the
                                                // constant dealt to the K slabs (pg_sc()) (0: no pivot groups)
  bool pg_sc() const { return pg_arith == 4; }
  int n_pg() const { return pg_arith ? (int)pg_real_end.size() : 0; }
  int pg_of_state(int64_t s) const {
    int p = 0;
    while (p + 1 < (int)pg_real_end.size() && s >= pg_begin[(size_t)p + 1]) p++;
    return p;
  }
};
void model_files_fingerprint(const char *gk, const char *mc, const char *ph, uint64_t fp[6]);

HostModel read_model_files(const char *gk, const char *mc, const char *ph);
// model_cache.cc: parsed model <-> one binary blob (magic, sizes, arrays, FNV-1a checksum)
void write_model_cache(const HostModel &m, const char *path);
HostModel read_model_cache(const char *path);
HostModel read_model_cache_checked(const char *path, const char *gk, const char *mc, const char *ph);

// Rows of the streamed operand are processed in tiles of TILE_ROWS; the
// epilogue reduces them in chunks of CHUNK_ROWS (one 32x32 MFMA block).
constexpr int TILE_ROWS = 64;
constexpr int CHUNK_ROWS = 32;
constexpr int FRAMES_PER_WAVE = 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int FRAMES_PER_BLOCK = FRAMES_PER_WAVE * WAVES_PER_BLOCK;
constexpr int TRACK_OUT_GROUP = 32;   // states written per frame row at a time (grouped)
constexpr int TRACK_MAX_SPLITS = 16;  // row-range cuts available to the launcher
constexpr int PG_MAX = 32;            // pivot groups of a multi-pivot layout
constexpr int PG_MAX_SPLITS = 48;     // ... and the cuts of its table (every group is at least one cut)
constexpr int CENTRED_MAX_SPLITS = 32;
// Conditioning limits of the expanded (matrix-core) forms.  kappa_g = sum_d p (mu - pivot)^2 -- how far a Gaussian's mean
// lies from the expansion point in units of its own standard deviations -- and kappa2, the 2-norm of the same terms.
//
// Round 6 (tools/exp_calib.py, tools/mfma_sum): what limits the expanded form is NOT the operands' precision
// (three bf16 terms are no better than two fp16 terms at the same kappa) but the matrix instruction's summation: every
// 8-product step aligns its terms and the accumulator to the largest exponent and cuts them 24 bits below it, and in
// the plain K layout the accumulator starts at -1/2 log2e kappa, so every instruction cuts at that magnitude: the
// error grows linearly with kappa (2e-7 kappa + 3e-5, per-state maxima over 3 072 frames on and up to 1.5 sigma off
// the data, 240 fitted models).  Sample maxima by (kappa, kappa2): plain two-term layout 4.8e-5 below (250, 80)
// [7.6e-5 in 13 dimensions], 8.5e-5 at (330, 160), 1.0e-4 at (500, 200), 1.2e-4 at 600+ -- the round-5 limits
// (330 / 80 two terms, 600 / 200 three terms) had been set on frames near the model and let 1.0-1.45e-4 through on
// frames 7-14 sigma out; slab-constant layout (TrackLayout::sc) 7.4e-5 below (450, 120), 8.2e-5 at (600, 200), 1.0e-4 at
// (1 100, 260).  Limits are set where the sample maxima stay at or below 7.5e-5; beyond them: the centred form.
#ifndef AASR_KAPPA_LIMIT_VALUE
#define AASR_KAPPA_LIMIT_VALUE 300.0
#endif
#ifndef AASR_KAPPA2_LIMIT_VALUE
#define AASR_KAPPA2_LIMIT_VALUE 130.0
#endif
constexpr double KAPPA_LIMIT = AASR_KAPPA_LIMIT_VALUE;
// the same estimate taken as a 2-norm over the dimensions: rounding errors of different dimensions add
// in quadrature, those of one dimension do not, so a model whose kappa sits in one or two dimensions
// (low-dimensional models above all) reaches the tolerance at a much smaller sum.  Fuzz seed 104
// iteration 247: D = 1, kappa 416, a frame 12 sigma out (ll = -71.2), 1.18e-4 in the expanded form.
constexpr double KAPPA2_LIMIT = AASR_KAPPA2_LIMIT_VALUE;
// The two-term fp16 split (AASR_PREC_F16X2) carries 22 bits per operand instead of 24: its error in the
// expanded form is 1.4-1.8x that of the bf16x3 / f32 forms at the same conditioning (tools/exp_fp16_split.py:
// 3.4e-5 against 1.9e-5 on 10^7 states at kappa 165; 1.24e-4 against 8.6e-5 at kappa 885), so a layout is
// packed for it only when every Gaussian on the matrix path stays below limits tighter by that factor.
// The 2-norm limit is the binding one.  Over many dimensions the terms' rounding errors add in quadrature and the worst
// of 3.1e9 values of the configs[1] model (kappa2 <= 74.5) is 4.0e-5; a model of one or two dimensions has no such
// averaging: a sweep's one-dimensional model (tools/fuzz_parity.py 3002, iteration 290) came out 8.85e-5 off with every
// Gaussian below kappa2 = 80 -- 1.1e-6 per unit against 0.5e-6 for the 39-dimensional model -- so models of fewer than
// 8 dimensions get the limit that keeps that case at the 5e-5 the other forms show.
#ifndef AASR_KAPPA_LIMIT_F16_VALUE
#define AASR_KAPPA_LIMIT_F16_VALUE 250.0
#endif
#ifndef AASR_KAPPA2_LIMIT_F16_VALUE
#define AASR_KAPPA2_LIMIT_F16_VALUE 80.0
#endif
constexpr double KAPPA_LIMIT_F16 = AASR_KAPPA_LIMIT_F16_VALUE;
constexpr double KAPPA2_LIMIT_F16 = AASR_KAPPA2_LIMIT_F16_VALUE;
// The slab-constant layout (TrackLayout::sc) of the engine parts: limits from tools/exp_calib.py (round 6)
#ifndef AASR_KAPPA_LIMIT_SC_VALUE
#define AASR_KAPPA_LIMIT_SC_VALUE 500.0
#endif
#ifndef AASR_KAPPA2_LIMIT_SC_VALUE
#define AASR_KAPPA2_LIMIT_SC_VALUE 160.0
#endif
constexpr double KAPPA_LIMIT_SC = AASR_KAPPA_LIMIT_SC_VALUE;
constexpr double KAPPA2_LIMIT_SC = AASR_KAPPA2_LIMIT_SC_VALUE;
#ifdef AASR_F16_LOWDIM80   // experiment build
constexpr double KAPPA2_LIMIT_F16_LOWDIM = 80.0;
#else
constexpr double KAPPA2_LIMIT_F16_LOWDIM = 45.0;   // dim < 8
#endif
// |x - pivot| beyond this is clamped in the f16x2 kernel's frame operand (the square must stay below 65504)
constexpr float kF16Clamp = 240.0f;

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

// Two-track row layout for the in-register epilogue (see gmm_build_tracks()).
struct TrackLayout {
  bool ok = false;
  bool grouped = false;
  int64_t states_f16 = 0;    // states the two-term fp16 rows cover (0: no a16h)
  PackedRows rows;
  // the same rows split into three bf16 terms for k_gmm_diag_score_bf16x3:
  // [tile][K/16 slabs][3 splits][2 row blocks][64 lanes][8 bf16]
  DevBuf<uint16_t> a16;
  // ... and into two fp16 terms for the f16x2 form of the same kernel (AASR_PREC_F16X2), only when the rows
  // are eligible (conditioning below KAPPA_LIMIT_F16, values and clamp inside the fp16 range):
  // [tile][K/16 slabs][2 splits][2 row blocks][64 lanes][8 fp16]; the constant rides in K slots dim and KH + dim
  DevBuf<uint16_t> a16h;
  DevBuf<float> f16tab;      // f16x2: [2 KH] per-column scales of the frame operand (2^s_k), [KH] clamp of |x - pivot|
  int nk16 = 0;              // K/16 (K = 2*KH, KH = 8*nk16 >= dim+1)
  // Slab-constant K layout (two fp16 terms, engine parts only): every slab of 16 K slots carries ITS seven dimensions'
  // share of the constant in its first two slots (value + remainder) -- -1/2 log2e sum p mu'^2 over the slab's dimensions;
  // the last slab in use also the rest, peak + log w + reference -- then (linear, quadratic) of dimensions 7 j .. 7 j + 6.
  // The accumulators then never hold more than the value itself: in the plain layout they start at -1/2 log2e kappa and
  // every matrix instruction cuts its smaller terms at the accumulator's exponent (tools/mfma_sum), an error that grows
  // with kappa whatever the operands' precision (tools/exp_calib.py).  Six slabs instead of five for 39 dimensions.
  bool sc = false;
  DevBuf<uint16_t> close;    // per tile: bit p (+8 for track 1) = a state closes after quad p
  DevBuf<int32_t> sid;       // [2][sid_stride] state index of the k-th close on each track
  int32_t sid_stride = 0;
  DevBuf<int32_t> splits;    // [MAX_SPLITS][MAX_SPLITS+1][4]: tile, closes track 0, closes track 1, pivot group of the cut
  int max_splits = 1;
  // multi-pivot layouts (HostModel::pg_*; grouped only): the groups are runs of whole tiles, a row cut lies inside one
  // group; the frame operand has one image per group (k_frame_operand), the cut table says which a workgroup takes
  int n_pg = 0;              // 0 / 1: the model's one pivot
  int split_cap = TRACK_MAX_SPLITS;   // rows of the cut table (PG_MAX_SPLITS for multi-pivot layouts)
  DevBuf<float> pg_pivot;    // [n_pg][dim]
  DevBuf<float> pg_tab;      // f16x2: [n_pg][3 KH] the groups' own column scales and clamps (f16tab's layout)
  DevBuf<int32_t> pg_colend; // [n_pg] one past the group's last output column
  int64_t rows_padded = 0;
  float ref_ln = 0.0f;       // reference exponent * ln 2
  double ref_log2 = 0;       // the reference exponent itself
  std::vector<int32_t> row_gauss;   // host: pool Gaussian of every packed row, -1 = null row
};

// Row layout of the full-covariance kernel: every mixture component occupies
// ceil(dim/4) consecutive quads of one track with the rows of
// sqrt(log2e/2) * R^-1 (Sigma = R R^T), so that the accumulator holds
// y = R^-1 (x - mu) and the quadratic form is a sum of squares.
// Two-term fp16 form of the factor rows: worst visible state error ~3.7e-5 at kappa = 2 400 and 9.8e-5 at 24 000
// (host emulation on 13-dimensional pools, tools/exp_fullcov_f16.py), the three-term form ~2x below it.  The error of a
// row goes with sqrt(kappa) (the two terms of y = R^-1 x' - R^-1 mu' that cancel are that large) times y itself, and a
// low-dimensional pool has no other rows to average it with: at 1 500 the sweeps' worst was 6.8e-5, at 600 still 6.2e-5
// (a 63-dimensional pool, one of 2 000), hence 300.
#define FULL_KAPPA_LIMIT_F16 300.0
#define FULL_KAPPA_LIMIT_F16_LOWDIM 40.0    // fewer than 8 dimensions (6.8e-5 on a 2-dimensional pool at kappa < 600, 5.7e-5 below 150)
constexpr float kFullF16Clamp = 30000.0f;   // the SCALED |x - pivot| beyond this is clamped in the f16x2 factor-row kernel

struct FullLayout {
  bool ok = false;
  PackedRows rows;
  DevBuf<uint32_t> close;   // per tile: bits 0-7/8-15 Gaussian closes (track 0/1),
                            //           bits 16-23/24-31 state closes
  DevBuf<float> gconst;     // [2][g_stride] (c_g + log w)*log2e + ref per Gaussian close
  DevBuf<int32_t> sid;      // [2][s_stride] state index per state close
  int32_t g_stride = 0, s_stride = 0;
  DevBuf<int32_t> splits;   // [MAX_SPLITS][MAX_SPLITS+1][8]: tile, ks0, ks1, kg0, kg1
  int max_splits = 1;
  float ref_ln = 0.0f;
  // the same rows as three bf16 terms for k_gmm_full_score_bf16x3:
  // [tile][K/16 slabs][3 splits][2 row blocks][64 lanes][8 bf16], K index = column
  DevBuf<uint16_t> a16;
  int nk16 = 0;
  // ... and as two fp16 terms (AASR_PREC_F16X2; [tile][slab][2 splits][2 row blocks][64 lanes][8 fp16]): packed only
  // where the pool's conditioning estimate kappa = max_g |R^-1 (mu - pivot)|^2 (in the rows' log2 scaling) is below
  // FULL_KAPPA_LIMIT_F16 and every coefficient is inside the fp16 range
  DevBuf<uint16_t> a16h;
  DevBuf<float> f16scale;   // f16x2: [dim] power-of-two scale of every frame-operand column (the rows carry the inverse)
  double kappa = 0.0;
  std::vector<int32_t> row_gauss;   // host: pool Gaussian of every packed row, -1 = unused row (Gaussian clustering)
  int64_t rows_padded = 0;
  // per tile and track, by quad position: constant of the component / index of the state that
  // closes there ([tiles][2][8]) -- the bf16x3 kernel fetches them with the tile instead of
  // chasing gconst / sid through dependent loads in its epilogue
  DevBuf<float> gc_tile;
  DevBuf<int32_t> sid_tile;
};

// Gaussian clustering (PDFPool::read_clustering + the cluster branch of
// PDFPool::precompute_likelihoods, aku/Distributions.cc:3114-3170, 2684-2722).
struct ClusterState {
  bool loaded = false;   // a clustering has been read
  bool enabled = false;  // set_clustering_min_evals() switches it on (HmmSet.cc:1359-1366)
  int32_t C = 0, Cs = 0;           // clusters, padded to a multiple of 8
  int min_clusters = 0, min_gaussians = 0;
  std::vector<int32_t> g2c;        // [G] cluster of each Gaussian, -1 = none
  std::vector<int32_t> csize_h;    // [C] members, duplicates counted as the reference does
  std::vector<double> c_mean, c_prec, c_cst;   // centres (host copy for the adapters/tests)
  int dimp = 0;                    // dimension padded to a multiple of 8
  DevBuf<double> rec;              // [Cs/8][dimp][8][2] (mean, precision) of the centres
  DevBuf<double> cconst;           // [Cs] constants
  DevBuf<double> rec_fma, cconst_fma;  // the same centres in the expanded form (p mu, -p/2), c - 1/2 sum p mu^2
  DevBuf<double> bpack;            // ... packed as f64 MFMA operands: [tile of 16][k step][64 lanes]
  int mfma_ks = 0;                 // k steps of 4 covering 2 dim + 1
  DevBuf<int32_t> csize;           // [Cs] members per cluster (0 beyond C)
  DevBuf<int32_t> crow[2];         // cluster of each packed row of the grouped / independent track layout
                                   // (C = no cluster or null row)
  DevBuf<int32_t> crow_full;       // ... of the factor rows of a full-covariance pool
  DevBuf<int32_t> crow_hyb, crow_centred;  // the same for the records of the centred kernel: the
                                   // outlier components (outlier routing) / every component
  // per-state centre weights W[s][c] = sum of the weights of s's components in
  // cluster c, ELL layout [nnz][S]
  int nnz = 0;
  DevBuf<int32_t> w_cluster;
  DevBuf<float> w_weight;
  double ref_log2 = 0;             // reference exponent shared with the track kernels
  bool log_merge = false;          // centre values as log2 and k_cluster_merge_log (no common f32 exponent)
  // per-call scratch: pass-wide buffers sized for Fc frames, ll64 for a sub-pass of Fs
  int64_t Fc = 0, Fs = 0;
  DevBuf<double> ll64;             // [Fs][Cs] centre log-likelihoods (ranking keys); on the float-key path: the rows of the frames left to the replay
  DevBuf<float> key32;             // [Fs][Cs] the keys as floats (k_cluster_select<KPL, float>)
  DevBuf<int32_t> crow_gauss;      // [G] cluster of every pool Gaussian, C = none (models scored as dimension parts)
  DevBuf<int32_t> pend_list;       // [Fs + 1] frames the float selection left open (k_cluster_select_pending); last slot = count
  DevBuf<unsigned long long> maskw;  // [Fc/64][C+1] bit f = frame f takes the exact values
  DevBuf<unsigned long long> maskrow;  // [Fc/64][rows_padded] the same per packed row
  DevBuf<float> cval;              // [Fc][C] 2^(log2e*ll_c + ref) of the centres that stand in
  DevBuf<int32_t> n_exact;         // [Fc] clusters evaluated exactly (diagnostic)
  // frames whose selection threshold falls inside a group of equal centre likelihoods: the
  // reference's priority queue decides which of them are popped, k_cluster_select_heap replays it
  DevBuf<int32_t> tie_list;        // [Fs] sub-pass frame indices, tie_list[Fs] = count
  DevBuf<double> heap_key;         // [C][kHeapThreads]
  DevBuf<int32_t> heap_idx;        // [C][kHeapThreads]
  int64_t tie_frames_total = 0;    // frames that took the replay since the clustering was set (diagnostic, updated lazily)
};
constexpr int kHeapThreads = 4096;

}  // namespace aasr

struct aasr_gmm {
  int device = 0;
  int dim = 0;
  int64_t G = 0, S = 0;
  int precision = AASR_PREC_F16X2;  // default: the kernel bench.py reports -- two fp16 terms where the model's conditioning
                                    // allows it, three bf16 terms otherwise (f32 rows where no split layout exists)
  aasr::HostModel host;             // kept for lazy f64 build / adapters
  std::vector<float> pivot;         // per-dimension centring pivot
  aasr::DevBuf<float> d_pivot;
  aasr::PackedRows mix;             // component-expanded, per-state reduce
  aasr::PackedRows pool;            // pool Gaussians, raw log-likelihoods
  bool pool_built = false;
  // track layouts for the in-register epilogue (built when eligible)
  aasr::TrackLayout paired;   // grouped: states 2j/2j+1 side by side
  aasr::TrackLayout tracks;   // independent tracks (built when `paired` is not)
  int64_t f16_bad_state = -1; // builder scratch: the state whose rows failed the fp16 range / clamp conditions
  // Engine parts (gmm_plan_engine_parts; round 6: the one routing mechanism -- round 4's mixed layout and its spare-column
  // sub-model are gone): the model as up to
  // three internal models over disjoint sets of its states -- [0] the states that qualify for two fp16 terms around the
  // pivot of their GROUP (a multi-pivot model: pivot groups, HostModel::pg_*), [1] the same for three bf16 terms, [2]
  // whatever is left, as an ordinary model (one pivot; outlier routing, centred form ...) -- each scored into its own
  // column range of the engine's score rows; engine_colmap says where a state's column is.
  struct EnginePart {
    std::unique_ptr<aasr_gmm> model;
    int64_t col0 = 0;        // first column of the part in an engine score row
    int64_t cols = 0;        // columns it occupies (a multiple of 32)
    int arith = 0;           // 2 / 3 / 4: pivot-group model in that arithmetic (HostModel::pg_arith), 0: ordinary model
    int64_t states = 0;      // real states
  };
  std::vector<EnginePart> engine_parts;
  std::string engine_plan_note;          // what the planner did and why (aasr_debug_engine_plan_note)
  aasr::DevBuf<int32_t> engine_colmap;   // [S]
  std::vector<int32_t> engine_colmap_h;
  int64_t engine_cols = 0;
  bool is_engine_part = false;
  mutable aasr::DevBuf<float> engine_scratch, engine_part_scratch;   // public-layout callers: engine rows of a chunk of frames
  std::vector<uint8_t> f16_state_ok;   // per state: eligible for the two-term fp16 form (conditioning limits, probe)
  int64_t f16_probe_moved = 0;         // states the load-time probe (gmm_probe_f16x2) took out of the fp16 form
  bool f16_whole_rejected = false;     // the probe took the whole-model two-term rows away: no layout packs them again
  // full-covariance path (k_gmm_full_score): rows are the rows of R^-1 of
  // every component, see gmm_build_fullcov()
  aasr::FullLayout full;
  int num_cus = 0;
  int layout_mask = 7;        // see aasr_debug_set_layouts()
  bool use_bf16x3 = true;     // score with the split-operand kernels (AASR_PREC_BF16X3 / AASR_PREC_F16X2)
  double kappa_matrix = 0, kappa2_matrix = 0;  // conditioning estimates over the Gaussians that stay on the matrix path
  // centred-form (numerically safe) kernel operands
  bool centred_ok = false, ill_conditioned = false;
  double kappa = 0;           // conditioning estimate of the expanded form
  int centred_dimp = 0;
  aasr::DevBuf<float> centred_recs;        // [rows][3*dimp+4]
  aasr::DevBuf<int32_t> centred_state_off; // [S+1]
  aasr::DevBuf<int32_t> centred_splits;    // [MAX][MAX+1] state boundaries
  int centred_max_splits = 1;
  // AASR_PREC_F64: double records [mean][precision][constant, weight] per mixture component
  bool f64_built = false;
  int f64_dimp = 0;
  aasr::DevBuf<double> f64_recs, f64_x, f64_out, f64_A, f64_b, f64_xframes;
  double f64_det = 1.0;          // |prod diag A| of a global transform
  aasr::DevBuf<int32_t> f64_state_off;
  // per-class model transforms under AASR_PREC_F64: class of every record (0 = unadapted), the classes' [A | b] and
  // |prod diag A|, the frames of every class as [class][dim][frame]
  int f64_classes = 0;
  aasr::DevBuf<int32_t> f64_rec_class;
  aasr::DevBuf<double> f64_class_A, f64_class_b, f64_class_det, f64_class_x;
  // the pool's Gaussians as single-record states (per-Gaussian view of ill-conditioned models)
  bool pool_centred_built = false;
  aasr::DevBuf<float> poolc_recs;
  aasr::DevBuf<int32_t> poolc_state_off, poolc_splits;
  int poolc_max_splits = 1;
  // Outlier routing: when only a minority of the Gaussians break the conditioning limit, those
  // (outlier[g] != 0) are taken out of the matrix layouts (null rows) and scored in the centred
  // form over the states that hold them; k_outlier_merge adds the two parts per state.
  std::vector<uint8_t> outlier;            // per pool Gaussian
  bool hyb_enabled = false;
  int64_t hyb_states = 0, hyb_rows = 0;    // states with outliers, outlier components
  aasr::DevBuf<float> hyb_recs;            // centred records of the outlier components
  aasr::DevBuf<int32_t> hyb_state_off;     // [hyb_states + 1]
  aasr::DevBuf<int32_t> hyb_splits;        // [MAX][MAX+1]
  int hyb_max_splits = 1;
  aasr::DevBuf<int32_t> hyb_map;           // [hyb_states] -> state index
  std::vector<int32_t> hyb_comps;          // mixture-component index of every outlier record (host)
  std::vector<int32_t> parent_gauss;       // a class sub-model: parent pool index of each of its Gaussians
  aasr::DevBuf<float> hyb_scratch;         // [frames of a pass][hyb_states]
  // the merge fused into the scoring kernel's close logic (k_gmm_diag_score_pl<..., HYB>): per state the next state of its
  // track parity with outlier components | that state's record << 16 ([S]; models of up to 65 534 states and records);
  // hyb_fuse: set by the scoring launcher around the launch that is to take the partial sums
  aasr::DevBuf<uint32_t> hyb_tab;
  mutable struct HybFuse { const float *part = nullptr; int64_t pitch = 0; } hyb_fuse;
  // Global constrained MLLR without re-packing: the rows stay those of the unadapted model
  // (rows_unbiased) and log|det| is added to every score at the kernels' output (out_bias_ln)
  bool rows_unbiased = false;
  double out_bias_ln = 0;
  // Class routing (per-class constrained MLLR on diagonal pools): one sub-model per regression
  // class (its mixture components only, built like any diagonal model), scored on the frames
  // transformed by that class's [b | A]; k_class_merge adds log|det| and sums the classes per
  // state.  The sub-models depend on the class membership only, so a speaker change uploads a
  // handful of matrices instead of re-packing the model.
  bool class_routing = false;
  std::vector<int32_t> class_g2t;                         // membership the sub-models were built for
  // Feature dimension > 63 (the matrix kernels keep a frame's K operand in registers: K = 2 dim + 1 <= 128): the
  // diagonal density factorises over the dimensions, so the model is cut into parts of <= 63 dimensions, each a pool
  // of one-component "states" scored per Gaussian by the kernels above, and k_dim_split_combine adds a component's
  // parts and forms the mixture sums.  A slower path (per-Gaussian scores cross HBM) for models the reference allows.
  std::vector<std::unique_ptr<aasr_gmm>> dim_parts;
  std::vector<int32_t> dim_part_off;          // first dimension of every part, + dim
  aasr::DevBuf<float> dim_part_x, dim_part_ll;  // a part's frame columns; [parts][chunk frames][G] scores
  aasr::DevBuf<int32_t> dim_mix_off, dim_mix_idx;
  aasr::DevBuf<float> dim_mix_logw;
  std::unique_ptr<aasr_gmm> pool_view;   // full-covariance pools: the Gaussians as one-component states (per-Gaussian view)
  std::vector<std::unique_ptr<aasr_gmm>> class_models;    // index = transform id + 1 (0: unadapted), may be null
  std::vector<aasr::DevBuf<double>> class_a, class_b;     // per class: A [dim x dim], b [dim]
  std::vector<double> class_logdet;                       // log |prod diag A|; -inf = class contributes nothing
  aasr::DevBuf<float> class_scratch, class_xframes;
  // device buffers of aasr_run_utterance, kept between calls (pipeline.cc: BlockRunner)
  std::shared_ptr<void> utt_scratch;
  std::shared_ptr<void> recipe_scratch;  // the recipe driver's device buffers, streams and pinned result slots
  // global CMLLR transform applied to the frames before scoring
  aasr::DevBuf<double> xf_a, xf_b;
  aasr::DevBuf<float> d_xframes;
  aasr::ClusterState cl;
  // staging for the host-buffer entry points
  aasr::DevBuf<float> d_frames, d_out;
  // frame operand of the split-term kernels (k_frame_operand): the K x 64 operand of every block of 64 frames, formed
  // once per launch; per-launch scratch, hence mutable
  mutable aasr::DevBuf<uint32_t> fop_scratch;
};

namespace aasr {
void gmm_build(aasr_gmm *g, const HostModel &m);
// model-side constrained MLLR (n_transforms = 0 resets): updates in place where it can (a global
// transform over the unadapted rows; per-class transforms with an unchanged membership), rebuilds otherwise
void gmm_set_transforms(aasr_gmm *g, int32_t n_transforms, const int32_t *gauss_to_transform, const double *W);
void gmm_build_pool(aasr_gmm *g);
// maskw / c1 / gclus (Gaussian clustering): a component counts for a frame only where its cluster's selection bit is set;
// no floor then (the merge adds the centres)
void gmm_dim_split_score(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, bool per_gaussian, hipStream_t stream,
                         const unsigned long long *maskw = nullptr, int c1 = 0, const int32_t *gclus = nullptr,
                         bool frames_adapted = false);
void gmm_build_pool_centred(aasr_gmm *g);
void gmm_build_f64(aasr_gmm *g);
void gmm_cluster_score_f64_launch(aasr_gmm *g, const double *d_frames, const double *d_members, int64_t F,
                                  double *d_out, int linear, double det, hipStream_t stream);
void gmm_f64_masked_launch(aasr_gmm *g, const double *d_members, int64_t F, double *d_out, int linear, double det,
                           const int32_t *crow, const unsigned long long *maskw, int c1, const double *ll64,
                           int64_t Cs, int C, hipStream_t stream);
void gmm_f64_classes_masked_launch(aasr_gmm *g, const double *d_frames, int64_t n, double *d_out, int linear,
                                   const int32_t *crow, const unsigned long long *maskw, int c1, const double *ll64,
                                   int64_t Cs, int C, hipStream_t stream);
void gmm_score_f64_launch(aasr_gmm *g, const double *d_frames, int64_t F, double *d_out, int linear,
                          hipStream_t stream);
void gmm_build_tracks(aasr_gmm *g, bool grouped);
void gmm_probe_f16x2(aasr_gmm *g);
void gmm_plan_engine_parts(aasr_gmm *g);
bool gmm_engine_parts_active(const aasr_gmm *g);
bool gmm_engine_parts_clustered(const aasr_gmm *g);   // ... under Gaussian clustering (gmm_cluster_score_launch)
void gmm_scatter_columns(const float *dense, int64_t F, int64_t n, float *out, int64_t pitch, hipStream_t stream);
void gmm_add_bias_nofloor(float *d_out, int64_t n, float bias, hipStream_t stream);
void gmm_gather_engine_columns(const aasr_gmm *g, const float *rows, int64_t F, int64_t in_pitch, float *out, int64_t out_pitch,
                               hipStream_t stream);
// the engine's own score layout: rows of gmm_engine_pitch() floats; state s in column gmm_engine_colmap()[s] (nullptr: s)
int64_t gmm_engine_pitch(const aasr_gmm *g);
int64_t gmm_engine_pitch_max(const aasr_gmm *g);
const int32_t *gmm_engine_colmap(const aasr_gmm *g);
void gmm_score_launch_engine(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, int64_t pitch, hipStream_t stream);
void lna_encode_launch(const float *d_loglik, int64_t F, int S, int normalize, int lnabytes, float *d_lp,
                       uint8_t *d_bytes, hipStream_t stream, int64_t in_pitch, const int32_t *d_colmap = nullptr);
void gmm_build_centred(aasr_gmm *g);
void gmm_build_fullcov(aasr_gmm *g);
void gmm_full_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                     hipStream_t stream);
bool gmm_score_pitch_ok(const aasr_gmm *g);
void gmm_score_launch_pitched(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, int64_t pitch,
                              hipStream_t stream);
void gmm_score_launch(aasr_gmm *g, const float *d_frames, int64_t F,
                      float *d_out, hipStream_t stream);
void gmm_gauss_launch(aasr_gmm *g, const float *d_frames, int64_t F,
                      float *d_out, hipStream_t stream);
// gmm_cluster.hip
void gmm_set_clustering(aasr_gmm *g, int32_t n_clusters, int64_t n_pairs,
                        const int32_t *gauss_index, const int32_t *cluster_index);
void gmm_read_clustering(aasr_gmm *g, const char *path);
void gmm_set_clustering_min_evals(aasr_gmm *g, double min_clusters, double min_gaussians);
const float *gmm_adapted_frames(aasr_gmm *g, const float *d_frames, int64_t F, hipStream_t stream);
// Clustered pass over per-class transforms: for every class with components, `exact` writes the
// class sub-model's exact part on the class's adapted frames into a scratch; the parts are added
// into out with the class's log|det| (no floors).
void gmm_classes_exact_launch(aasr_gmm *g, const float *d_frames, int64_t n, float *d_out,
                              const std::function<void(aasr_gmm *, size_t, const float *, float *)> &exact,
                              hipStream_t stream);
void gmm_outliers_masked_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, const int32_t *crow,
                                const unsigned long long *maskw, int c1, int64_t n_words, hipStream_t stream);
void gmm_centred_masked_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out, const int32_t *crow,
                               const unsigned long long *maskw, int c1, int64_t n_words, hipStream_t stream);
void gmm_cluster_score_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                              hipStream_t stream, int64_t pitch = 0);
bool gmm_cluster_pitch_ok(const aasr_gmm *g);
void gmm_tracks_masked_launch(aasr_gmm *g, int which, const float *d_frames, int64_t F,
                              float *d_out, const unsigned long long *maskrow,
                              hipStream_t stream, int64_t pitch = 0);
void gmm_full_masked_launch(aasr_gmm *g, const float *d_frames, int64_t F, float *d_out,
                            const unsigned long long *maskrow, hipStream_t stream);
}  // namespace aasr
