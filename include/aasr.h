/*
 * aasr.h -- C ABI of the MI355X-native acoustic-likelihood engine.
 *
 * Drop-in boundary for AaltoASR's frame-parallel hot path
 *     16 kHz PCM -> MFCC chain -> diagonal-GMM state likelihoods -> LNA
 * i.e. what aku/phone_probs.cc and aku/PhoneProbsToolbox.cc drive through
 * aku::FeatureGenerator and aku::HmmSet.  Plain pointers and sizes only; no
 * C++/torch types.  Each entry point cites the reference interface it
 * replaces (paths relative to the AaltoASR tree).
 *
 * Conventions
 *  - Every function returns AASR_OK (0) or a negative aasr_status; the message
 *    is available from aasr_last_error() (thread-local).  Nothing throws
 *    across this boundary.  The C++ adapters in aaltoasr_amd/csrc/aku/ rethrow
 *    as std::string / HmmSet::*Error exactly where the reference throws.
 *  - Handles own their device memory.  Caller owns every host buffer.
 *  - "_dev" variants take device pointers (hipMalloc'ed / torch tensors) and a
 *    hipStream_t passed as void*; they enqueue work and do not synchronise.
 *    Host variants copy in, run, copy out, and synchronise.
 *  - A handle is bound to the HIP device current at creation; handles are not
 *    thread-safe, distinct handles are independent.
 *  - There is NO CPU fallback: without a usable HIP device the compute entry
 *    points return AASR_ERR_NO_DEVICE.
 */
#ifndef AASR_H
#define AASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int aasr_status;
enum {
  AASR_OK = 0,
  AASR_ERR_INVALID = -1,     /* bad argument / malformed config or model      */
  AASR_ERR_UNSUPPORTED = -2, /* valid for the reference, not built here (loud) */
  AASR_ERR_NO_DEVICE = -3,   /* no HIP device / HIP runtime error              */
  AASR_ERR_IO = -4,          /* file open/read/write failure                   */
  AASR_ERR_SHORT_AUDIO = -5  /* "audio shorter than frame" FeatureModules.cc:409 */
};

typedef struct aasr_feat aasr_feat;   /* compiled feature graph (.cfg)  */
typedef struct aasr_gmm aasr_gmm;     /* acoustic model resident in HBM */

const char *aasr_last_error(void);
const char *aasr_version(void);
/* number of visible HIP devices (0 when none / no driver) */
int aasr_device_count(void);
aasr_status aasr_set_device(int ordinal);

/* ------------------------------------------------------------------------ */
/* Feature chain: replaces aku::FeatureGenerator + FeatureModule::generate  */
/* ------------------------------------------------------------------------ */

/* FeatureGenerator::load_configuration (aku/FeatureGenerator.cc:96-219):
 * cfg_text is the text of a feature configuration file ("module { ... }"
 * blocks, aku/ModuleConfig.cc:166-202).  Supported module types: audiofile,
 * fft, mel, power, dct, delta, normalization, lin_transform, merge,
 * mean_subtractor; any other reference type -> AASR_ERR_UNSUPPORTED, unknown
 * type -> AASR_ERR_INVALID ("Unknown module type"). */
aasr_status aasr_feat_create(const char *cfg_text, aasr_feat **out);
void aasr_feat_destroy(aasr_feat *h);

/* FeatureGenerator::dim / frame_rate / sample_rate
 * (aku/FeatureGenerator.cc:281-300) */
int aasr_feat_dim(const aasr_feat *h);
float aasr_feat_frame_rate(const aasr_feat *h);
int aasr_feat_sample_rate(const aasr_feat *h);
/* dim of a named module (FeatureGenerator::module(name)->dim()), -1 unknown */
int aasr_feat_module_dim(const aasr_feat *h, const char *module_name);
/* base-module frames of look-around one output frame needs (left, right):
 * the sum of DeltaModule / MeanSubtractorModule offsets along the graph
 * (aku/FeatureModules.cc:1014-1015, 1390-1400), plus 7 frames on the left per
 * mean subtractor: its kernel anchors the window sum on frame numbers that are
 * multiples of 8 and slides it from there, so it reads up to 7 rows before the window */
void aasr_feat_halo(const aasr_feat *h, int *left, int *right);

/* AudioFileModule::last_frame (aku/FeatureModules.cc:305-308) for a file of
 * n_samples; the whole-file frame count phone_probs emits is last_frame+1
 * (aku/phone_probs.cc:217-221). */
int aasr_feat_last_frame(const aasr_feat *h, int64_t n_samples);
/* The frame at which a sequential reader of n_samples meets the end: the first frame whose window
 * crosses it (AudioFileModule::generate, aku/FeatureModules.cc:399-413 sets m_eof_frame there).
 * phone_probs emits frames 0 .. eof_frame - 1 and the border copy repeats frame eof_frame - 1.
 * Equal to last_frame() + 1 except where last_frame()'s float formula is off by one (inputs
 * beyond 2^24 samples, fractional window advances). */
int aasr_feat_eof_frame(const aasr_feat *h, int64_t n_samples);

/* Graphs whose first module is a `pre` module (PreModule,
 * aku/FeatureModules.cc:572-755) take float feature frames instead of audio:
 * what `feacat --raw-output -H` writes (int32 dimension, then float32 frames;
 * legacy_file 1: a one-byte dimension).  Frames before 0 repeat frame 0, frames
 * past the end repeat the last one.  n_values = frames x input dimension.
 * aasr_feat_last_frame and the batch entry points count such input in int16
 * units (2 per float) so that audio and feature input share every buffer. */
aasr_status aasr_feat_run_features(aasr_feat *h, const float *features, int64_t n_values,
                                   int32_t first_frame, int32_t n_frames,
                                   const char *module_name, float *out);
aasr_status aasr_feat_run_features_f64(aasr_feat *h, const float *features, int64_t n_values,
                                       int32_t first_frame, int32_t n_frames,
                                       const char *module_name, double *out);
/* FeatureGenerator::write_configuration (aku/FeatureGenerator.cc:222-243): the graph as .cfg text in
 * the reference's own layout (every option a module saves, "%g" / "%d" values, "sources" last), so
 * that re-reading it rebuilds the same graph.  *text is malloc'ed; free it with aasr_free. */
aasr_status aasr_feat_write_config(const aasr_feat *h, char **text, int64_t *len);
int aasr_feat_input_is_features(const aasr_feat *h);   /* 1 when the base module is `pre` */
int aasr_feat_pre_legacy(const aasr_feat *h);          /* its legacy_file option */
int aasr_feat_input_dim(const aasr_feat *h);           /* dimension of the base module */

/* FeatureGenerator::generate(frame) for frames first_frame ..
 * first_frame+n_frames-1 of one utterance whose complete PCM (mono int16,
 * what AudioReader::fetch delivers, aku/AudioReader.cc:219-230) is pcm[0..
 * n_samples).  Negative frames and frames past EOF follow copy_borders
 * (aku/FeatureModules.cc:381-397).  out is float32 [n_frames x dim].
 * module_name NULL = last module (the generator's output). */
aasr_status aasr_feat_run(aasr_feat *h, const int16_t *pcm, int64_t n_samples,
                          int32_t first_frame, int32_t n_frames,
                          const char *module_name, float *out);
aasr_status aasr_feat_run_dev(aasr_feat *h, const int16_t *d_pcm,
                              int64_t n_samples, int32_t first_frame,
                              int32_t n_frames, float *d_out, void *stream);
/* double-precision output of the same frames (the reference's FeatureVec is
 * double, aku/FeatureBuffer.hh:15-89); used by the adapters and parity tests */
aasr_status aasr_feat_run_f64(aasr_feat *h, const int16_t *pcm,
                              int64_t n_samples, int32_t first_frame,
                              int32_t n_frames, const char *module_name,
                              double *out);

/* Batched form for a recipe slice: n_utts utterances concatenated in d_pcm;
 * utterance u occupies samples [pcm_off[u], pcm_off[u+1]) and emits frames
 * 0 .. last_frame(u) into d_out rows [frame_off[u], frame_off[u+1]).
 * pcm_off/frame_off are HOST arrays of n_utts+1 entries. */
aasr_status aasr_feat_run_batch_dev(aasr_feat *h, const int16_t *d_pcm,
                                    const int64_t *pcm_off,
                                    const int64_t *frame_off, int32_t n_utts,
                                    float *d_out, void *stream);

/* FeatureModule::set_parameters for "normalization" / "lin_transform"
 * (aku/FeatureModules.cc:1094-1117, 1188-1196): params_text is a
 * "{ key value ... }" block as in .spkc files. */
aasr_status aasr_feat_set_parameters(aasr_feat *h, const char *module_name,
                                     const char *params_text);
/* FeatureModule::get_parameters (aku/FeatureModules.cc, per module): the module's adaptation
 * parameters as the same kind of block ("%g" values).  *text is malloc'ed; free it with aasr_free. */
aasr_status aasr_feat_get_parameters(const aasr_feat *h, const char *module_name, char **text,
                                     int64_t *len);
/* FeatureGenerator::module(name) bookkeeping: modules in configuration order
 * (FeatureModule::name / type_str, aku/FeatureModule.hh:47-154); NULL past the end */
int aasr_feat_num_modules(const aasr_feat *h);
const char *aasr_feat_module_name(const aasr_feat *h, int index);
const char *aasr_feat_module_type(const aasr_feat *h, int index);

/* ------------------------------------------------------------------------ */
/* Acoustic model: replaces aku::HmmSet / PDFPool / Mixture scoring          */
/* ------------------------------------------------------------------------ */

/* In-memory construction.  mean/var are [G x dim] row-major doubles as
 * DiagonalGaussian::read stores them (aku/Distributions.cc:1131-1150;
 * var<=0 -> precision 0); mixtures in CSR form: state s owns components
 * mix_off[s] .. mix_off[s+1]-1 with pool indices mix_idx[] (arbitrary, tied
 * pools allowed) and weights mix_w[] (renormalised to sum 1 like
 * Mixture::read, aku/Distributions.cc:2418-2434). */
aasr_status aasr_gmm_create_diag(int32_t dim, int32_t num_gaussians,
                                 const double *mean, const double *var,
                                 int32_t num_states, const int32_t *mix_off,
                                 const int32_t *mix_idx, const double *mix_w,
                                 aasr_gmm **out);
/* Full-covariance pool (FullCovarianceGaussian::read / set_covariance,
 * aku/Distributions.cc:1466-1488, 1559-1586): cov is [G x dim x dim] row-major.
 * Non-SPD covariances become the reference's "invalid" Gaussian (precision and
 * constant 0).  Scored like PDFPool::precompute_likelihoods' exponential-form
 * branch (:2664-2680) but through the Cholesky factor, see DESIGN.md. */
aasr_status aasr_gmm_create_full(int32_t dim, int32_t num_gaussians,
                                 const double *mean, const double *cov,
                                 int32_t num_states, const int32_t *mix_off,
                                 const int32_t *mix_idx, const double *mix_w,
                                 aasr_gmm **out);
/* HmmSet::read_all(base) = read_mc + read_ph + read_gk
 * (aku/HmmSet.cc:351-357); individual paths like phone_probs -g -m -p.
 * ph_path may be NULL (state count = mixture count). */
aasr_status aasr_gmm_create_from_files(const char *gk_path, const char *mc_path,
                                       const char *ph_path, aasr_gmm **out);
/* Binary model cache (new): everything the text parser of
 * aasr_gmm_create_from_files produced, in double precision, with a checksum.
 * A model created from the cache scores bit-identically to one created from
 * the .gk/.mc/.ph files it was written from; the text parse (seconds for a
 * 50 000-Gaussian pool, paid by every per-GPU process of a run) is skipped. */
aasr_status aasr_gmm_write_cache(const aasr_gmm *h, const char *cache_path);
aasr_status aasr_gmm_create_from_cache(const char *cache_path, aasr_gmm **out);
/* The same, for a caller that names the text files too (phone_probs --model-cache):
 * a cache records (size, content hash) of the .gk/.mc/.ph it was written from and
 * is refused (AASR_ERR_INVALID, "stale model cache") when they differ from the
 * files given here -- retraining, or another -b next to the same cache path, then
 * falls back to aasr_gmm_create_from_files instead of scoring with the old model. */
aasr_status aasr_gmm_create_from_cache_checked(const char *cache_path, const char *gk_path,
                                               const char *mc_path, const char *ph_path,
                                               aasr_gmm **out);
void aasr_gmm_destroy(aasr_gmm *h);

/* Model structure for host-side views (aku::Mixture::size / get_base_pdf_index /
 * get_mixture_coefficient, aku/Distributions.hh:795-812; Gaussian::get_mean /
 * get_covariance of a diagonal Gaussian): weights are the normalised ones
 * (Mixture::read -> normalize_weights), index[] / weight[] take
 * aasr_gmm_mixture_size(h, state) values, mean[] / var[] take dim values. */
int32_t aasr_gmm_mixture_size(const aasr_gmm *h, int32_t state);
aasr_status aasr_gmm_mixture_get(const aasr_gmm *h, int32_t state, int32_t *index, double *weight);
aasr_status aasr_gmm_gaussian_get(const aasr_gmm *h, int32_t gaussian, double *mean, double *var);

int aasr_gmm_dim(const aasr_gmm *h);            /* HmmSet::dim()        */
int aasr_gmm_num_states(const aasr_gmm *h);     /* HmmSet::num_states() */
int aasr_gmm_num_gaussians(const aasr_gmm *h);  /* PDFPool::size()      */
/* rows of the component-expanded layout the kernel streams (>= sum n_s) */
int64_t aasr_gmm_expanded_rows(const aasr_gmm *h);

/* Model-side constrained MLLR (ConstrainedMllr::load_transform /
 * AdaptedGaussian, aku/ModelModules.cc:164-232, aku/ModelModules.hh:128-212):
 * pool Gaussian g scores A_t f + b_t instead of f, t = gauss_to_transform[g]
 * (-1 = unadapted), and its likelihood is multiplied by |prod diag A_t| -- the
 * reference's full_matrix_determinant (aku/LinearAlgebra.cc:73-86) returns the
 * product of A's diagonal, kept for parity.  W is [n][dim][dim+1] row-major with
 * column 0 = b_t and columns 1..dim = A_t, the layout of the reference's W
 * matrices.  One transform shared by every Gaussian is applied to the frames
 * (cost of a dim x dim product per frame, like the reference); per-class
 * transforms are folded into per-Gaussian factor rows and scored by the
 * full-covariance kernel.  n_transforms = 0 removes the adaptation
 * (ConstrainedMllr::reset_transform). */
aasr_status aasr_gmm_set_cmllr(aasr_gmm *h, int32_t n_transforms,
                               const int32_t *gauss_to_transform, const double *W);

/* Gaussian clustering: HmmSet::read_clustering / PDFPool::read_clustering
 * (aku/HmmSet.cc:1353-1357, aku/Distributions.cc:3114-3170) and
 * HmmSet::set_clustering_min_evals (aku/HmmSet.cc:1359-1366) -- what
 * phone_probs -C FILE --eval-minc R --eval-ming R sets up
 * (aku/phone_probs.cc:112-117) and PPToolbox::set_clustering
 * (aku/PhoneProbsToolbox.cc:50-53).
 * Once enabled, scoring follows the cluster branch of
 * PDFPool::precompute_likelihoods (aku/Distributions.cc:2684-2722): per frame the
 * cluster centres (unit-weight merges of their members, diagonal) are ranked;
 * members of the best clusters are evaluated exactly until int(min_clusters *
 * clusters) clusters and int(min_gaussians * pool size) Gaussians are done;
 * every other Gaussian takes its centre's likelihood, except where that is 0 in
 * double precision or the Gaussian is in no cluster (PDFPool::compute_likelihood
 * re-evaluates cached values <= 0, aku/Distributions.cc:2636-2644).
 *
 * aasr_gmm_read_clustering reads a .gcl file ("clusters" then "gaussian
 * cluster" pairs).  Like the reference's reader it counts the LAST pair of the
 * file twice (its while(in) loop runs once more on stale operands), which
 * weights that Gaussian double in its centre and in the Gaussian count.
 * aasr_gmm_set_clustering takes the pairs literally (n_clusters = 0 removes the
 * clustering).  Built for diagonal pools (any constrained-MLLR adaptation) and unadapted full-covariance pools, up to
 * 16384 clusters (beyond 4096 every frame takes the slower replay of the reference's priority queue); more than
 * 0.3 * pool size clusters is rejected like the reference does. */
aasr_status aasr_gmm_read_clustering(aasr_gmm *h, const char *gcl_path);
aasr_status aasr_gmm_set_clustering(aasr_gmm *h, int32_t n_clusters, int64_t n_pairs,
                                    const int32_t *gauss_index, const int32_t *cluster_index);
aasr_status aasr_gmm_set_clustering_min_evals(aasr_gmm *h, double min_clusters,
                                              double min_gaussians);
/* PDFPool::number_of_clusters(); 0 without a clustering */
int32_t aasr_gmm_num_clusters(const aasr_gmm *h);

/* Arithmetic used for the frame x Gaussian quadratic forms.
 *  AASR_PREC_F32          f32 matrix-core contraction of the expanded form; models
 *                         whose conditioning would break the 1e-4 budget are
 *                         switched to the centred form automatically
 *  AASR_PREC_F32_CENTRED  always the centred form (x-mu)^2*p on the vector ALU,
 *                         the reference's own arithmetic shape in f32
 *  AASR_PREC_BF16X3       both operands split into three bf16 terms, six
 *                         bf16 matrix-core products per f32 product accumulated
 *                         in f32: f32-class accuracy (same 1e-4 parity bar) at
 *                         ~1.8x the speed of the f32 kernel; diagonal, full-
 *                         covariance and per-class CMLLR models (ill-conditioned
 *                         models still take the centred form).  The environment
 *                         variable AASR_PREC=0 selects AASR_PREC_F32 globally.
 *  AASR_PREC_F16X2        default: both operands as two fp16 terms (22 bits), three
 *                         fp16 matrix-core products per product -- half the matrix
 *                         instructions of BF16X3 at 1.4-1.8x its rounding error, so it
 *                         is used only for models whose conditioning estimate leaves
 *                         that room (same 1e-4 bar, tighter limits: gmm.h
 *                         KAPPA_LIMIT_F16 for diagonal pools, FULL_KAPPA_LIMIT_F16 for the
 *                         factor rows of full-covariance / subspace pools); every other
 *                         model runs as under AASR_PREC_BF16X3.
 *                         aasr_gmm_effective_precision tells which form a model got.
 *  AASR_PREC_F64          the reference's own arithmetic in double, operation by operation (diagonal
 *                         pools; unadapted, under one global CMLLR transform or under per-class transforms, with or
 *                         without clustering): a verification / training-side
 *                         mode, ~1.2 M frames/s at 50 k Gaussians.  Float entry points widen the frames
 *                         and round the scores once; aasr_gmm_score_f64 takes and returns doubles;
 *                         aasr_run_utterance / aasr_run_recipe then run the whole path in double
 *                         (features, scoring, the LNA tail as written) -- AASR_PREC=1 in the
 *                         environment selects it for the command-line tools. */
enum { AASR_PREC_F32 = 0, AASR_PREC_F64 = 1, AASR_PREC_F32_CENTRED = 2, AASR_PREC_BF16X3 = 3, AASR_PREC_F16X2 = 4 };
aasr_status aasr_gmm_set_precision(aasr_gmm *h, int prec);
int aasr_gmm_get_precision(const aasr_gmm *h);
/* the arithmetic the matrix scoring path of this model actually runs under the current setting */
int aasr_gmm_effective_precision(const aasr_gmm *h);
/* Per-state precision routing (new; no aku counterpart -- the reference scores everything in double,
 * aku/Distributions.cc:1040-1062).  States are independent output columns (Mixture::compute_likelihood,
 * aku/Distributions.cc:2078-2086), so under AASR_PREC_F16X2 a diagonal model whose Gaussians do not ALL satisfy the
 * two-term form's conditioning limits is scored in two sections: the states whose Gaussians all qualify with two fp16
 * terms, the others with three bf16 terms (one Gaussian over the limit costs its state the slower arithmetic, not the
 * model).  states_f16x2: how many of the model's states the two-term rows cover under the current setting (0 ... S);
 * states_probe_moved: how many the load-time probe took out of that form (a few hundred frames on the model's own
 * Gaussians, incl. +-6 sigma, scored in f16x2 and in exact f32 when the model is created; a state that differs by more
 * than 5e-5 is scored with three terms).  Either pointer may be null. */
aasr_status aasr_gmm_precision_states(const aasr_gmm *h, int64_t *states_f16x2, int64_t *states_probe_moved);

/* HmmSet::precompute_likelihoods + state_likelihood for a block of frames
 * (aku/HmmSet.cc:484-501, aku/HmmSet.hh:309): frames float32 [F x dim];
 * state_loglik float32 [F x S] = log(max(sum_k w_k exp(ll_k), 1e-50)). */
aasr_status aasr_gmm_score(aasr_gmm *h, const float *frames, int64_t F,
                           float *state_loglik);
/* AASR_PREC_F64 with double frames in and double log state likelihoods out (any precision setting) */
aasr_status aasr_gmm_score_f64(aasr_gmm *h, const double *frames, int64_t F, double *state_loglik);
aasr_status aasr_gmm_score_f64_dev(aasr_gmm *h, const double *d_frames, int64_t F, double *d_state_loglik, void *stream);
aasr_status aasr_gmm_score_dev(aasr_gmm *h, const float *d_frames, int64_t F,
                               float *d_state_loglik, void *stream);

/* PDFPool::precompute_likelihoods (aku/Distributions.cc:2647-2682): the
 * log-likelihood of every pool Gaussian, float32 [F x G]. */
/* The same with a row pitch (floats between consecutive frame rows, >= S) for callers that keep the
 * score matrix on the device: rows padded to a multiple of 32 floats turn every 128-byte output
 * group of the scoring kernel into one whole L2 line (1.1 ms of 32.4 per 10^6 frames x 50 k
 * Gaussians, a quarter less HBM write traffic).  Only the track kernels (f32 and bf16x3) write pitched rows: aasr_gmm_score_pitch_ok() says
 * whether this model / precision does; otherwise pitch must equal the state count. */
int aasr_gmm_score_pitch_ok(const aasr_gmm *h);
aasr_status aasr_gmm_score_dev_pitched(aasr_gmm *h, const float *d_frames, int64_t F,
                                       float *d_state_loglik, int64_t pitch, void *stream);

aasr_status aasr_gmm_gauss_loglik(aasr_gmm *h, const float *frames, int64_t F,
                                  float *gauss_loglik);
aasr_status aasr_gmm_gauss_loglik_dev(aasr_gmm *h, const float *d_frames,
                                      int64_t F, float *d_gauss_loglik,
                                      void *stream);

/* ------------------------------------------------------------------------ */
/* LNA: replaces the frame loop tail of aku/phone_probs.cc:224-262           */
/* ------------------------------------------------------------------------ */

/* state_loglik float32 [F x S] (output of aasr_gmm_score) -> normalised
 * float log-probabilities and LNA bytes.  Emulates the reference's float
 * storage of linear likelihoods (values below FLT_TRUE_MIN flush to 0 ->
 * log(1e-50); denormal quantisation), Z = sum over states (Z==0 or
 * !normalize -> 1), safe_log, then 2-byte big-endian (int)(-1820*lp+.5)
 * (0xFFFF below -36.008) or 4-byte little-endian float.
 * bytes_out: [F x S x lnabytes] (may be NULL); lp_out: [F x S] (may be NULL) */
aasr_status aasr_lna_encode(const float *state_loglik, int64_t F, int32_t S,
                            int normalize, int lnabytes, float *lp_out,
                            uint8_t *bytes_out);
/* device input with a row pitch (see aasr_gmm_score_dev_pitched); outputs are dense */
aasr_status aasr_lna_encode_dev_pitched(const float *d_state_loglik, int64_t in_pitch, int64_t F,
                                        int32_t S, int normalize, int lnabytes, float *d_lp_out,
                                        uint8_t *d_bytes_out, void *stream);
aasr_status aasr_lna_encode_dev(const float *d_state_loglik, int64_t F,
                                int32_t S, int normalize, int lnabytes,
                                float *d_lp_out, uint8_t *d_bytes_out,
                                void *stream);
/* Frames straight to LNA codes on the device: aasr_gmm_score_dev followed by
 * aasr_lna_encode_dev, except that the engine may keep the state scores in its
 * own layout in between (rows padded to whole cache lines; a model whose conditioning needs several pivots -- or three
 * terms for part of its states, aasr_gmm_precision_states -- as internal models over disjoint sets of its states, each in
 * its own column range, read back through a column map).  d_scratch takes aasr_gmm_score_scratch_floats(h, F) floats: a
 * property of the MODEL -- it does not change with aasr_gmm_set_precision, clustering or transforms, so a buffer sized
 * once stays valid for the handle's life --, d_bytes_out F * num_states * lnabytes bytes.  What the recipe driver runs
 * per block. */
int64_t aasr_gmm_score_scratch_floats(const aasr_gmm *h, int64_t F);
aasr_status aasr_gmm_score_lna_dev(aasr_gmm *h, const float *d_frames, int64_t F, int normalize, int lnabytes,
                                   float *d_scratch, uint8_t *d_bytes_out, void *stream);

/* 5-byte file header: big-endian uint32 S + 1 byte lnabytes
 * (aku/phone_probs.cc:32-43, 213-214) */
void aasr_lna_header(int32_t num_states, int lnabytes, uint8_t out[5]);

/* Reads an LNA file the way the recogniser's reader does (decoder/src/LnaReaderCircular.cc:
 * header :63-96, frame decoding :166-198): 4-byte little-endian floats, 2-byte big-endian codes
 * (code / -1820.0) or the legacy 1-byte codes (code / -24.0) -- phone_probs never writes the
 * 1-byte form, old acoustic files hold it.  Host only.  A trailing partial frame is dropped, as
 * the reader's fread does.  *log_probs is malloc'ed [*frames x *num_states]; free with aasr_free. */
aasr_status aasr_lna_read_file(const char *path, int32_t *num_states, int32_t *lnabytes, int64_t *frames,
                               float **log_probs);

/* ------------------------------------------------------------------------ */
/* Recipe + whole-path driver: replaces the body of phone_probs main()       */
/* ------------------------------------------------------------------------ */

/* Recipe::read batch selection (aku/Recipe.cc:23-149): returns in
 * first_line/num_lines the contiguous slice of the L non-empty,
 * non-comment recipe lines that batch batch_index (1-based) of num_batches
 * receives (cluster_speakers=false form). */
aasr_status aasr_recipe_batch_range(int32_t num_lines_total, int32_t num_batches,
                                    int32_t batch_index, int32_t *first_line,
                                    int32_t *num_lines);

/* Recipe::read itself (aku/Recipe.cc:23-149), host only: parses recipe text and
 * returns the utterances of one batch as a malloc'ed text table (aasr_free), one
 * line per utterance with the fields audio, lna, speaker, utterance, start-time,
 * end-time separated by 0x1f (times are the float fields of Recipe::Info, printed "%.9g").  Kept from the reference: lines
 * are cleaned of " \t\n" only, fields split on blanks and tabs, `key=value`
 * through str::split (a trailing '=' is dropped), keys persist across lines. */
aasr_status aasr_recipe_read(const char *recipe_text, int32_t num_batches, int32_t batch_index,
                             char **table_out, int64_t *table_len);

/* The same with every field of Recipe::Info (aku/Recipe.hh:40-52) and the cluster_speakers
 * flag of Recipe::read (a batch only ends where the speaker changes, aku/Recipe.cc:86-101):
 * audio, alt-audio, transcript, alignment, hmmnet, den-hmmnet, lna, start-time, end-time,
 * start-line, end-line, speaker, utterance -- 13 fields separated by 0x1f per line. */
aasr_status aasr_recipe_read_all(const char *recipe_text, int32_t num_batches, int32_t batch_index,
                                 int32_t cluster_speakers, char **table_out, int64_t *table_len);

/* start/end frame of an utterance as phone_probs derives them from the recipe's
 * start-time / end-time (aku/phone_probs.cc:199-206): `(int)(time * frame_rate)`
 * with BOTH operands float (Recipe::Info::start_time is a float field,
 * aku/Recipe.hh:48-49; FeatureGenerator::frame_rate() returns float), so the
 * product is rounded to float before truncation; end frame 0 means "to the end"
 * and is returned as INT32_MAX.  Host only. */
void aasr_recipe_frame_limits(float start_time, float end_time, float frame_rate,
                              int32_t *start_frame, int32_t *end_frame);

/* ---------------------------------------------------------------------------
 * User-defined feature module types (the plugin side of aku::FeatureModule,
 * aku/FeatureModule.hh:47-154: a subclass with set_module_config / generate(frame),
 * one more `else if` in FeatureGenerator::load_configuration, aku/FeatureGenerator.cc:145-175).
 * A registered type may appear in any .cfg after the base module.  Its frames are
 * computed by the callback ON THE HOST: when the graph is evaluated, the rows of its
 * sources are copied from the device, `generate` runs once per frame, the result goes
 * back to the device and the modules behind it continue there -- an escape hatch for
 * experiments, not a fast path (the 16 built-in types are kernels).  A built-in
 * name cannot be taken.  The adapter class aku::FeatureModule
 * (aaltoasr_amd/csrc/aku/FeatureModule.hh) wraps this for C++ subclasses.
 *   configure: parse the module's option block ("{\n name value\n ... }\n"), report
 *     the output dimension and how many frames to the left / right of the current
 *     one the module reads from its sources; *instance is handed back to the other
 *     callbacks.  Return 0, or nonzero with a message in err.
 *   generate: sources[k] points at source k's frames frame-left .. frame+right,
 *     row-major [left+right+1][source_dims[k]] doubles (frames outside the file are
 *     the border copies the chain defines); write dim doubles to out. */
typedef struct aasr_host_module {
  int (*configure)(void *user, const char *module_name, const char *options_block, int32_t n_sources,
                   const int32_t *source_dims, int32_t *dim, int32_t *left, int32_t *right, void **instance,
                   char *err, int32_t err_len);
  int (*generate)(void *instance, int32_t frame, const double *const *sources, double *out, char *err,
                  int32_t err_len);
  void (*destroy)(void *instance);
} aasr_host_module;
aasr_status aasr_feat_register_module_type(const char *type_name, const aasr_host_module *vtbl, void *user);

/* ---------------------------------------------------------------------------
 * Speaker / utterance configuration: aku::SpeakerConfig
 * (aku/SpeakerConfig.hh:15-60, aku/SpeakerConfig.cc) -- what phone_probs -S FILE
 * drives (aku/phone_probs.cc:94-95, 191-196).  A .spkc file holds, per speaker
 * and per utterance (and for the "default" of each), parameter blocks for named
 * feature modules ("feature NAME" or just "NAME": normalization, lin_transform,
 * vtln, sr_norm, quanteq take parameters) and for the model module "cmllr"
 * (constrained MLLR matrices w1, w2, ... with unitmode UNIT_NO / UNIT_GAUSSIAN /
 * UNIT_MIX / UNIT_PHONE; the engine maps them onto aasr_gmm_set_cmllr).
 * set_speaker / set_utterance follow the reference step by step, including the
 * read-back of the current speaker's parameters through "%g" before a switch;
 * see aaltoasr_amd/csrc/speaker_config.cc for the list of kept quirks.
 * The handle borrows feat and gmm (gmm may be NULL, or given later with
 * aasr_spkc_set_model -- phone_probs reads the speaker file before the model). */
typedef struct aasr_spkc aasr_spkc;
aasr_status aasr_spkc_create(aasr_feat *feat, aasr_gmm *gmm, aasr_spkc **out);
void aasr_spkc_destroy(aasr_spkc *h);
aasr_status aasr_spkc_set_model(aasr_spkc *h, aasr_gmm *gmm);
/* SpeakerConfig::read_speaker_file */
aasr_status aasr_spkc_read_file(aasr_spkc *h, const char *path);
aasr_status aasr_spkc_read_text(aasr_spkc *h, const char *text);
/* SpeakerConfig::set_speaker / set_utterance; "" (or NULL) selects the default */
aasr_status aasr_spkc_set_speaker(aasr_spkc *h, const char *speaker_id);
aasr_status aasr_spkc_set_utterance(aasr_spkc *h, const char *utterance_id);
/* SpeakerConfig::write_speaker_file (aku/SpeakerConfig.cc:156-236): the speaker file as text, with
 * the current speaker's / utterance's module parameters fetched back first (what the adaptation
 * tools -- vtln, mllr -- write after estimating).  speakers / utterances filter the entries by id
 * ("default" names the default entries); a count < 0 writes all of them.  *text_out is malloc'ed. */
aasr_status aasr_spkc_write_text(aasr_spkc *h, const char *const *speakers, int32_t n_speakers,
                                 const char *const *utterances, int32_t n_utterances, char **text_out,
                                 int64_t *text_len);
/* number of times a module's device parameters were actually rewritten */
int64_t aasr_spkc_num_changes(const aasr_spkc *h);

typedef struct aasr_run_options {
  int32_t lnabytes;        /* 2 or 4          (--lnabytes)          */
  int32_t normalize;       /* 0 = -N / --no-normalization           */
  int32_t num_batches;     /* -B (0/1 = no batching)                */
  int32_t batch_index;     /* -I, 1-based                           */
  int32_t no_overwrite;    /* -n: skip utterances whose LNA exists  */
  int32_t raw_audio;       /* treat inputs as headerless PCM16      */
  int32_t info;            /* -i verbosity                          */
  int32_t afname;          /* -a: name outputs after the audio file */
  const char *out_dir;     /* -o: prefix for LNA paths or NULL      */
  struct aasr_spkc *speakers; /* -S: speaker configuration or NULL   */
  int32_t sort_recipe;     /* --sort-recipe: stable sort of the slice by speaker id
                              (Recipe::sort_infos, aku/Recipe.hh:86-88,115-117) */
} aasr_run_options;

typedef struct aasr_run_stats {
  int64_t utterances;
  int64_t frames;
  double seconds_total;
  double seconds_device;   /* input upload + feature + scoring + LNA kernels (device events) */
  double seconds_copy_out; /* packed rows device -> host; overlaps the next block's kernels */
} aasr_run_stats;

/* phone_probs main loop (aku/phone_probs.cc:145-267) for one recipe slice on
 * the current device: read audio, features, scoring, LNA files. */
aasr_status aasr_run_recipe(aasr_feat *feat, aasr_gmm *gmm,
                            const char *recipe_path,
                            const aasr_run_options *opt, aasr_run_stats *stats);

/* One engine process per GPU: `processes` of them share this host's cores.  The reference scales
 * out the same way -- N independent phone_probs processes on recipe slices (-B n -I k,
 * aku/Recipe.cc:63-115, aku/phone_probs.cc:136-141) -- each single-threaded; here a process runs
 * reader / writer helper threads, and their number is usable_cores / processes (0 = take
 * AASR_LOCAL_RANKS or torchrun's LOCAL_WORLD_SIZE from the environment, else 1). */
aasr_status aasr_set_host_share(int32_t processes);
/* cores this process may use: affinity mask capped by the cgroup CPU quota */
int32_t aasr_host_usable_cores(void);

/* Where the last aasr_run_recipe on this model handle spent its wall time (seconds): what the
 * calling thread waited for, what the two device streams were busy with, and the helper-thread
 * sizing that was in force.  Diagnostics for multi-rank runs; the reference has no counterpart
 * (its loop is serial, aku/phone_probs.cc:145-267). */
typedef struct aasr_recipe_timing {
  double seconds_total;
  double wait_reader;       /* calling thread idle: next utterance not read yet           */
  double wait_result_slot;  /* ... idle: both pinned result slots still being written out */
  double enqueue;           /* ... enqueueing a block (incl. pageable uploads)            */
  double wait_copies;       /* ... waiting for a block's device -> host copy              */
  double device;            /* compute stream busy (upload + features + scoring + LNA)    */
  double copy_out;          /* copy stream busy (packed rows device -> host)              */
  int32_t writer_threads;
  int32_t usable_cores;
  int32_t host_share;
} aasr_recipe_timing;
aasr_status aasr_recipe_last_timing(const aasr_gmm *gmm, aasr_recipe_timing *out);

/* PPToolbox::generate_from_file_to_fd equivalent for one utterance
 * (aku/PhoneProbsToolbox.cc:135-208: lnabytes 2, normalised): returns a
 * malloc'ed LNA image (header + frames) the caller frees with aasr_free. */
aasr_status aasr_run_utterance(aasr_feat *feat, aasr_gmm *gmm,
                               const int16_t *pcm, int64_t n_samples,
                               int32_t start_frame, int32_t end_frame,
                               int normalize, int lnabytes, uint8_t **lna_out,
                               int64_t *lna_len, int64_t *frames_out);
void aasr_free(void *p);

/* Audio input of the audiofile module, host only (no device needed).  Replaces
 * AudioReader::open + check_audio_parameters + read_from_file
 * (aku/AudioReader.cc:86-110, 145-156, 170-213) and AudioFileModule::set_fname's
 * sample-rate check (aku/FeatureModules.cc:244-262): RIFF/WAVE, AU, AIFF/AIFF-C
 * and NIST SPHERE files holding integer PCM or G.711, converted to 16-bit the
 * way sf_read_short() does; anything else is read as headerless PCM16 in the
 * module's byte order -- the reference's fallback.  `feat` supplies the
 * module's `sample_rate`, `raw` and `endian` options (NULL: no rate check,
 * container detection on, little endian).  *pcm is malloc'ed; free it with
 * aasr_free. */
aasr_status aasr_audio_read(const aasr_feat *feat, const char *path, int16_t **pcm,
                            int64_t *n_samples, int32_t *sample_rate);

/* The same decoding for input already in memory -- what FeatureGenerator::open(FILE*, ...) /
 * open_fd (aku/FeatureGenerator.cc:54-84) and PPToolbox::generate_to_fd (aku/PhoneProbsToolbox.cc:
 * 55-82) read from a descriptor.  For graphs that start with a `pre` module the data is a feature
 * file and *pcm receives the engine's input units for it (see aasr_feat_input_is_features). */
aasr_status aasr_audio_decode(const aasr_feat *feat, const void *data, int64_t n_bytes, int16_t **pcm,
                              int64_t *n_samples, int32_t *sample_rate);

#ifdef __cplusplus
}
#endif
#endif /* AASR_H */
