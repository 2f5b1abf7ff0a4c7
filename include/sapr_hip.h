/* sapr_hip.h — C ABI of libsapr_hip.so, the MI355X (gfx950) implementation of the
 * data-parallel hot path of frankcholula/sapr assignment2 (MFCC front-end +
 * Gaussian-HMM Viterbi / forward-backward).
 *
 * The reference has no FFI layer: its hot path is plain Python (numpy, hmmlearn,
 * librosa).  Each entry point below names the reference call (file:line under
 * /root/reference/assignment2) whose arithmetic it replaces; the ctypes stub a
 * maintainer adds on the reference side is shown in INTEGRATION.md and shipped in
 * sapr_amd/_lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - no torch / HIP types in signatures: `stream` is a hipStream_t passed as void*
 *     (NULL = default stream); launches are asynchronous on that stream;
 *   - feature batches are ragged and frame-major: feats[total_frames][D] float32,
 *     offsets[n_utts+1] int64 (utterance u owns frames offsets[u] .. offsets[u+1]-1).
 *     This is the transpose of the reference's per-utterance (D,T) numpy arrays
 *     (mfcc_extract.py:15-24; decoder.py:59 already hands hmmlearn the (T,D) view);
 *   - word models are float64 arrays means[W][S][D], vars[W][S][D], gconst[W][S], log_start[W][S],
 *     log_trans[W][S][S] (W word models, S states), packed once by sapr_diag_pack;
 *   - return value: 0 on success, <0 argument/shape error, >0 hipError_t;
 *     sapr_last_error() returns a thread-local message for the last failure.
 */
#ifndef SAPR_HIP_H
#define SAPR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAPR_ABI_VERSION 2

/* transition topology of a model pack */
#define SAPR_TOPO_DENSE 0  /* any S x S log_trans */
#define SAPR_TOPO_BIDIAG 1 /* only log_trans[i][i] and log_trans[i][i+1] are > -inf
                              (hmmlearn_hmm.py:45-78, custom_hmm.py:94-116) */

/* back-trace tie-break of GaussianHMM.decode (see oracle/hmmlearn_oracle.py) */
#define SAPR_TIE_LOW 0  /* equal scores -> lower predecessor index  (Cython _argmax, hmmlearn <= 0.2.7) */
#define SAPR_TIE_HIGH 1 /* equal scores -> higher predecessor index (std::max over (value,index), 0.3.x) */

/* order of the sum over the D feature dimensions inside the log-density: numpy's reduction
 * order depends on the memory layout of the X array hmmlearn receives (see viterbi.hip) */
#define SAPR_SUM_PAIRWISE 0 /* X is a C-contiguous (T,D) array: fit/score, hmmlearn_hmm.py:80-81 */
#define SAPR_SUM_TVIEW 1    /* X is the transposed view of a (D,T) array, decoder.py:59: left-to-right
                               sum when T > 1, pair-wise when T == 1 */
#define SAPR_SUM_SEQ 2      /* left-to-right sum for every utterance: numpy's order for fewer than 8 dimensions in
                               either layout — what a model narrower than 8 dimensions needs when it runs padded
                               to an instantiated width (sapr_amd/trellis.py kernel_dims) */

/* bits of sapr_diag_pack's *pack_flags output, passed on to the decode entry points */
#define SAPR_PACK_FAST_DIV 1 /* parameters inside the proven domain of the FMA-based exactly-rounded division */
#define SAPR_PACK_BOUND_OK 2 /* variances in [1e-20, 1e20]: the pruned decoder's float32 bounding pass is valid */
#define SAPR_ESTEP_STAGED 256 /* sapr_estep_diag only, OR-ed into its fast_div argument: the workspace still holds the
                                 slot-major feature copy a previous call made for the SAME feats / offsets / slot_utt
                                 (the features do not change between EM iterations) */
#define SAPR_PACK_GEMM_OK 4  /* the bounding pass may run on the matrix cores (finite coefficients; states without a
                                self-loop only at chain positions 0, 4, 8, 12) */
#define SAPR_PACK_BIDIAG 8   /* every log_trans entry off the i -> i, i -> i + 1 band is -inf (hmmlearn_hmm.py:45-78
                                topology): what sapr_viterbi_decode_pruned walks; it refuses packs without this bit */

#define SAPR_PACK_EXACT_ONLY 16 /* INPUT bit of *pack_flags (the caller sets it before the call; it is echoed back): build
                                   the exact-kernel operands only — what sapr_estep_diag, sapr_forward_diag and the
                                   all-vocabulary Viterbi read — and leave out the bounding-pass operands; BOUND_OK and
                                   GEMM_OK stay clear, so the pruned decoder refuses the pack.  A Baum-Welch loop
                                   (hmmlearn_hmm.py:103 -> base.fit) packs a new model every iteration and never
                                   decodes with it */

#define SAPR_ERR_ARG (-1)
#define SAPR_ERR_UNSUPPORTED (-2)
#define SAPR_ERR_WORKSPACE (-3)

int sapr_abi_version(void);
const char *sapr_last_error(void);
/* number of CUs / wave size / gcnArchName of device `dev`; arch buffer may be NULL */
int sapr_device_info(int dev, int *cu_count, int *wave_size, char *arch, size_t arch_len);
/* self-test of the device arithmetic behind the E-step's two-term log-sum-exp (csrc/lse_unit.h; the CPU check of that
 * header cannot see the hardware's reciprocal estimate, v_ldexp_f64 and v_rndne_f64): for d[i] >= 0
 * out = [exp(-d) | 1 / (1 + exp(-d)) | log(1 + exp(-d)) | exp_unit(-d)], four runs of n doubles */
int sapr_selftest_lse(const double *d, int64_t n, double *out, void *stream);

/* ------------------------------------------------------------------------------------
 * Viterbi decode, diagonal Gaussians, every state emitting.
 * Replaces GaussianHMM.decode(X) as called at decoder.py:43 for ALL W word models of
 * decoder.py:42 at once (log-density: hmmlearn stats.py _log_multivariate_normal_density_diag;
 * lattice + back-trace: hmmlearn _hmmc.cpp viterbi).  Scores are bit-identical to the
 * float64 numpy/C++ evaluation (same operation order, IEEE division, no FMA contraction).
 *
 *   pass 1  sapr_viterbi_diag_scores   scores[n_utts][W], last_state[n_utts][W], back-pointer
 *                                      words in `workspace`
 *   pass 2  sapr_viterbi_backtrace     state path of ONE model per utterance: the arg-max word
 *                                      (decoder.py:42-47, strict '>' in model order) when
 *                                      word_sel == NULL, else model word_sel[u]
 *
 * `order` (optional, may be NULL) is a permutation of utterances, normally sorted by length so
 * that the 64 lanes of a wavefront walk trellises of similar T.
 * ---------------------------------------------------------------------------------- */
int sapr_viterbi_workspace_bytes(int64_t n_utts, int32_t W, int32_t S, int32_t max_T,
                                 int32_t topology, size_t *bytes);

/* Model preparation (once per set of word models, not per batch): interleaves the float64
 * arrays  means[W][S][D], vars[W][S][D] (covars floored at DBL_MIN like hmmlearn stats.py),
 * gconst[W][S] = D*log(2*pi) + sum_d log var, log_start[W][S], log_trans[W][S][S]
 * into one device blob {mean, var, RN(1/var), RN(1/var - RN(1/var))} ... that the kernels read with scalar loads.
 * *pack_flags: SAPR_PACK_FAST_DIV is set when every parameter lies in the domain where the FMA-based
 * exactly-rounded division of viterbi.hip is proven equal to IEEE division (pass it on as `fast_div`; 0
 * selects the IEEE-division instantiation — same bits, slower); SAPR_PACK_BOUND_OK when the pruned decoder
 * may be used.  *pack_flags is read on entry as well: initialise it to 0, or to SAPR_PACK_EXACT_ONLY (above).
 * Synchronises `stream`. */
int sapr_diag_pack_bytes(int32_t W, int32_t S, int32_t D, size_t *bytes);
int sapr_diag_pack(const double *means, const double *vars, const double *gconst,
                   const double *log_start, const double *log_trans, int32_t W, int32_t S, int32_t D,
                   void *pack, size_t pack_bytes, int32_t *pack_flags /* host */, void *stream);

int sapr_viterbi_diag_scores(const float *feats, const int64_t *offsets, const int32_t *order,
                             int64_t n_utts, int32_t D, int32_t max_T, const void *pack,
                             int32_t W, int32_t S, int32_t topology, int32_t tie, int32_t sum_order,
                             int32_t fast_div, void *workspace, size_t workspace_bytes,
                             double *scores, int32_t *last_state, void *stream);

int sapr_viterbi_backtrace(const int64_t *offsets, const int32_t *order, int64_t n_utts,
                           int32_t max_T, int32_t W, int32_t S, int32_t topology,
                           const void *workspace, size_t workspace_bytes,
                           const double *scores, const int32_t *last_state,
                           const int32_t *word_sel, /* NULL -> arg-max over words */
                           int32_t *best_word, double *best_score,
                           int32_t *path /* [total_frames] */, void *stream);

/* Pruned decode: Decoder.decode_sequence (decoder.py:35-49) returns only the best word, its score and its
 * state path, so the exact lattice is evaluated only for the words that can still be the arg-max.
 *   pass A  float32 emission sums (3 instead of 7 VALU instructions per state and dimension, at the float32
 *           rate, no back-pointers) give every word an interval [score - eps, score + eps] that provably
 *           contains its exact score (viterbi.hip states the bound);
 *   pass B  a word is dropped when its interval lies strictly below another word's;
 *   pass C  sapr_viterbi_diag_scores' kernel over the remaining (utterance, word) pairs;
 *   pass D  arg-max among them (first strict maximum in model order) and back-trace.
 * best_word / best_score / path are bit-identical to sapr_viterbi_diag_scores + sapr_viterbi_backtrace
 * (word_sel == NULL).  pack_flags & SAPR_PACK_BIDIAG (bidiagonal topology, scanned by sapr_diag_pack) and
 * pack_flags & SAPR_PACK_BOUND_OK required (SAPR_ERR_UNSUPPORTED otherwise: use the two-call form).  sapr_viterbi_pruned_views exposes the intermediate arrays inside
 * `workspace` ([n_utts][W] each; cand_slot < 0 = dropped; cand_count[W]) for tests and diagnostics. */
int sapr_viterbi_pruned_workspace_bytes(int64_t n_utts, int32_t W, int32_t S, int32_t max_T, size_t *bytes);
int sapr_viterbi_decode_pruned(const float *feats, const int64_t *offsets, const int32_t *order, int64_t n_utts,
                               int32_t D, int32_t max_T, const void *pack, int32_t W, int32_t S, int32_t tie,
                               int32_t sum_order, int32_t pack_flags, void *workspace, size_t workspace_bytes,
                               int32_t *best_word, double *best_score, int32_t *path /* [total_frames] */,
                               void *stream);
int sapr_viterbi_pruned_views(int64_t n_utts, int32_t W, int32_t max_T, void *workspace, double **approx_score,
                              double **approx_eps, double **exact_score, int32_t **cand_slot,
                              int32_t **cand_count);

/* ------------------------------------------------------------------------------------
 * Forward scoring and Baum-Welch E-step, diagonal Gaussians, every state emitting.
 * Replace GaussianHMM.score / the E-step of GaussianHMM.fit as called at hmmlearn_hmm.py:103-104
 * (hmmlearn _hmmc.cpp forward_log, backward_log, compute_log_xi_sum; base.py
 * _compute_posteriors_log; hmm.py _accumulate_sufficient_statistics) for a whole batch.
 *
 * Utterances are presented in TILES of 256 slots that share one word model:
 *   slot_utt[n_tiles*256]  utterance index of each slot, -1 = empty
 *   tile_model[n_tiles]    word model of each tile; tiles sorted by model, and
 *   model_tile_off[W+1]    first tile of each model (prefix offsets).
 * Features are consumed as the C-contiguous (T,D) concatenation hmmlearn_hmm.py:80-81 builds
 * (pair-wise numpy summation order inside the log-density).
 *
 *   sapr_forward_diag  loglik[n_utts]: log P(utterance | its tile's model)
 *   sapr_estep_diag    loglik[n_utts] and stats[W][width], width from sapr_stats_width():
 *                      {n_sequences, sum log-prob, start[S], trans[S][S], post[S], obs[S][D], obs2[S][D]}
 *                      = hmmlearn's stats dict {nobs, -, start, trans, post, obs, obs**2}, reduced
 *                      over each model's utterances in a fixed order (deterministic, no atomics).
 *                      Across GPUs the caller all-reduces `stats` (sum) before the M-step.
 *                      The first call on a batch copies the features into slot-major order inside the
 *                      workspace; later calls on the same batch and workspace may pass
 *                      fast_div | SAPR_ESTEP_STAGED to skip that copy.
 * ---------------------------------------------------------------------------------- */
int sapr_fb_workspace_bytes(int64_t n_utts, int64_t n_tiles, int32_t S, int32_t D, int32_t max_T,
                            size_t *bytes);
int sapr_stats_width(int32_t S, int32_t D, int32_t *width);
int sapr_forward_diag(const float *feats, const int64_t *offsets, const int32_t *slot_utt,
                      const int32_t *tile_model, int64_t n_tiles, int32_t D, const void *pack,
                      int32_t W, int32_t S, int32_t topology, int32_t fast_div, double *loglik,
                      void *stream);
int sapr_estep_diag(const float *feats, const int64_t *offsets, const int32_t *slot_utt,
                    const int32_t *tile_model, const int32_t *model_tile_off, int64_t n_utts,
                    int64_t n_tiles, int32_t D, int32_t max_T, const void *pack, int32_t W, int32_t S,
                    int32_t topology, int32_t fast_div, void *workspace, size_t workspace_bytes,
                    double *loglik, double *stats, void *stream);

/* Flat start of HMMLearnModel (hmmlearn_hmm.py:83-94: np.mean / np.var over axis 0 of the concatenated float32
 * features): numpy adds row after row in float32, so each column is one sequential float32 chain — reproduced
 * bit for bit.  center == NULL: out[d] = sum_r x[r][d]; else out[d] = sum_r RN32(RN32(x[r][d] - center[d])^2).
 * The divisions by N stay with the caller (numpy's own true_divide). */
int sapr_colsum_f32(const float *x, int64_t n_rows, int32_t D, const float *center /* may be NULL */, float *out,
                    void *stream);

/* ------------------------------------------------------------------------------------
 * The reference's from-scratch HMM (custom_hmm.py): non-emitting entry/exit states, full
 * covariances, the Gram-row-sum emission term — every quirk kept (see custom.hip).  Models are
 * float64 arrays prepared on the host exactly as the reference prepares them per call:
 *   means[W][S][D], inv[W][S][D][D] = inv(cov + 1e-6 I), cterm[W][S] = D*log(2*pi) + logdet,
 *   A[W][S][S], logA[W][S][S] = log(A)   (custom_hmm.py:160-165, :191-205).
 *
 *   sapr_custom_estep       custom_hmm.py:146-322,:434-439 per utterance (model utt_model[u], or 0):
 *                           lattices E/alpha/beta/gamma [total_frames][S], optional dense
 *                           xi [total_frames][S][S] (rows t < T-1), and
 *                           utt_out[u] = {LL, scale, agg_gamma[S], agg_xi[S][S]}
 *   sapr_custom_emission_exact  custom_hmm.py:146-174 in the reference's evaluation order (bit-exact)
 *   sapr_custom_decode      custom_hmm.py:462-514 for every (utterance, model): trellis over the first
 *                           Tq frames; scores[n_utts][W], paths[n_utts][W][Tq]
 *   sapr_custom_update_b    custom_hmm.py:366-400 (means, occupancies, raw covariances / occupancy;
 *                           symmetrisation and flooring are host work)
 *   sapr_custom_global_sum / _cov   custom_hmm.py:70-92 (flat-start sums)
 * ---------------------------------------------------------------------------------- */
/* lane_slots: 0 = lattices in the reference's row layout [total_frames][S]; > 0 (>= n_utts) = lattices
 * [max_T][S][lane_slots] with the utterance index fastest (coalesced; gamma then goes to the update_b
 * entry points with the same lane_slots).  In the lane_slots layout with xi == NULL (the batched training path) gamma
 * and utt_out are the outputs: E is filled, the alpha (unshifted) and beta lattices are scratch except for the
 * utterances whose backward half had to run in the reference's own order (custom.hip) */
int sapr_custom_estep(const float *feats, const int64_t *offsets, const int32_t *utt_model, int64_t n_utts,
                      int32_t D, int32_t S, int32_t W, const double *means, const double *inv,
                      const double *cterm, const double *A, const double *logA, int64_t lane_slots, double *E,
                      double *alpha, double *beta, double *gamma, double *xi_dense /* may be NULL */,
                      double *utt_out, void *stream);
/* the slot-major copy of the features, feat_t[max_T][D][lane_slots] float32 (zero past each utterance), and the E-step
 * reading from it: one coalesced row per wavefront and value instead of 64 private 4-byte reads — what the batched
 * training shapes of sapr_custom_estep spend most of their time on.  custom_hmm.py:402-460's loop stages once per
 * call of baum_welch (the features do not change between iterations).  frame_sums[D][lane_slots] float64 (may be NULL
 * in both calls): every utterance's sum over its frames, in frame order — the vector custom_hmm.py:168-172's row sum
 * of the Gram matrix needs; given to the E-step it saves every iteration a pass over the features */
int sapr_custom_stage_features(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t D, int32_t max_T,
                               int64_t lane_slots, float *feat_t, double *frame_sums /* may be NULL */, void *stream);
int sapr_custom_estep_staged(const float *feats, const int64_t *offsets, const int32_t *utt_model, int64_t n_utts,
                             int32_t D, int32_t S, int32_t W, const double *means, const double *inv,
                             const double *cterm, const double *A, const double *logA, int64_t lane_slots, double *E,
                             double *alpha, double *beta, double *gamma, double *xi /* may be NULL */,
                             double *utt_out, const float *feat_t /* may be NULL */,
                             const double *frame_sums /* may be NULL */, void *stream);
/* single-utterance pieces on caller-supplied lattices (the reference's per-method API: forward(E),
 * backward(E, scale), compute_gamma(alpha, beta), compute_xi(alpha, beta, E)); op: 0 emission,
 * 1 forward (scale -> scalar[0]), 2 backward (scale <- scalar[0]), 3 gamma, 4 xi */
int sapr_custom_piece(int32_t op, const float *x, int32_t T, int32_t D, int32_t S, const double *means,
                      const double *inv, const double *cterm, const double *A, const double *logA, double *E,
                      double *alpha, double *beta, double *gamma, double *xi, double *scalar, void *stream);
/* evaluation-order-faithful emission rows (custom_hmm.py:168-172 as numpy/OpenBLAS evaluate it: two
 * fused-multiply-add chains over the contraction index and numpy's pair-wise row sum of the (T,T) Gram matrix),
 * bit-identical to the reference's compute_emission_matrix on the golden build.  n_rows > 0: the first n_rows
 * frames of every utterance against every model, E[n_utts][W][n_rows][S]; n_rows == 0 (W == 1): all frames,
 * E[total_frames][S]. */
int sapr_custom_emission_exact(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t W, int32_t D,
                               int32_t S, int32_t n_rows, int32_t max_T /* longest utterance; used when n_rows == 0 */,
                               const double *means, const double *inv, const double *cterm, double *E, void *stream);
/* HMM.decode (custom_hmm.py:462-514) for every (utterance, model) and, optionally, Decoder.decode_sequence's
 * arg-max over the models (decoder.py:35-49): e_rows is workspace for n_utts*W*Tq*S doubles; every utterance
 * must hold >= Tq frames; best_word / best_score / best_path[n_utts][Tq] may all be NULL */
int sapr_custom_decode(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t W, int32_t D,
                       int32_t S, int32_t num_states, int32_t Tq, const double *means, const double *inv,
                       const double *cterm, const double *A, const double *logA, double *e_rows, double *scores,
                       int32_t *paths, int32_t *best_word, double *best_score, int32_t *best_path, void *stream);
int sapr_custom_update_b_workspace_bytes(int64_t n_utts, int32_t W, int32_t D, int32_t S, size_t *bytes);
int sapr_custom_update_b(const float *feats, const int64_t *offsets, const int32_t *utt_model, int64_t n_utts,
                         int32_t W, int32_t D, int32_t S, const double *gamma, int64_t lane_slots,
                         double *means_out, double *occ_out, double *covs_out, void *workspace,
                         size_t workspace_bytes, void *stream);
/* the same two passes split so that a sharded run can sum across ranks in between (custom_hmm.py:366-400 is
 * two-pass: covariances are taken about the NEW means): unnormalised sum_x[W][S][D] + occ[W][S], then
 * unnormalised scatter[W][S][D][D] about `means`; sapr_custom_normalise divides by occ where occ > 0 */
int sapr_custom_update_b_sums(const float *feats, const int64_t *offsets, const int32_t *utt_model, int64_t n_utts,
                              int32_t W, int32_t D, int32_t S, const double *gamma, int64_t lane_slots,
                              double *sum_x_out, double *occ_out, void *workspace, size_t workspace_bytes,
                              void *stream);
int sapr_custom_update_b_scatter(const float *feats, const int64_t *offsets, const int32_t *utt_model,
                                 int64_t n_utts, int32_t W, int32_t D, int32_t S, const double *gamma,
                                 int64_t lane_slots, const double *means, double *scatter_out, void *workspace,
                                 size_t workspace_bytes, void *stream);
/* both passes as one (one model, D = 13, S <= 16): out[16][112] = posterior-weighted moments about `center`[D] —
 * row s: columns 0..90 the upper triangle of sum g x'x'^T, 91..103 sum g x', 104 sum g (x' = x - center); after the
 * cross-rank sum mean = center + s1/occ, cov = S2/occ - (s1/occ)(s1/occ)^T: custom_hmm.py:366-400's values from one
 * read of the data on the float64 matrix cores */
int sapr_custom_update_b_moments(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t D, int32_t S,
                                 const double *gamma, int64_t lane_slots, const double *center, double *out,
                                 void *workspace, size_t workspace_bytes, void *stream);
int sapr_custom_normalise(double *x, const double *occ, int64_t n_states, int32_t per, void *stream);
/* out[K] = sum over rows of part[n_rows][K], rows added one after another in row order — the reference's
 * accumulation over sequences (custom_hmm.py:434-439) applied to sapr_custom_estep's utt_out, on the device */
int sapr_custom_fold_rows(const double *part, int64_t n_rows, int64_t K, double *out, void *stream);
int sapr_custom_global_workspace_bytes(int64_t n_utts, int64_t total_frames, int32_t D, size_t *bytes);
int sapr_custom_global_sum(const float *feats, const int64_t *offsets, int64_t n_utts, int32_t D,
                           double *sum_out, void *workspace, size_t workspace_bytes, void *stream);
int sapr_custom_global_cov(const float *feats, int64_t total_frames, int32_t D, const double *mean,
                           double *cov_out, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * MFCC front-end.  Replaces librosa.feature.mfcc(y, sr, n_mfcc=13, win_length, hop_length,
 * window="hamming", center=True) as called at mfcc_extract.py:15-23 (librosa defaults:
 * n_fft 2048, 128 Slaney mels, power 2, power_to_db(top_db=80), DCT-II ortho), batched over
 * utterances, plus BASELINE.json's north-star options (pre-emphasis, delta / delta-delta).
 *
 *   plan   host-side tables (Hamming window, FFT twiddles, banded mel filterbank as MFMA
 *          fragments, DCT rows, Savitzky-Golay taps) uploaded once; n_fft is 512 or 2048.
 *          max_frames > 0: FUSED mode — the log-mel matrix of one utterance (at most max_frames
 *          frames) lives in LDS, so the utterance-global top_db maximum costs no second HBM pass;
 *          max_frames == 0 (or a fused layout that does not fit 160 KiB of LDS, or one that fits only one workgroup
 *          per CU where the layout without the log-mel matrix fits two — the reference preset: ask
 *          sapr_mfcc_plan_info / sapr_mfcc_workspace_bytes, do not assume): TWO-PASS mode —
 *          log-mel rows go through a caller-supplied HBM workspace and a second small kernel does
 *          clip / DCT / deltas; utterances of any length.
 *   batch  pcm[total_samples] float32 (librosa.load's mono float32, mfcc_extract.py:12),
 *          sample_offsets[n_utts+1], frame_offsets[n_utts+1] with
 *          frames(u) = 1 + n_samples(u) / hop  (center=True);
 *          out[total_frames][d_out] float32 frame-major, d_out = n_mfcc * (deltas ? 3 : 1)
 *          — the layout sapr_viterbi_diag_scores consumes (transpose of the reference's (13,T)).
 *          HARD PRECONDITION: total_frames == frame_offsets[n_utts] (the value on the device).  `out`, the
 *          workspace and the launch are sized from total_frames; the offsets are only read on the device, so
 *          the call cannot return an error for a mismatch.  The 512-point wave-private core checks it on the
 *          device: offsets that describe MORE frames than total_frames make it write nothing but NaN into all
 *          of `out` (tests/test_capi_errors_gpu.py) instead of running past the buffers.
 * ---------------------------------------------------------------------------------- */
int sapr_mfcc_plan_create(double sr, int32_t n_fft, int32_t win_length, int32_t hop,
                          int32_t n_mels, int32_t n_mfcc, double fmin, double fmax /* <=0: sr/2 */,
                          double top_db, double preemph /* 0 = off */, int32_t deltas,
                          int32_t max_frames, void **plan_out);
int sapr_mfcc_plan_destroy(void *plan);
int sapr_mfcc_plan_info(const void *plan, int32_t *d_out, int32_t *max_frames, int64_t *lds_bytes,
                        int32_t *mel_ksteps);
int sapr_mfcc_workspace_bytes(const void *plan, int64_t total_frames, int64_t n_utts, size_t *bytes); /* 0 if fused */
int sapr_mfcc_batch(const void *plan, const float *pcm, const int64_t *sample_offsets,
                    const int64_t *frame_offsets, int64_t n_utts, int64_t total_frames, float *out,
                    int32_t grid_blocks /* <=0: auto */, void *workspace /* may be NULL if fused */,
                    size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Sample-rate conversion ahead of the MFCC chain: librosa.load(path) resamples every file to
 * 22 050 Hz (mfcc_extract.py:12).  Polyphase FIR with scipy.signal.resample_poly's definition; the
 * low-pass taps (already scaled by `up` and left-padded as scipy does) and n_pre_remove come from the
 * host (sapr_amd/mfcc_extract.py: resample_design).  out_offsets[u+1]-out_offsets[u] must equal
 * ceil(n_in(u) * up / down).  float32 in/out, float64 accumulation.
 * ---------------------------------------------------------------------------------- */
int sapr_resample_poly(const float *x, const int64_t *in_offsets, const int64_t *out_offsets, int64_t n_utts,
                       int64_t max_out, int32_t up, int32_t down, const float *taps, int32_t n_taps,
                       int32_t n_pre_remove, float *y, void *stream);

/* 16-bit PCM (WAV sample format) -> float32 in [-1, 1) on the device: x / 32768, as the host WAV reader of
 * sapr_amd/mfcc_extract.py does, so that host-resident audio crosses PCIe as 2 bytes per sample */
int sapr_pcm16_to_f32(const int16_t *pcm16, int64_t n_samples, float *out, void *stream);

/* diagnostic build of sapr_mfcc_batch (BENCH-style plans only): stamps[grid_blocks][4][12] receives
 * per-wavefront, per-phase s_memtime sums.  Read the shares, not the run time. */
int sapr_mfcc_batch_stamped(const void *plan, const float *pcm, const int64_t *sample_offsets,
                            const int64_t *frame_offsets, int64_t n_utts, float *out,
                            int32_t grid_blocks, uint64_t *stamps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SAPR_HIP_H */
