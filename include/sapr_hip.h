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

#define SAPR_ABI_VERSION 1

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

#define SAPR_ERR_ARG (-1)
#define SAPR_ERR_UNSUPPORTED (-2)
#define SAPR_ERR_WORKSPACE (-3)

int sapr_abi_version(void);
const char *sapr_last_error(void);
/* number of CUs / wave size / gcnArchName of device `dev`; arch buffer may be NULL */
int sapr_device_info(int dev, int *cu_count, int *wave_size, char *arch, size_t arch_len);

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
 * into one device blob {mean, var, RN(1/var)} ... that the kernels read with scalar loads.
 * *fast_div_ok = 1 when every parameter lies in the domain where the FMA-based exactly-rounded
 * division of viterbi.hip is proven equal to IEEE division (pass it on as `fast_div`; 0 selects
 * the IEEE-division instantiation — same bits, slower).  Synchronises `stream`. */
int sapr_diag_pack_bytes(int32_t W, int32_t S, int32_t D, size_t *bytes);
int sapr_diag_pack(const double *means, const double *vars, const double *gconst,
                   const double *log_start, const double *log_trans, int32_t W, int32_t S, int32_t D,
                   void *pack, size_t pack_bytes, int32_t *fast_div_ok /* host */, void *stream);

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

/* ------------------------------------------------------------------------------------
 * MFCC front-end.  Replaces librosa.feature.mfcc(y, sr, n_mfcc=13, win_length, hop_length,
 * window="hamming", center=True) as called at mfcc_extract.py:15-23 (librosa defaults:
 * n_fft 2048, 128 Slaney mels, power 2, power_to_db(top_db=80), DCT-II ortho), batched over
 * utterances, plus BASELINE.json's north-star options (pre-emphasis, delta / delta-delta).
 *
 *   plan   host-side tables (Hamming window, FFT twiddles, banded mel filterbank as MFMA
 *          fragments, DCT rows, Savitzky-Golay taps) uploaded once; n_fft is 512 or 2048;
 *          utterances may have at most max_frames frames (the log-mel matrix of one utterance
 *          lives in LDS so the utterance-global top_db maximum costs no second HBM pass)
 *   batch  pcm[total_samples] float32 (librosa.load's mono float32, mfcc_extract.py:12),
 *          sample_offsets[n_utts+1], frame_offsets[n_utts+1] with
 *          frames(u) = 1 + n_samples(u) / hop  (center=True);
 *          out[total_frames][d_out] float32 frame-major, d_out = n_mfcc * (deltas ? 3 : 1)
 *          — the layout sapr_viterbi_diag_scores consumes (transpose of the reference's (13,T)).
 * ---------------------------------------------------------------------------------- */
int sapr_mfcc_plan_create(double sr, int32_t n_fft, int32_t win_length, int32_t hop,
                          int32_t n_mels, int32_t n_mfcc, double fmin, double fmax /* <=0: sr/2 */,
                          double top_db, double preemph /* 0 = off */, int32_t deltas,
                          int32_t max_frames, void **plan_out);
int sapr_mfcc_plan_destroy(void *plan);
int sapr_mfcc_plan_info(const void *plan, int32_t *d_out, int32_t *max_frames, int64_t *lds_bytes,
                        int32_t *mel_ksteps);
int sapr_mfcc_batch(const void *plan, const float *pcm, const int64_t *sample_offsets,
                    const int64_t *frame_offsets, int64_t n_utts, float *out,
                    int32_t grid_blocks /* <=0: auto */, void *stream);

/* diagnostic build of sapr_mfcc_batch (BENCH-style plans only): stamps[grid_blocks][4][12] receives
 * per-wavefront, per-phase s_memtime sums.  Read the shares, not the run time. */
int sapr_mfcc_batch_stamped(const void *plan, const float *pcm, const int64_t *sample_offsets,
                            const int64_t *frame_offsets, int64_t n_utts, float *out,
                            int32_t grid_blocks, uint64_t *stamps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SAPR_HIP_H */
