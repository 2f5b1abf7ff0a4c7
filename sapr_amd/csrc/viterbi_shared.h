// Shared declarations of the Viterbi translation units (viterbi.hip: model packing, selection, back-trace and the
// C ABI; viterbi_exact_<D>_<S>.hip: the exact lattice kernels of one (D, S) shape; viterbi_bound.hip: the pruned
// decoder's bounding passes).  One translation unit per kernel family keeps a cold build at a few minutes on a
// handful of cores instead of one nine-minute compile.
#pragma once

#include "emission.h"

namespace sapr {

constexpr int kBlock = 256;  // 4 wavefronts per workgroup
constexpr int kXcd = 8;
#ifndef SAPR_EXACT_NF  // dev switches: frames per parameter walk of the pruned decoder's exact pass (2: unfused)
#define SAPR_EXACT_NF 4    // 13 dims
#endif
#ifndef SAPR_EXACT_NF39
#define SAPR_EXACT_NF39 4  // 39 dims
#endif

__host__ __device__ inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------
// grid decode: block id -> (utterance tile, word model), XCD-aware (blocks b and b+8 share
// an XCD under round-robin dispatch; a different placement only changes speed).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_block(int W, int64_t n_tiles, int64_t &tile, int &w) {
  const int64_t id = blockIdx.x;
  const int xcd = static_cast<int>(id % kXcd);
  const int64_t k = id / kXcd;
  tile = (k / W) * kXcd + xcd;
  w = static_cast<int>(k % W);
  (void)n_tiles;
}

struct ScoreArgs {
  const float *feats;
  const int64_t *offsets;
  const int32_t *order;
  int64_t n_utts, n_tiles, n_slots;
  int32_t max_T, W;
  const double4 *prm;
  const double *gconst, *log_start, *log_trans;
  void *bp;
  double *scores;
  int32_t *last_state;
  hipStream_t stream;
  const int32_t *cand_utt = nullptr;  // pruned decoder, pass C: per-word utterance lists ...
  const int32_t *cand_cnt = nullptr;  // ... and their lengths
  int32_t seq_all = 0;                // SAPR_SUM_SEQ: no pair-wise exception for one-frame utterances
};

struct PrunedLayout {
  size_t bp, cand_utt, cand_slot, ascore, aeps, scores, last, cnt, total;
};
__host__ inline PrunedLayout pruned_layout(int64_t n_utts, int W, int max_T) {
  const int64_t n_slots = round_up(n_utts > 0 ? n_utts : 1, kBlock);
  const size_t nw = static_cast<size_t>(n_utts > 0 ? n_utts : 1) * W;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  PrunedLayout L;
  size_t o = 0;
  L.bp = o;
  o += al(static_cast<size_t>(W) * static_cast<size_t>(max_T > 0 ? max_T : 1) * n_slots * sizeof(uint32_t));
  L.cand_utt = o;
  o += al(static_cast<size_t>(W) * n_slots * sizeof(int32_t));
  L.cand_slot = o;
  o += al(nw * sizeof(int32_t));
  L.ascore = o;
  o += al(nw * sizeof(double));
  L.aeps = o;
  o += al(nw * sizeof(double));
  L.scores = o;
  o += al(nw * sizeof(double));
  L.last = o;
  o += al(nw * sizeof(int32_t));
  L.cnt = o;
  o += al(static_cast<size_t>(W) * sizeof(int32_t));
  L.total = o;
  return L;
}

inline size_t workspace_bytes(int64_t n_utts, int W, int S, int max_T, int topology) {
  const int64_t n_slots = round_up(n_utts > 0 ? n_utts : 1, kBlock);
  const size_t per = topology == SAPR_TOPO_BIDIAG ? sizeof(uint32_t) : static_cast<size_t>(S);
  return static_cast<size_t>(W) * static_cast<size_t>(max_T > 0 ? max_T : 1) * n_slots * per;
}

// per-shape launchers, one translation unit each (D features, S kernel states); the 39-dimensional shapes, whose
// unrolled bodies compile for minutes, are cut once more by tie rule and summation order
int launch_scores_13_10(const ScoreArgs &a, int topology, int tie, int sum_order, int fast);
int launch_scores_13_18(const ScoreArgs &a, int topology, int tie, int sum_order, int fast);
int launch_scores_39_18_t1s1(const ScoreArgs &a, int topology, int fast);
int launch_scores_39_18_t1s0(const ScoreArgs &a, int topology, int fast);
int launch_scores_39_18_t0s1(const ScoreArgs &a, int topology, int fast);
int launch_scores_39_18_t0s0(const ScoreArgs &a, int topology, int fast);
inline int launch_scores_39_18(const ScoreArgs &a, int topology, int tie, int sum_order, int fast) {
  if (tie == SAPR_TIE_HIGH) return sum_order ? launch_scores_39_18_t1s1(a, topology, fast) : launch_scores_39_18_t1s0(a, topology, fast);
  return sum_order ? launch_scores_39_18_t0s1(a, topology, fast) : launch_scores_39_18_t0s0(a, topology, fast);
}
int launch_scores_39_10_t1s1(const ScoreArgs &a, int topology, int fast);
int launch_scores_39_10_t1s0(const ScoreArgs &a, int topology, int fast);
int launch_scores_39_10_t0s1(const ScoreArgs &a, int topology, int fast);
int launch_scores_39_10_t0s0(const ScoreArgs &a, int topology, int fast);
inline int launch_scores_39_10(const ScoreArgs &a, int topology, int tie, int sum_order, int fast) {
  if (tie == SAPR_TIE_HIGH) return sum_order ? launch_scores_39_10_t1s1(a, topology, fast) : launch_scores_39_10_t1s0(a, topology, fast);
  return sum_order ? launch_scores_39_10_t0s1(a, topology, fast) : launch_scores_39_10_t0s0(a, topology, fast);
}
int launch_approx_13_10(const ScoreArgs &a, const emission::PackView &pv, double *ascore, double *aeps, int pack_flags);
int launch_approx_13_18(const ScoreArgs &a, const emission::PackView &pv, double *ascore, double *aeps, int pack_flags);
int launch_approx_39_10(const ScoreArgs &a, const emission::PackView &pv, double *ascore, double *aeps, int pack_flags);
int launch_approx_39_18(const ScoreArgs &a, const emission::PackView &pv, double *ascore, double *aeps, int pack_flags);

}  // namespace sapr
