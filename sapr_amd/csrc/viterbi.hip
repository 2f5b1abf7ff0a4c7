// Batched Viterbi decode for diagonal-Gaussian HMMs on gfx950 (MI355X).
//
// Replaces GaussianHMM.decode(X) as the reference calls it for every word model
// (decoder.py:42-43): log-density = hmmlearn stats.py _log_multivariate_normal_density_diag,
// lattice / back-trace = hmmlearn _hmmc.cpp viterbi.  See oracle/hmmlearn_oracle.py for the
// CPU restatement these kernels are tested against (bit-identical scores and paths).
//
// Mapping (MI355X-first, not a translation of the CPU loops):
//   * one LANE walks one utterance's trellis against one word model; a wavefront is 64
//     utterances of similar length (host passes a length-sorted `order`), so there are no
//     cross-lane dependencies in the time loop and every lane does useful work;
//   * the word model is wavefront-uniform, so its parameters (mean, variance, constants,
//     log-transitions) are fetched with SCALAR loads and feed the fp64 VALU as SGPR operands —
//     no LDS or vector-memory traffic for parameters at all;
//   * the per-frame state is registers only: S lattice values (fp64), D feature values;
//   * back-pointers: for the left-to-right (bidiagonal) topology one bit per state, packed
//     into one 32-bit word per (model, frame, utterance) and stored lane-contiguous
//     (bp[w][t][slot]) → one coalesced 256-B store per wavefront per frame;
//   * the 1-D grid is decoded XCD-aware: the W blocks that read the SAME 256 utterances
//     get consecutive per-XCD slots, so the features cross HBM once and are re-read from
//     that XCD's L2 by the other W-1 models.
//
// The kernel is fp64-VALU bound (W*S*D IEEE divisions per frame), not HBM bound; DESIGN.md
// §5 carries the arithmetic.  Compiled with -ffp-contract=off: every add/mul/div is the
// individually rounded IEEE operation numpy performs.
//
// Division.  q = RN(a / var) is obtained without the 11-instruction IEEE division expansion
// (v_div_scale x2, v_rcp_f64, 4 FMA, mul, FMA, v_div_fmas, v_div_fixup): the reciprocal is
// precomputed per (model, state, dim) by sapr_diag_pack as a two-word value yh = RN(1/var),
// yl = RN(1/var - yh), and
//     t = RN(a*yl);  q0 = fma(a, yh, t);          (q0 faithful: one of the two neighbours of a/var)
//     r = fma(-var, q0, a);  q = fma(r, yh, q0)   (Markstein's correction: r exact, q == RN(a/var))
// whenever nothing under/overflows (emission.h quad_term; scripts/verify/fastdiv_check.c).
// sapr_diag_pack checks the model against a conservative domain (variances in [1e-30, 1e30],
// |mean| in {0} U [1e-30, 1e30]; float32 features then keep every intermediate normal) and
// reports it; outside it the exact-division instantiation runs.
#include "emission.h"

namespace sapr {
namespace {

using namespace emission;

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

// ---------------------------------------------------------------------------------------
// pass 1, bidiagonal topology
// ---------------------------------------------------------------------------------------
// CAND = false: every utterance against every word model (block -> (utterance tile, word), XCD-aware).
// CAND = true (pruned decoder, second pass): word w's list cand_utt[w][0 .. cand_cnt[w]) only; block ->
// (word, tile of that list), blocks past the end of a list exit at once; slot = position in the list.
template <int D, int S, bool TIE_HIGH, bool SEQ, bool FASTDIV, bool CAND>
__global__ __launch_bounds__(kBlock) void viterbi_bidiag_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ order, int64_t n_utts, int64_t n_tiles, int64_t n_slots,
    int32_t max_T, int32_t W, const double4 *__restrict__ prm_all,
    const double *__restrict__ gconst, const double *__restrict__ log_start,
    const double *__restrict__ log_trans, uint32_t *__restrict__ bp, double *__restrict__ scores,
    int32_t *__restrict__ last_state, const int32_t *__restrict__ cand_utt,
    const int32_t *__restrict__ cand_cnt) {
  static_assert(S <= 32, "one back-pointer bit per state in a 32-bit word");
  constexpr bool kFuseColumn = D >= 39;
  int64_t tile;
  int w;
  int64_t n_live = n_utts;
  if constexpr (CAND) {
    w = static_cast<int>(blockIdx.x / n_tiles);
    tile = blockIdx.x - static_cast<int64_t>(w) * n_tiles;
    n_live = cand_cnt[w];
    if (tile * kBlock >= n_live) return;
  } else {
    decode_block(W, n_tiles, tile, w);
    if (tile >= n_tiles) return;  // grid padding (whole block leaves together)
  }

  const int64_t slot = tile * kBlock + threadIdx.x;
  const bool live = slot < n_live;
  int64_t u = 0;
  if constexpr (CAND) {
    if (live) u = cand_utt[static_cast<int64_t>(w) * n_slots + slot];
  } else {
    if (live) u = order ? static_cast<int64_t>(order[slot]) : slot;
  }
  const int64_t beg = live ? offsets[u] : 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
  // SUM_TVIEW: a one-frame utterance is the exception (numpy sees a C-contiguous (1,D) view and
  // sums pair-wise); such lanes are finished in a separate, rarely executed pre-pass so that the
  // main loop's summation order is a compile-time constant
  const bool single = SEQ && T == 1;
  const int Tw = wave_max_i32(single ? 0 : T);

  // wavefront-uniform model pointers → scalar loads
  const double4 *__restrict__ prm = prm_all + static_cast<int64_t>(w) * S * D;
  const double *__restrict__ gc = gconst + static_cast<int64_t>(w) * S;
  const double *__restrict__ ls = log_start + static_cast<int64_t>(w) * S;
  const double *__restrict__ lt = log_trans + static_cast<int64_t>(w) * S * S;

  const float *__restrict__ xp = feats + beg * D;
  uint32_t *__restrict__ bpw = bp + (static_cast<int64_t>(w) * max_T) * n_slots + slot;

  double delta[S];
  double x[D];
#pragma unroll
  for (int s = 0; s < S; ++s) delta[s] = ls[s];

  // t = 0 shares the loop body (one copy of the S*D-division emission code in the kernel):
  // delta starts as log_start, frame 0 adds b without a transition and writes no
  // back-pointer word.
  if (__any(single)) {
    if (single) {
      load_frame<D>(xp, x);
      double b[S];
      frame_log_densities<D, S, FASTDIV, false>(x, prm, gc, b);
#pragma unroll
      for (int s = 0; s < S; ++s) delta[s] += b[s];
    }
  }

  // column update of one frame from its log-densities (non-fused form): descending j so delta[j-1] is still the
  // value of the previous frame when state j reads it
  auto column = [&](int t, const double (&b)[S]) {
    const bool first = (t == 0);
    uint32_t bits = 0;
#pragma unroll
    for (int j = S - 1; j >= 1; --j) {
      // frame 0 has no transition: the (wavefront-uniform, scalar) selects make the predecessor
      // candidate -inf and the self-loop weight 0, so max() returns delta[j] itself
      const double cp = delta[j - 1] + (first ? neg_inf() : lt[(j - 1) * S + j]);  // from j-1
      const double cs = delta[j] + (first ? 0.0 : lt[j * S + j]);                  // self loop
      // hmmlearn back-trace: max over predecessors of (value, index); among the two finite
      // candidates index j-1 < j.
      const bool from_prev = TIE_HIGH ? (cp > cs) : (cp >= cs);
      delta[j] = fmax(cp, cs) + b[j];  // == from_prev ? cp : cs (equal candidates are the same value)
      bits |= static_cast<uint32_t>(from_prev) << j;
    }
    delta[0] = (delta[0] + (first ? 0.0 : lt[0])) + b[0];
    if (!first) bpw[static_cast<int64_t>(t) * n_slots] = bits;
  };

  if constexpr (FASTDIV && CAND) {
    // two frames per walk over the parameters (emission.h EmitLoop2): the exact pass over the pruned decoder's
    // survivors runs at ~1.5 wavefronts per SIMD, where the scalar loads' latency shows (0.89 -> 0.75 ms per
    // 100 000 utterances at D = 13); on a full grid (CAND = false) the single-frame walk's lower register
    // count wins.  39-dimensional frames stay float32 in registers and are promoted inside the chain.
    using XT = std::conditional_t<(D >= 39), float, double>;
    XT xa[D], xb[D];
    auto load2 = [&](const float *p, XT (&dst)[D]) {
      if constexpr (D >= 39)
        load_frame_f32<D>(p, dst);
      else
        load_frame<D>(p, dst);
    };
    if constexpr ((D < 39 ? SAPR_EXACT_NF : SAPR_EXACT_NF39) > 2) {
      // 13 dimensions: FOUR frames per walk (48 fp64 instructions per s_load pair), with the four column updates
      // fused into the emission loop: for each state, in frame order, on the one lattice column, every frame
      // carrying its own predecessor value (the value delta[j-1] had when THAT frame's turn came) — the same
      // operations on the same operands as four column() calls
      constexpr int NF = D < 39 ? SAPR_EXACT_NF : SAPR_EXACT_NF39;
      for (int t = 0; t < Tw; t += NF) {
        if (t < T && !single) {
          XT xs[NF][D];
#pragma unroll
          for (int f = 0; f < NF; ++f) load2(xp + static_cast<int64_t>(t + f < T ? t + f : T - 1) * D, xs[f]);
          uint32_t bits[NF];
          double carry[NF];
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            bits[f] = 0u;
            carry[f] = 0.0;
          }
          frame_log_densities_n_each<D, S, SEQ, NF>(xs, prm, gc, [&](auto jc, const double (&b)[NF]) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              if (t + f < T) {
                const bool first = (t + f == 0);
                const double old = delta[j];
                if constexpr (j == 0) {
                  delta[0] = (first ? old : (old + lt[0])) + b[f];
                } else {
                  const double cp = carry[f] + lt[(j - 1) * S + j];  // from j-1
                  const double cs = old + lt[j * S + j];             // self loop
                  const bool from_prev = TIE_HIGH ? (cp > cs) : (cp >= cs);
                  delta[j] = (first ? old : (from_prev ? cp : cs)) + b[f];
                  bits[f] |= static_cast<uint32_t>(from_prev) << j;
                }
                carry[f] = old;
              }
            }
          });
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (t + f < T && t + f > 0) bpw[static_cast<int64_t>(t + f) * n_slots] = bits[f];
        }
      }
    } else {
    for (int t = 0; t < Tw; t += 2) {
      if (t < T && !single) {
        const bool two = (t + 1) < T;
        load2(xp + static_cast<int64_t>(t) * D, xa);
        load2(xp + static_cast<int64_t>(two ? t + 1 : t) * D, xb);
        double ba[S], bb[S];
        frame_log_densities2<D, S, SEQ>(xa, xb, prm, gc, ba, bb);
        column(t, ba);
        if (two) column(t + 1, bb);
      }
    }
    }
  } else {
  for (int t = 0; t < Tw; ++t) {
    if (t < T && !single) {
      load_frame<D>(xp + static_cast<int64_t>(t) * D, x);
      const bool first = (t == 0);
      if constexpr (kFuseColumn) {
        uint32_t bits = 0;
        // 39-dimensional features: the column update consumes each state's log-density as the emission
        // loop produces it (ascending j, the previous column's delta[j-1] carried in a register), so no
        // b[S] array lives next to x[D] — 36 VGPRs at S = 18, one more wavefront per SIMD
        double carry = 0.0;  // delta[j-1] of frame t-1
        frame_log_densities_each<D, S, FASTDIV, SEQ>(x, prm, gc, [&](auto jc, double bj) {
          constexpr int j = decltype(jc)::value;
          const double old = delta[j];
          if constexpr (j == 0) {
            const double m = first ? old : (old + lt[0]);
            delta[0] = m + bj;
          } else {
            const double cp = carry + lt[(j - 1) * S + j];  // from j-1
            const double cs = old + lt[j * S + j];          // self loop
            const bool from_prev = TIE_HIGH ? (cp > cs) : (cp >= cs);
            const double m = first ? old : (from_prev ? cp : cs);
            delta[j] = m + bj;
            bits |= static_cast<uint32_t>(from_prev) << j;
          }
          carry = old;
        });
        if (!first) bpw[static_cast<int64_t>(t) * n_slots] = bits;
      } else {
        double b[S];
        frame_log_densities<D, S, FASTDIV, SEQ>(x, prm, gc, b);
        column(t, b);
      }
    }
  }
  }

  if (live) {
    // final state: std::max_element → first maximum
    double best = delta[0];
    int arg = 0;
#pragma unroll
    for (int s = 1; s < S; ++s) {
      if (delta[s] > best) {
        best = delta[s];
        arg = s;
      }
    }
    scores[u * W + w] = T > 0 ? best : neg_inf();
    last_state[u * W + w] = arg;
  }
}

// ---------------------------------------------------------------------------------------
// pass 1, dense topology (any transmat): one byte back-pointer per state
// ---------------------------------------------------------------------------------------
template <int D, int S, bool TIE_HIGH, bool SEQ, bool FASTDIV>
__global__ __launch_bounds__(kBlock) void viterbi_dense_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ order, int64_t n_utts, int64_t n_tiles, int64_t n_slots,
    int32_t max_T, int32_t W, const double4 *__restrict__ prm_all,
    const double *__restrict__ gconst, const double *__restrict__ log_start,
    const double *__restrict__ log_trans, uint8_t *__restrict__ bp, double *__restrict__ scores,
    int32_t *__restrict__ last_state) {
  int64_t tile;
  int w;
  decode_block(W, n_tiles, tile, w);
  if (tile >= n_tiles) return;

  const int64_t slot = tile * kBlock + threadIdx.x;
  const bool live = slot < n_utts;
  const int64_t u = live ? (order ? static_cast<int64_t>(order[slot]) : slot) : 0;
  const int64_t beg = live ? offsets[u] : 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
  // SUM_TVIEW: a one-frame utterance is the exception (numpy sees a C-contiguous (1,D) view and
  // sums pair-wise); such lanes are finished in a separate, rarely executed pre-pass so that the
  // main loop's summation order is a compile-time constant
  const bool single = SEQ && T == 1;
  const int Tw = wave_max_i32(single ? 0 : T);

  const double4 *__restrict__ prm = prm_all + static_cast<int64_t>(w) * S * D;
  const double *__restrict__ gc = gconst + static_cast<int64_t>(w) * S;
  const double *__restrict__ ls = log_start + static_cast<int64_t>(w) * S;
  const double *__restrict__ lt = log_trans + static_cast<int64_t>(w) * S * S;

  const float *__restrict__ xp = feats + beg * D;
  uint8_t *__restrict__ bpw = bp + (static_cast<int64_t>(w) * max_T) * S * n_slots + slot;

  double delta[S], prev[S];
  double x[D];
#pragma unroll
  for (int s = 0; s < S; ++s) delta[s] = ls[s];

  if (__any(single)) {
    if (single) {
      load_frame<D>(xp, x);
      double b[S];
      frame_log_densities<D, S, FASTDIV, false>(x, prm, gc, b);
#pragma unroll
      for (int s = 0; s < S; ++s) delta[s] += b[s];
    }
  }

  for (int t = 0; t < Tw; ++t) {
    if (t < T && !single) {
      load_frame<D>(xp + static_cast<int64_t>(t) * D, x);
      const bool first = (t == 0);
      double b[S];
      frame_log_densities<D, S, FASTDIV, SEQ>(x, prm, gc, b);
#pragma unroll
      for (int s = 0; s < S; ++s) prev[s] = delta[s];
#pragma unroll
      for (int j = 0; j < S; ++j) {
        // std::max over (value, index) pairs from (-inf, 0): TIE_HIGH replaces on >=, else on >
        double best = neg_inf();
        int arg = 0;
#pragma unroll
        for (int i = 0; i < S; ++i) {
          const double v = prev[i] + lt[i * S + j];
          const bool take = TIE_HIGH ? (i == 0 ? v > best : v >= best) : (v > best);
          best = take ? v : best;
          arg = take ? i : arg;
        }
        const double m = first ? prev[j] : best;
        delta[j] = m + b[j];
        if (!first) bpw[(static_cast<int64_t>(t) * S + j) * n_slots] = static_cast<uint8_t>(arg);
      }
    }
  }

  if (live) {
    double best = delta[0];
    int arg = 0;
#pragma unroll
    for (int s = 1; s < S; ++s) {
      if (delta[s] > best) {
        best = delta[s];
        arg = s;
      }
    }
    scores[u * W + w] = T > 0 ? best : neg_inf();
    last_state[u * W + w] = arg;
  }
}

// ---------------------------------------------------------------------------------------
// Pruned decoder.  decoder.py:35-49 returns only the best word, its score and its state path, so the
// exact lattice is needed for the words that can still be the arg-max.  Pass A bounds every word's score:
//
//   ascore[u][w]  the Viterbi score with float32 emission sums (3 float32 VALU instructions per (state, dim)
//                 instead of 7 float64 ones, no back-pointers), lattice recursion in float64;
//   aeps[u][w]    a bound on |ascore - exact score|, accumulated alongside.  Error sources, with u32 = 2^-24,
//                 u = 2^-53, q = a state's float32 sum, C = sum_d mean^2/var (per state, precomputed):
//                   float(mean), float(1/var), x - mean, square, 13-term fma chain:  |q - Q| <= 19 u32 Q + 1.1 u32 C
//                   (the mean's rounding enters as 2 |x-mean| |mean| u32 / var <= u32 (Q_d + C_d));
//                   the exact kernel's own float64 rounding of Q: <= 20 u Q;
//                   b = -0.5 (gconst + Q): one rounding each side; the lattice: <= 2 additions per frame on
//                   each side, each within u of the running magnitude.
//                 With M = sum over frames and states of q (per frame in float32, frames in float64):
//                   eps = 2 * [ u32 (10 M + 0.6 T Cmax)
//                   + (8T + 16) u (0.5 M + T (0.5 sum|gconst| + sum|log_trans|) + sum|log_start|) ] + T 1e-14 + 1e-30
//                 (factor 2 = safety; the absolute terms cover float32 underflow inside the model domain
//                 var in [1e-20, 1e20] that sapr_diag_pack checks).  Non-finite arithmetic anywhere makes eps
//                 non-finite, which keeps the word.
//
// Pass B keeps word w of utterance u unless ascore + eps < max_w' (ascore - eps) — then its exact score is
// strictly below another word's and it can be neither the arg-max nor a tie — and builds per-word lists.
// Pass C is the exact kernel (CAND = true) over the lists, pass D the arg-max (first strict maximum in
// model order, among the kept words) and the back-trace.  Same best_word / best_score / path bits as the
// all-vocabulary evaluation; tests/test_viterbi_gpu.py checks the bound itself and the outputs.
// ---------------------------------------------------------------------------------------
template <int D, int S>
__global__ __launch_bounds__(kBlock) void viterbi_approx_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
    int64_t n_utts, int64_t n_tiles, int32_t W, const double *__restrict__ prm32_all,
    const double *__restrict__ hgc_all, const double *__restrict__ log_start,
    const double *__restrict__ log_trans, const double *__restrict__ wconst, double *__restrict__ ascore,
    double *__restrict__ aeps) {
  int64_t tile;
  int w;
  decode_block(W, n_tiles, tile, w);
  if (tile >= n_tiles) return;
  const int64_t slot = tile * kBlock + threadIdx.x;
  const bool live = slot < n_utts;
  const int64_t u = live ? (order ? static_cast<int64_t>(order[slot]) : slot) : 0;
  const int64_t beg = live ? offsets[u] : 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
  const int Tw = wave_max_i32(T);

  const double *__restrict__ prm32 = prm32_all + static_cast<int64_t>(w) * pack_p32(S, D);
  const double *__restrict__ hg = hgc_all + static_cast<int64_t>(w) * S;
  const double *__restrict__ ls = log_start + static_cast<int64_t>(w) * S;
  const double *__restrict__ lt = log_trans + static_cast<int64_t>(w) * S * S;
  const float *__restrict__ xp = feats + beg * D;

  double delta[S];
  float x[D];
  double mag = 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) delta[s] = ls[s];
  for (int t = 0; t < Tw; ++t) {
    if (t < T) {
      load_frame_f32<D>(xp + static_cast<int64_t>(t) * D, x);
      const bool first = (t == 0);
      double carry = 0.0;  // delta[j-1] of frame t-1
      float magf = 0.0f;   // this frame's sum of q over the states (float32: S terms), added to mag once
      frame_quads_f32_each<D, S>(x, prm32, [&](auto jc, float qf) {
        constexpr int j = decltype(jc)::value;
        magf += qf;
        const double bj = __builtin_fma(static_cast<double>(qf), -0.5, hg[j]);
        const double old = delta[j];
        // frame 0 has no transition: (wavefront-uniform, scalar) selects make the predecessor candidate -inf and
        // the self-loop weight 0.  fmax may drop a NaN candidate; a NaN can only come from the features or the
        // model, and then mag / the word's constants are NaN too, eps is NaN and the word is kept anyway.
        if constexpr (j == 0) {
          delta[0] = (old + (first ? 0.0 : lt[0])) + bj;
        } else {
          const double cp = carry + (first ? neg_inf() : lt[(j - 1) * S + j]);
          const double cs = old + (first ? 0.0 : lt[j * S + j]);
          delta[j] = fmax(cp, cs) + bj;
        }
        carry = old;
      });
      mag += static_cast<double>(magf);
    }
  }
  if (live) {
    double best = delta[0];
#pragma unroll
    for (int s = 1; s < S; ++s) best = (delta[s] > best || delta[s] != delta[s]) ? delta[s] : best;
    const double *wc = wconst + static_cast<int64_t>(w) * 4;
    const double cmax = wc[0], gcs = wc[1], lts = wc[2], lss = wc[3];
    constexpr double u32 = 5.9604644775390625e-08, u64 = 1.1102230246251565e-16;
    const double Td = static_cast<double>(T);
    const double e32 = u32 * (10.0 * mag + 0.6 * Td * cmax);
    const double e64 = (8.0 * Td + 16.0) * u64 * (0.5 * mag + Td * (0.5 * gcs + lts) + lss);
    ascore[u * W + w] = T > 0 ? best : neg_inf();
    aeps[u * W + w] = T > 0 ? 2.0 * (e32 + e64) + Td * 1e-14 + 1e-30 : 0.0;
  }
}

// ---------------------------------------------------------------------------------------
// pass A on the matrix cores (pack_flags & SAPR_PACK_GEMM_OK; otherwise the kernel above).
//
// The log-density is a quadratic in the features, so against a FIXED centre m (x' = x - m, mu' = mean - m)
//     b_j(x) + sg_j = sum_d (-y_d/2) x'_d^2 + sum_d (y_d mu'_d) x'_d + [ -(c0_j + gconst_j)/2 + sg_j ],
//     c0_j = sum_d y_d mu'_d^2,   y = 1/var,
// is one row of P (16 states x K) times phi(x') = [x'^2 .., 1, 0 .. | x' .., 0 ..] (K = 32 slots for 13
// dims): 16 utterances x 16 states per v_mfma_f32_16x16x32_f16.  Halves have 11 significand bits and a narrow
// exponent range, so every slot carries a power-of-two factor chosen by sapr_diag_pack from the model (x' a_d
// within 2^14 for the linear slots, (x' a_d)^2 within 2^14 for the squared ones over the range mean +- 8 sigma
// of every state, 1024 for the constant), P carries its inverse times 2^g (largest entry in [2^13, 2^14)), and
// the lattice runs in units of 2^-g.  Each operand is a float32 number cut into two halves hi + lo (22 bits);
// the products hi*lo, lo*hi, hi*hi are kept (small ones first), float32 accumulation.
// The self-transition weight rides in the constant slot: with sg_j = lt_jj (0 where the state has no self-loop,
// lt_jj = -inf) and u[j] = delta[j] + sg_j the lattice is u[j] = max(u[j-1] + r_j, u[j]) + (b_j + sg_j),
// r_j = lt_(j-1)j - sg_(j-1) — one weight per state and one add less; a state without a self-loop (the
// reference's entry state, hmmlearn_hmm.py:45-78) has its own candidate turned into a NaN, which v_max_f64
// drops (one v_cndmask_b32 for position 0 of each quarter; a model with such a state elsewhere in the chain is
// bounded by the kernel above).
//
// Lanes: the MFMA result puts states 4q .. 4q+3 (q = lane / 16) of utterance lane % 16 into one lane, so a lane
// owns a quarter of one utterance's lattice column for WC words (fp64, registers); u[4q-1] comes from lane - 16
// (one ds_bpermute pair per word and frame).  More than 16 states: a second row tile (states 16 + 4q ..), whose
// first quarter continues from lane + 48 of the first.  A workgroup is ONE wavefront: 16 utterances x WC words.
//
// Interval.  Let R = sum_k |P_k phi_k| for a (frame, state), in log-density units.  The computed value differs
// from the real-number one by at most cacc * 2^-24 * R + A, cacc = 36 + 68 KC:
//   feature centring, squaring and the float32 rounding of P                          4
//   phi as two truncated halves (2^-20), P as two rounded halves (2^-22), lo*lo (2^-21) 28
//   33 additions per MFMA, each allowed a whole ulp: the two small MFMAs (sums <= 2^-9 R) then the KC leading ones
//   A = 2^-14 2^-g (T max_j sum_k |P_jk 2^g| + sum_t sum_k |slot value|): a half below 2^-14 may be flushed
// With A2 = sum y x'^2, Q = quadratic form >= 0 and |2 y mu' x'| <= y x'^2 / 2 + 2 y mu'^2:  A2 <= 2 Q + 2 c0 and
// R <= 3 |value| + 3 c0 + 2 |gconst| + 4 |sg_j|.  A path meets one state per frame, so its error is at most
// sum_t max_j; each lane keeps sum_t max over ITS four states and the four quarters are added at the end (an
// upper bound of sum_t max_j).  The fp64 terms are as for the VALU kernel.  A slot value beyond the largest half
// (v_cvt_pkrtz saturates silently) or any non-finite arithmetic makes eps non-finite, which keeps the word.
// ---------------------------------------------------------------------------------------
#ifndef SAPR_MFMA_WC  // dev switches: words per wavefront pass / occupancy target of the matrix-core bounding pass
#define SAPR_MFMA_WC 4
#endif
#ifndef SAPR_MFMA_WPE
#define SAPR_MFMA_WPE 2
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// high word -> quiet NaN in the lanes of `mask` (one v_cndmask_b32 on a wavefront-uniform mask)
__device__ __forceinline__ double nan_where(double v, unsigned long long mask) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  unsigned hi = static_cast<unsigned>(b >> 32);
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(hi), "v"(0x7FF80000u), "s"(mask));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | (b & 0xFFFFFFFFull));
}

// v_max_f64 as the hardware does it (IEEE maxNum: a quiet NaN operand is dropped), without the canonicalising
// self-max the compiler puts in front of fmax() for values it cannot prove quiet
__device__ __forceinline__ double max_drop_nan(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

#ifndef SAPR_MFMA_ABL  // dev switch (timing ablations, wrong results): 1 no MFMAs, 2 no lattice update, 4 no piece split
#define SAPR_MFMA_ABL 0
#endif
__device__ __forceinline__ f32x4 mfma_f16(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
  if constexpr (SAPR_MFMA_ABL & 1) {
    f32x4 r = c;
    r[0] += __uint_as_float(a[0] ^ b[1]);
    return r;
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                0);
}

template <int D, int S, int WC>
// waves_per_eu(2): at most 256 registers, which also makes the compiler put the MFMA results in VGPRs (no
// v_accvgpr_read per use)
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(SAPR_MFMA_WPE))) void viterbi_approx_mfma_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
    int64_t n_utts, int32_t W, const uint4 *__restrict__ gfrag, const float *__restrict__ gctr,
    const double *__restrict__ gkw, const double *__restrict__ log_start, const double *__restrict__ log_trans,
    const double *__restrict__ wconst, double *__restrict__ ascore, double *__restrict__ aeps) {
  static_assert(S <= 32, "at most two 16-state row tiles");
  constexpr int G = gemm_groups(D), KC = gemm_kchunks(D), RT = gemm_rtiles(S), NS = 4 * RT, iC = D % 8;
  const int lane = threadIdx.x, col = lane & 15, q = lane >> 4;
  const int n_chunks = (W + WC - 1) / WC;
  const int64_t tile = blockIdx.x / n_chunks;
  const int w0 = static_cast<int>(blockIdx.x - tile * n_chunks) * WC;
  const int nw = W - w0 < WC ? W - w0 : WC;
  const int64_t slot = tile * 16 + col;
  const bool live = slot < n_utts;
  const int64_t u = live ? (order ? static_cast<int64_t>(order[slot]) : slot) : 0;
  const int64_t beg = live ? offsets[u] : 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
  const int Tw = wave_max_i32(T);

  const int Tmin = -wave_max_i32(-T);  // frames every lane of the wavefront has (0 when a lane is idle)
  const int64_t n_floats = offsets[n_utts] * D;

  // which eight slots of phi this lane builds in chunk c: group g = 4c + q; slot value = (x a - ctr a)^(1 or 2)
  constexpr int G8 = 8 * G;
  const double up = gkw[2 * W], down = gkw[2 * W + 1];  // 2^g, 2^-g
  int fbase[KC];
  float ctra[KC][8], fa[KC][8], onev[KC];
  bool sq[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int g = 4 * c + q;
    const int half = g < G ? 0 : (g < 2 * G ? 1 : 2);
    const int gg = half == 2 ? 0 : g - (half == 1 ? G : 0);
    sq[c] = half == 0;
    onev[c] = (half == 0 && gg == D / 8) ? 1024.0f : 0.0f;
    fbase[c] = 8 * gg;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = 8 * gg + i;
      const bool ok = half < 2 && f < D;
      const float a = ok ? gctr[(half == 0 ? G8 : 2 * G8) + f] : 0.0f;
      fa[c][i] = a;
      ctra[c][i] = ok ? gctr[f] * a : 0.0f;  // exact: a is a power of two
    }
  }
  float bigsum = 0.0f;  // sum over frames of this lane's largest |slot value|; NaN once one left the half range
  // this chunk's words: A fragments (zeros past the vocabulary), this lane's states 16 rt + 4 q + i of the lattice
  // column (index rt * 4 + i), per-state weights
  u32x4 afr[WC][RT][KC][2];
  double uu[WC][NS], rr[WC][NS];
  unsigned long long noself0[WC][RT];  // lanes whose state 16 rt + 4 q has no self-loop (wavefront-uniform mask)
  float mag[WC];
#pragma unroll
  for (int wc = 0; wc < WC; ++wc) {
    const bool has = wc < nw;
    const int w = has ? w0 + wc : w0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const uint4 v = gfrag[(((static_cast<int64_t>(w) * RT + rt) * KC + c) * 2 + p) * kWave + lane];
          afr[wc][rt][c][p] = has ? u32x4{v.x, v.y, v.z, v.w} : u32x4{0u, 0u, 0u, 0u};
        }
    const double *ls = log_start + static_cast<int64_t>(w) * S;
    const double *lt = log_trans + static_cast<int64_t>(w) * S * S;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int j = 16 * rt + 4 * q + i;
        uu[wc][rt * 4 + i] = j < S ? ls[j] * up : neg_inf();  // the lattice runs in units of 2^-g
        double sg_prev = 0.0;
        if (j >= 1 && j < S) sg_prev = lt[(j - 1) * S + (j - 1)];
        if (sg_prev == neg_inf()) sg_prev = 0.0;
        rr[wc][rt * 4 + i] = (j >= 1 && j < S) ? (lt[(j - 1) * S + j] - sg_prev) * up : neg_inf();
      }
      const int j0 = 16 * rt + 4 * q;
      noself0[wc][rt] = __ballot(j0 < S && lt[(j0 < S ? j0 : 0) * S + (j0 < S ? j0 : 0)] == neg_inf());
    }
    mag[wc] = 0.0f;
  }

  // a lane reads the eight consecutive floats of its group with two 16-byte loads; the slots past the frame's
  // D values (next frame's data) are multiplied by zero.  Only where that would run past the end of the feature
  // buffer (last frame of the last utterance) does it fall back to clamped single loads.
  float xr[KC][8];
  auto load = [&](int t) {
    const int tt = t < T ? t : T - 1;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) xr[c][i] = 0.0f;
    if (T > 0) {
      const int64_t at = (beg + tt) * D;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float *p = feats + at + fbase[c];
        if (at + fbase[c] + 8 <= n_floats) {
          const FeatQuad v0 = *reinterpret_cast<const FeatQuad *>(p);
          const FeatQuad v1 = *reinterpret_cast<const FeatQuad *>(p + 4);
          xr[c][0] = v0.a, xr[c][1] = v0.b, xr[c][2] = v0.c, xr[c][3] = v0.d;
          xr[c][4] = v1.a, xr[c][5] = v1.b, xr[c][6] = v1.c, xr[c][7] = v1.d;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) xr[c][i] = (fbase[c] + i < D) ? p[i] : 0.0f;
        }
      }
    }
  };

  // UNIFORM: every lane has frame t (no predication of the lattice update)
  auto step = [&](auto first_c, auto uniform_c, int t) {
    constexpr bool first = decltype(first_c)::value, uniform = decltype(uniform_c)::value;
    // B fragments: eight slots of phi(x') per chunk as two halves each (v_cvt_pkrtz: truncation, two values per
    // instruction; the residual of a truncated half is exact in float32)
    u32x4 bh[KC], bl[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      float ph[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xv = __builtin_fmaf(xr[c][i], fa[c][i], -ctra[c][i]);
        const float m = sq[c] ? xv : 1.0f;
        ph[i] = __builtin_fmaf(xv, m, i == iC ? onev[c] : 0.0f);
      }
      const float big = fmaxf(fmaxf(fmaxf(fabsf(ph[0]), fabsf(ph[1])), fmaxf(fabsf(ph[2]), fabsf(ph[3]))),
                              fmaxf(fmaxf(fabsf(ph[4]), fabsf(ph[5])), fmaxf(fabsf(ph[6]), fabsf(ph[7]))));
      bigsum += big > 65504.0f ? __builtin_nanf("") : big;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = ph[2 * e], b = ph[2 * e + 1];
        const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
        if constexpr (!(SAPR_MFMA_ABL & 4)) {
          a -= static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(h2 & 0xFFFFu)));
          b -= static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>(h2 >> 16)));
        }
        bh[c][e] = h2;
        bl[c][e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
      }
    }
    if (t + 1 < Tw) load(t + 1);  // next frame's features: in flight behind this frame's work
    // lattice value of the state just below this lane's first one, per row tile: state 16 rt + 4 q - 1 lives in
    // lane - 16 (same tile, position 3) or, for q == 0 and rt > 0, in lane + 48 of the tile below
    double p3[WC][RT];
    if constexpr (!first) {
#pragma unroll
      for (int wc = 0; wc < WC; ++wc)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          p3[wc][rt] = __shfl_up(uu[wc][rt * 4 + 3], 16);
          if constexpr (RT > 1) {
            if (rt > 0) {
              const double wrap = __shfl(uu[wc][(rt - 1) * 4 + 3], (lane + 48) & 63);
              p3[wc][rt] = q == 0 ? wrap : p3[wc][rt];
            }
          }
        }
    }
    // the WC * RT accumulation chains are independent: issue them interleaved, small products first
    f32x4 acc[WC][RT];
#pragma unroll
    for (int wc = 0; wc < WC; ++wc)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[wc][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
      for (int wc = 0; wc < WC; ++wc)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[wc][rt] = mfma_f16(afr[wc][rt][c][0], bl[c], acc[wc][rt]);
#pragma unroll
      for (int wc = 0; wc < WC; ++wc)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[wc][rt] = mfma_f16(afr[wc][rt][c][1], bh[c], acc[wc][rt]);
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int wc = 0; wc < WC; ++wc)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[wc][rt] = mfma_f16(afr[wc][rt][c][0], bh[c], acc[wc][rt]);
    auto update = [&]() {
#pragma unroll
      for (int wc = 0; wc < WC; ++wc) {
        // (a non-finite slot value makes every row NaN, pads included, and fmaxf(NaN, NaN) is NaN)
        float big = fmaxf(fmaxf(fabsf(acc[wc][0][0]), fabsf(acc[wc][0][1])),
                          fmaxf(fabsf(acc[wc][0][2]), fabsf(acc[wc][0][3])));
#pragma unroll
        for (int rt = 1; rt < RT; ++rt) {
          const f32x4 a = acc[wc][rt];
          big = fmaxf(big, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))));
        }
        mag[wc] += big;
        // descending state order: the predecessor read is still the previous frame's value
#pragma unroll
        for (int rt = RT - 1; rt >= 0; --rt) {
          const f32x4 a = acc[wc][rt];
          if constexpr (first) {
#pragma unroll
            for (int i = 0; i < 4; ++i) uu[wc][rt * 4 + i] += static_cast<double>(a[i]);
          } else {
#pragma unroll
            for (int i = 3; i >= 0; --i) {
              const int k = rt * 4 + i;
              const double pred = (i == 0 ? p3[wc][rt] : uu[wc][k - 1]) + rr[wc][k];
              const double self = i == 0 ? nan_where(uu[wc][k], noself0[wc][rt]) : uu[wc][k];
              uu[wc][k] = max_drop_nan(pred, self) + static_cast<double>(a[i]);
            }
          }
        }
      }
    };
    if constexpr (SAPR_MFMA_ABL & 2) {
#pragma unroll
      for (int wc = 0; wc < WC; ++wc)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) mag[wc] += acc[wc][rt][0] + acc[wc][rt][1] + acc[wc][rt][2] + acc[wc][rt][3];
    } else if constexpr (uniform) {
      update();
    } else {
      if (t < T) update();
    }
  };

  load(0);
  if (Tw > 0) step(std::true_type{}, std::false_type{}, 0);
  int t = 1;
  for (; t < Tmin; ++t) step(std::false_type{}, std::true_type{}, t);
  for (; t < Tw; ++t) step(std::false_type{}, std::false_type{}, t);

  constexpr double u32 = 5.9604644775390625e-08, u64 = 1.1102230246251565e-16;
  constexpr double cacc = 36.0 + 68.0 * KC;
  double phi_sum = static_cast<double>(bigsum);  // over the four k groups: >= sum_t sum_k |slot value| / 8
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) phi_sum += __shfl_xor(phi_sum, off);
#pragma unroll
  for (int wc = 0; wc < WC; ++wc) {
    const bool has = wc < nw;
    const int w = has ? w0 + wc : w0;
    const double *lt = log_trans + static_cast<int64_t>(w) * S * S;
    double best = neg_inf();
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int j = 16 * (k / 4) + 4 * q + (k % 4);
      double sg = j < S ? lt[j * S + j] : 0.0;
      if (sg == neg_inf()) sg = 0.0;
      const double d = uu[wc][k] * down - sg;
      best = (d > best || d != d) ? d : best;
    }
    double m = static_cast<double>(mag[wc]) * down;
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      const double o = __shfl_xor(best, off);
      best = (o > best || o != o) ? o : best;
      m += __shfl_xor(m, off);
    }
    if (has && live && q == 0) {
      const double *wc4 = wconst + static_cast<int64_t>(w) * 4;
      const double lts = wc4[2], lss = wc4[3], Td = static_cast<double>(T);
      const double span = 3.0 * m + Td * gkw[w];
      const double e32 = cacc * u32 * 1.001 * span + 0x1p-14 * 1.01 * (Td * gkw[W + w] + 8.0 * phi_sum) * down;
      const double e64 = (8.0 * Td + 16.0) * u64 * (span + Td * lts + lss);
      ascore[u * W + w] = T > 0 ? best : neg_inf();
      aeps[u * W + w] = T > 0 ? 2.0 * (e32 + e64) + Td * 1e-14 + 1e-30 : 0.0;
    }
  }
}

// pass B: one thread per utterance (in `order`, so that the lists stay roughly length-sorted).  List slots are
// handed out per workgroup: wavefront ballots -> LDS counts -> ONE atomic per (workgroup, word) — 11 addresses
// shared by 1564 wavefronts made the per-wavefront atomics the kernel's whole run time.
__global__ __launch_bounds__(kBlock) void viterbi_select_kernel(
    const int32_t *__restrict__ order, int64_t n_utts, int64_t n_slots, int32_t W,
    const double *__restrict__ ascore, const double *__restrict__ aeps, int32_t *__restrict__ cand_cnt,
    int32_t *__restrict__ cand_utt, int32_t *__restrict__ cand_slot) {
  constexpr int kMaxW = 64;  // words per LDS pass (more words: the loop below repeats)
  __shared__ int wave_cnt[kMaxW][kBlock / kWave];
  __shared__ int block_base[kMaxW];
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const bool live = p < n_utts;
  const int64_t u = live ? (order ? static_cast<int64_t>(order[p]) : p) : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double thr = neg_inf();
  if (live)
    for (int w = 0; w < W; ++w) {
      const double lo = ascore[u * W + w] - aeps[u * W + w];
      if (lo > thr) thr = lo;  // NaN and -inf never raise the threshold
    }
  for (int w0 = 0; w0 < W; w0 += kMaxW) {
    const int nw = W - w0 < kMaxW ? W - w0 : kMaxW;
    unsigned long long keep_bits = 0;  // bit i: this utterance keeps word w0 + i
    for (int i = 0; i < nw; ++i) {
      bool keep = false;
      if (live) keep = !(ascore[u * W + w0 + i] + aeps[u * W + w0 + i] < thr);
      const unsigned long long mask = __ballot(keep);
      if (lane == 0) wave_cnt[i][wave] = __popcll(mask);
      if (keep) keep_bits |= 1ull << i;
    }
    __syncthreads();
    if (threadIdx.x < nw) {
      int tot = 0;
#pragma unroll
      for (int v = 0; v < kBlock / kWave; ++v) tot += wave_cnt[threadIdx.x][v];
      block_base[threadIdx.x] = tot ? atomicAdd(&cand_cnt[w0 + threadIdx.x], tot) : 0;
    }
    __syncthreads();
    for (int i = 0; i < nw; ++i) {
      const bool keep = (keep_bits >> i) & 1ull;
      const unsigned long long mask = __ballot(keep);
      if (live) {
        int my = -1;
        if (keep) {
          int base = block_base[i];
          for (int v = 0; v < wave; ++v) base += wave_cnt[i][v];
          my = base + __popcll(mask & ((1ull << lane) - 1ull));
          cand_utt[static_cast<int64_t>(w0 + i) * n_slots + my] = static_cast<int32_t>(u);
        }
        cand_slot[u * W + w0 + i] = my;
      }
    }
    __syncthreads();
  }
}

// pass D: arg-max over the kept words (exact scores) and the back-trace of that word
__global__ __launch_bounds__(kBlock) void viterbi_backtrace_pruned_kernel(
    const int64_t *__restrict__ offsets, const int32_t *__restrict__ order, int64_t n_utts, int64_t n_slots,
    int32_t max_T, int32_t W, const uint32_t *__restrict__ bp_all, const double *__restrict__ scores,
    const int32_t *__restrict__ last_state, const int32_t *__restrict__ cand_slot,
    int32_t *__restrict__ best_word, double *__restrict__ best_score, int32_t *__restrict__ path) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= n_utts) return;
  const int64_t u = order ? static_cast<int64_t>(order[p]) : p;
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  int bw = -1, bslot = -1;
  double bs = neg_inf();
  for (int w = 0; w < W; ++w) {
    const int sl = cand_slot[u * W + w];
    if (sl < 0) continue;
    const double sc = scores[u * W + w];
    if (sc > bs) {
      bs = sc;
      bw = w;
      bslot = sl;
    }
  }
  if (best_word) best_word[u] = bw;
  if (best_score) best_score[u] = bs;
  if (!path || T <= 0) return;
  // no word beats -inf: every word was kept (their intervals all reach -inf or are NaN) and, like the
  // all-vocabulary pass, the path of model 0 is reported
  const int wsel = bw < 0 ? 0 : bw;
  if (bw < 0) bslot = cand_slot[u * W];
  int s = last_state[u * W + wsel];
  path[beg + T - 1] = s;
  const uint32_t *__restrict__ bp = bp_all + (static_cast<int64_t>(wsel) * max_T) * n_slots + bslot;
  // the back-pointer words' addresses do not depend on the state chain: eight loads in flight per step of the walk
  // (one load, one dependent store at a time left this kernel waiting on memory: 0.13 ms per 100 000 utterances)
  constexpr int kAhead = 8;
  for (int t = T - 1; t >= 1; t -= kAhead) {
    uint32_t bits[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) bits[k] = t - k >= 1 ? bp[static_cast<int64_t>(t - k) * n_slots] : 0u;
#pragma unroll
    for (int k = 0; k < kAhead; ++k)
      if (t - k >= 1) {
        s -= static_cast<int>((bits[k] >> s) & 1u);
        path[beg + t - k - 1] = s;
      }
  }
}

// ---------------------------------------------------------------------------------------
// pass 2: pick the word (decoder.py:42-47 — strict '>' from -inf in model order, so NaN or
// all -inf leaves "no word" = -1 and the path of model 0 is reported) and walk the
// back-pointers of that model.
// ---------------------------------------------------------------------------------------
template <bool BIDIAG>
__global__ __launch_bounds__(kBlock) void viterbi_backtrace_kernel(
    const int64_t *__restrict__ offsets, const int32_t *__restrict__ order, int64_t n_utts,
    int64_t n_slots, int32_t max_T, int32_t W, int32_t S, const void *__restrict__ bp_raw,
    const double *__restrict__ scores, const int32_t *__restrict__ last_state,
    const int32_t *__restrict__ word_sel, int32_t *__restrict__ best_word,
    double *__restrict__ best_score, int32_t *__restrict__ path) {
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (slot >= n_utts) return;
  const int64_t u = order ? static_cast<int64_t>(order[slot]) : slot;
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);

  int bw = -1;
  double bs = neg_inf();
  if (word_sel) {
    bw = word_sel[u];
    bs = scores[u * W + bw];
  } else {
    for (int w = 0; w < W; ++w) {
      const double sc = scores[u * W + w];
      if (sc > bs) {
        bs = sc;
        bw = w;
      }
    }
  }
  if (best_word) best_word[u] = bw;
  if (best_score) best_score[u] = bs;
  if (!path || T <= 0) return;
  const int wsel = bw < 0 ? 0 : bw;

  int s = last_state[u * W + wsel];
  path[beg + T - 1] = s;
  if constexpr (BIDIAG) {
    const uint32_t *__restrict__ bp =
        static_cast<const uint32_t *>(bp_raw) + (static_cast<int64_t>(wsel) * max_T) * n_slots + slot;
    for (int t = T - 1; t >= 1; --t) {
      const uint32_t bits = bp[static_cast<int64_t>(t) * n_slots];
      s -= static_cast<int>((bits >> s) & 1u);
      path[beg + t - 1] = s;
    }
  } else {
    const uint8_t *__restrict__ bp = static_cast<const uint8_t *>(bp_raw) +
                                     (static_cast<int64_t>(wsel) * max_T) * S * n_slots + slot;
    for (int t = T - 1; t >= 1; --t) {
      s = bp[(static_cast<int64_t>(t) * S + s) * n_slots];
      path[beg + t - 1] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------
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
};

template <int D, int S, bool TIE, bool SEQ, bool FAST>
int launch_scores4(const ScoreArgs &a, int topology) {
  const int64_t tiles_pad = a.cand_utt ? a.n_tiles : round_up(a.n_tiles, kXcd);
  const int64_t blocks = tiles_pad * a.W;
  if (blocks > 0x7fffffffLL) return fail(SAPR_ERR_ARG, "grid too large (%lld blocks)", (long long)blocks);
  dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
  if (topology == SAPR_TOPO_BIDIAG) {
    if constexpr (S <= 32) {
      if (a.cand_utt)
        SAPR_LAUNCH((viterbi_bidiag_kernel<D, S, TIE, SEQ, FAST, true>), grid, block, 0, a.stream, a.feats, a.offsets,
                    a.order, a.n_utts, a.n_tiles, a.n_slots, a.max_T, a.W, a.prm, a.gconst, a.log_start, a.log_trans,
                    static_cast<uint32_t *>(a.bp), a.scores, a.last_state, a.cand_utt, a.cand_cnt);
      else
        SAPR_LAUNCH((viterbi_bidiag_kernel<D, S, TIE, SEQ, FAST, false>), grid, block, 0, a.stream, a.feats, a.offsets,
                    a.order, a.n_utts, a.n_tiles, a.n_slots, a.max_T, a.W, a.prm, a.gconst, a.log_start, a.log_trans,
                    static_cast<uint32_t *>(a.bp), a.scores, a.last_state, a.cand_utt, a.cand_cnt);
    } else {
      return fail(SAPR_ERR_UNSUPPORTED, "bidiagonal kernel needs S <= 32");
    }
  } else {
    SAPR_LAUNCH((viterbi_dense_kernel<D, S, TIE, SEQ, FAST>), grid, block, 0, a.stream, a.feats,
                       a.offsets, a.order, a.n_utts, a.n_tiles, a.n_slots, a.max_T, a.W, a.prm, a.gconst,
                       a.log_start, a.log_trans, static_cast<uint8_t *>(a.bp), a.scores, a.last_state);
  }
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

template <int D, int S>
int launch_scores(const ScoreArgs &a, int topology, int tie, int sum_order, int fast) {
  const int key = (tie == SAPR_TIE_HIGH ? 4 : 0) | (sum_order ? 2 : 0) | (fast ? 1 : 0);
  switch (key) {
    case 0: return launch_scores4<D, S, false, false, false>(a, topology);
    case 1: return launch_scores4<D, S, false, false, true>(a, topology);
    case 2: return launch_scores4<D, S, false, true, false>(a, topology);
    case 3: return launch_scores4<D, S, false, true, true>(a, topology);
    case 4: return launch_scores4<D, S, true, false, false>(a, topology);
    case 5: return launch_scores4<D, S, true, false, true>(a, topology);
    case 6: return launch_scores4<D, S, true, true, false>(a, topology);
    default: return launch_scores4<D, S, true, true, true>(a, topology);
  }
}

// builds the interleaved parameter blob and checks the fast-division domain (flag bit 0) and the domain of
// the pruned decoder's float32 bounding pass (flag bit 1: var in [1e-20, 1e20])
__global__ void diag_pack_kernel(const double *__restrict__ means, const double *__restrict__ vars,
                                 const double *__restrict__ gconst, const double *__restrict__ log_start,
                                 const double *__restrict__ log_trans, int W, int S, int D,
                                 double *__restrict__ blob, int *__restrict__ bad) {
  const int64_t n_prm = static_cast<int64_t>(W) * S * D;
  const int64_t n_ws = static_cast<int64_t>(W) * S;
  const int64_t n_tr = n_ws * S;
  const int64_t total = n_prm + 2 * n_ws + n_tr;
  const PackView pv = pack_view(blob, W, S, D);
  const int p32 = pack_p32(S, D);
  for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    if (i < n_prm) {
      const double m = means[i], v = vars[i];
      blob[4 * i + 0] = m;
      blob[4 * i + 1] = v;
      const double yh = 1.0 / v;  // IEEE division: correctly rounded reciprocal
      blob[4 * i + 2] = yh;
      blob[4 * i + 3] = __builtin_fma(-v, yh, 1.0) * yh;  // 1/v - yh = (1 - v*yh)/v; the residual is exact
      const double am = m < 0 ? -m : m;
      // an all-ones significand is the one operand class the literature singles out for
      // reciprocal-based division (Markstein 1990); it goes to the IEEE instantiation as well
      const bool all_ones = (__double_as_longlong(v) & 0xFFFFFFFFFFFFFll) == 0xFFFFFFFFFFFFFll;
      const bool ok = v >= 1e-30 && v <= 1e30 && !all_ones && (am == 0.0 || (am >= 1e-30 && am <= 1e30));
      if (!ok) atomicOr(bad, 1);
      if (!(v >= 1e-20 && v <= 1e20)) atomicOr(bad, 2);
      const int64_t w = i / (static_cast<int64_t>(S) * D), e = i - w * S * D;
      float2 pr;
      pr.x = static_cast<float>(m);
      pr.y = static_cast<float>(yh);
      reinterpret_cast<float2 *>(const_cast<double *>(pv.prm32))[w * p32 + e] = pr;
    } else if (i < n_prm + n_ws) {
      blob[4 * n_prm + (i - n_prm)] = gconst[i - n_prm];
      const_cast<double *>(pv.hgc)[i - n_prm] = -0.5 * gconst[i - n_prm];
    } else if (i < n_prm + 2 * n_ws) {
      blob[4 * n_prm + (i - n_prm)] = log_start[i - n_prm - n_ws];
    } else {
      blob[4 * n_prm + (i - n_prm)] = log_trans[i - n_prm - 2 * n_ws];
    }
  }
}

// per-word constants of the bounding pass: one thread per word
__device__ __forceinline__ double nan_max(double a, double b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }
__global__ void diag_pack_consts_kernel(const double *__restrict__ means, const double *__restrict__ vars,
                                        const double *__restrict__ gconst, const double *__restrict__ log_start,
                                        const double *__restrict__ log_trans, int W, int S, int D,
                                        double *__restrict__ blob) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const PackView pv = pack_view(blob, W, S, D);
  double cmax = 0.0, gcs = 0.0, lts = 0.0, lss = 0.0;
  for (int s = 0; s < S; ++s) {
    double c = 0.0;
    for (int d = 0; d < D; ++d) {
      const int64_t i = (static_cast<int64_t>(w) * S + s) * D + d;
      c += means[i] * means[i] / vars[i];
    }
    cmax = nan_max(cmax, c);
    gcs += fabs(gconst[w * S + s]);
    const double l0 = log_start[w * S + s];
    if (l0 != neg_inf()) lss += fabs(l0);
    for (int s2 = 0; s2 < S; ++s2) {
      const double l = log_trans[(static_cast<int64_t>(w) * S + s) * S + s2];
      if (l != neg_inf()) lts += fabs(l);
    }
  }
  double *wc = const_cast<double *>(pv.wconst) + static_cast<int64_t>(w) * 4;
  wc[0] = cmax;
  wc[1] = gcs;
  wc[2] = lts;
  wc[3] = lss;
  // pad slots of the float32 parameter rows (never consumed; defined for reproducible blobs)
  for (int e = S * D; e < pack_p32(S, D); ++e)
    const_cast<double *>(pv.prm32)[static_cast<int64_t>(w) * pack_p32(S, D) + e] = 0.0;
}

// ---- operands of the matrix-core bounding pass (layout: emission.h PackView) ----
// centre (mean of the state means), the range r_d = max_(w,s) |mean - centre| + 8 sigma the features are expected
// in, and from it the power-of-two factors that put x' (linear slots: |x' a| <= 2^14) and x'^2 (squared slots:
// (x' a)^2 <= 2^14) into half range.  A frame outside that range overflows the halves; the kernel notices and
// keeps every word of that utterance.
__global__ void diag_pack_center_kernel(const double *__restrict__ means, const double *__restrict__ vars, int W,
                                        int S, int D, double *__restrict__ blob) {
  const PackView pv = pack_view(blob, W, S, D);
  const int G8 = 8 * gemm_groups(D);
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d == 0) const_cast<double *>(pv.gkw)[2 * W + 2] = 0.0;  // running maximum of |P / slot scale| (bits)
  if (d >= G8) return;
  float *out = const_cast<float *>(pv.gctr);
  float c = 0.0f, a_sq = 1.0f, a_lin = 1.0f;
  if (d < D) {
    const int64_t n = static_cast<int64_t>(W) * S;
    double m = 0.0;
    for (int64_t ws = 0; ws < n; ++ws) m += means[ws * D + d];
    c = static_cast<float>(m / static_cast<double>(n));
    double r = 0.0;
    for (int64_t ws = 0; ws < n; ++ws)
      r = nan_max(r, fabs(means[ws * D + d] - static_cast<double>(c)) + 8.0 * sqrt(vars[ws * D + d]));
    auto pow2_below = [](double v) {  // largest power of two <= v, exponent clamped to [-40, 40]
      int e = (v > 0.0 && v < 1e300) ? static_cast<int>(floor(log2(v))) : 40;
      e = e < -40 ? -40 : (e > 40 ? 40 : e);
      return static_cast<float>(ldexp(1.0, e));
    };
    a_sq = pow2_below(128.0 / r);
    a_lin = pow2_below(16384.0 / r);
  }
  out[d] = c;
  out[G8 + d] = a_sq;
  out[2 * G8 + d] = a_lin;
}

// entry (state j of word w, slot i of group g) of P with the slot's feature factor divided out:
//   squared slot  -y/2 / a_sq^2     linear slot  y mu' / a_lin     constant slot (phi = 1024)  [-(c0 + gconst)/2 + sg] / 1024
// *noself_elsewhere: the state has no self-loop and sits at a chain position the kernel has no mask for
__device__ double gemm_entry(const double *__restrict__ means, const double *__restrict__ vars,
                             const double *__restrict__ gconst, const double *__restrict__ log_trans,
                             const PackView &pv, int W, int S, int D, int w, int j, int g, int i, bool *noself_elsewhere) {
  const int G = gemm_groups(D), G8 = 8 * G;
  if (j >= S) return 0.0;
  const int64_t row = (static_cast<int64_t>(w) * S + j) * D;
  if (g < G) {
    const int f = 8 * g + i;
    if (f < D) {
      const double a = static_cast<double>(pv.gctr[G8 + f]);
      return -0.5 * (1.0 / vars[row + f]) / (a * a);
    }
    if (f == D) {
      double c0 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double mu = means[row + d] - static_cast<double>(pv.gctr[d]);
        c0 += mu * mu / vars[row + d];
      }
      double sg = log_trans[(static_cast<int64_t>(w) * S + j) * S + j];
      if (sg == neg_inf()) {  // no self-loop: handled in the lattice (positions 0, 4, 8, 12 of the chain only)
        if (j % 4 != 0) *noself_elsewhere = true;
        sg = 0.0;
      }
      return (-0.5 * (c0 + gconst[static_cast<int64_t>(w) * S + j]) + sg) * (1.0 / 1024.0);
    }
  } else if (g < 2 * G) {
    const int f = 8 * (g - G) + i;
    if (f < D)
      return (means[row + f] - static_cast<double>(pv.gctr[f])) / vars[row + f] /
             static_cast<double>(pv.gctr[2 * G8 + f]);
  }
  return 0.0;
}

// pass 1 over the entries: their largest magnitude (positive doubles order like their bit patterns)
__global__ void diag_pack_gemm_max_kernel(const double *__restrict__ means, const double *__restrict__ vars,
                                          const double *__restrict__ gconst, const double *__restrict__ log_trans,
                                          int W, int S, int D, double *__restrict__ blob, int *__restrict__ bad) {
  const PackView pv = pack_view(blob, W, S, D);
  const int G = gemm_groups(D);
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= static_cast<int64_t>(W) * S * 2 * G) return;
  const int g = static_cast<int>(idx % (2 * G));
  const int j = static_cast<int>((idx / (2 * G)) % S);
  const int w = static_cast<int>(idx / (2 * G) / S);
  bool elsewhere = false;
  double m = 0.0;
  for (int i = 0; i < 8; ++i) {
    const double v = fabs(gemm_entry(means, vars, gconst, log_trans, pv, W, S, D, w, j, g, i, &elsewhere));
    if (!(v <= 1e300)) atomicOr(bad, 4);  // NaN or infinite coefficient
    else if (v > m) m = v;
  }
  if (elsewhere) atomicOr(bad, 4);
  atomicMax(reinterpret_cast<unsigned long long *>(const_cast<double *>(pv.gkw) + 2 * W + 2),
            static_cast<unsigned long long>(__double_as_longlong(m)));
}

// pass 2, one thread per (word, row tile, k chunk, lane): its eight entries times 2^g (g: the largest entry
// lands in [2^13, 2^14)), each as two halves hi = RN(v), lo = RN(v - hi)
__global__ void diag_pack_gemm_kernel(const double *__restrict__ means, const double *__restrict__ vars,
                                      const double *__restrict__ gconst, const double *__restrict__ log_trans,
                                      int W, int S, int D, double *__restrict__ blob, int *__restrict__ bad) {
  const PackView pv = pack_view(blob, W, S, D);
  const int KC = gemm_kchunks(D), RT = gemm_rtiles(S);
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= static_cast<int64_t>(W) * RT * KC * 64) return;
  const double pmax = pv.gkw[2 * W + 2];
  int ge = (pmax > 0.0) ? static_cast<int>(floor(log2(16384.0 / pmax))) : 0;
  ge = ge < -60 ? -60 : (ge > 60 ? 60 : ge);
  const double scale = ldexp(1.0, ge);
  if (idx == 0) {
    const_cast<double *>(pv.gkw)[2 * W] = scale;
    const_cast<double *>(pv.gkw)[2 * W + 1] = ldexp(1.0, -ge);
    if (!(pmax > 0.0) || pmax * scale >= 32768.0) atomicOr(bad, 4);
  }
  const int lane = static_cast<int>(idx & 63);
  int64_t rest = idx >> 6;
  const int c = static_cast<int>(rest % KC);
  rest /= KC;
  const int rt = static_cast<int>(rest % RT);
  const int w = static_cast<int>(rest / RT);
  const int j = 16 * rt + (lane & 15), g = 4 * c + (lane >> 4);
  unsigned pc[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  bool elsewhere = false;
  for (int i = 0; i < 8; ++i) {
    const float v32 = static_cast<float>(gemm_entry(means, vars, gconst, log_trans, pv, W, S, D, w, j, g, i, &elsewhere) * scale);
    const _Float16 hi = static_cast<_Float16>(v32);
    const _Float16 lo = static_cast<_Float16>(v32 - static_cast<float>(hi));
    const int sh = 16 * (i & 1);
    pc[0][i >> 1] |= static_cast<unsigned>(__builtin_bit_cast(unsigned short, hi)) << sh;
    pc[1][i >> 1] |= static_cast<unsigned>(__builtin_bit_cast(unsigned short, lo)) << sh;
  }
  uint4 *out = const_cast<uint4 *>(pv.gfrag);
  for (int p = 0; p < 2; ++p)
    out[(((static_cast<int64_t>(w) * RT + rt) * KC + c) * 2 + p) * 64 + lane] =
        make_uint4(pc[p][0], pc[p][1], pc[p][2], pc[p][3]);
}

// per-word constants of the matrix-core bound: max_j 3 c0_j + 2 |gconst_j| + 4 |sg_j| (log-density units) and
// max_j sum_k |P_jk 2^g| (the units of the half operands); runs after diag_pack_gemm_kernel
__global__ void diag_pack_gemm_consts_kernel(const double *__restrict__ means, const double *__restrict__ vars,
                                             const double *__restrict__ gconst, const double *__restrict__ log_trans,
                                             int W, int S, int D, double *__restrict__ blob) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const PackView pv = pack_view(blob, W, S, D);
  const double scale = pv.gkw[2 * W];
  double k = 0.0, psum = 0.0;
  for (int s = 0; s < S; ++s) {
    double c0 = 0.0;
    for (int d = 0; d < D; ++d) {
      const int64_t i = (static_cast<int64_t>(w) * S + s) * D + d;
      const double mu = means[i] - static_cast<double>(pv.gctr[d]);
      c0 += mu * mu / vars[i];
    }
    double sg = log_trans[(static_cast<int64_t>(w) * S + s) * S + s];
    if (sg == neg_inf()) sg = 0.0;
    k = nan_max(k, 3.0 * c0 + 2.0 * fabs(gconst[w * S + s]) + 4.0 * fabs(sg));
    double row = 0.0;
    bool unused = false;
    for (int g = 0; g < 2 * gemm_groups(D); ++g)
      for (int i = 0; i < 8; ++i)
        row += fabs(gemm_entry(means, vars, gconst, log_trans, pv, W, S, D, w, s, g, i, &unused)) * scale;
    psum = nan_max(psum, row);
  }
  const_cast<double *>(pv.gkw)[w] = k;
  const_cast<double *>(pv.gkw)[W + w] = psum;
}

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

template <int D, int S, int WC>
int launch_approx_mfma(const ScoreArgs &a, const PackView &pv, double *ascore, double *aeps) {
  const int64_t blocks = (a.n_utts + 15) / 16 * ((a.W + WC - 1) / WC);
  if (blocks > 0x7fffffffLL) return fail(SAPR_ERR_ARG, "grid too large (%lld blocks)", (long long)blocks);
  SAPR_LAUNCH((viterbi_approx_mfma_kernel<D, S, WC>), dim3(static_cast<unsigned>(blocks)), dim3(kWave), 0, a.stream,
              a.feats, a.offsets, a.order, a.n_utts, a.W, pv.gfrag, pv.gctr, pv.gkw, pv.log_start, pv.log_trans,
              pv.wconst, ascore, aeps);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

template <int D, int S>
int launch_approx(const ScoreArgs &a, const PackView &pv, double *ascore, double *aeps, int pack_flags) {
  if constexpr (S <= 32) {
    if (pack_flags & SAPR_PACK_GEMM_OK) {  // callers clear the bit to keep pass A on the vector ALU
      // words per wavefront pass: as many as 256 registers hold (two wavefronts per SIMD).  13 dims, 10 states:
      // 6 amortise the feature operands best (0.78 vs 0.91 ms for 4 on the benchmark shape); small vocabularies
      // waste fewer slots with 4
      if constexpr (D <= 16 && S <= 16) {
        if (a.W >= 5 && SAPR_MFMA_WC == 4) return launch_approx_mfma<D, S, 6>(a, pv, ascore, aeps);
        return launch_approx_mfma<D, S, SAPR_MFMA_WC>(a, pv, ascore, aeps);
      } else if constexpr (D <= 16) {
        return launch_approx_mfma<D, S, 3>(a, pv, ascore, aeps);
      } else if constexpr (S <= 16) {
        return launch_approx_mfma<D, S, 2>(a, pv, ascore, aeps);
      } else {
        return launch_approx_mfma<D, S, 1>(a, pv, ascore, aeps);
      }
    }
  }
  const int64_t blocks = round_up(a.n_tiles, kXcd) * a.W;
  if (blocks > 0x7fffffffLL) return fail(SAPR_ERR_ARG, "grid too large (%lld blocks)", (long long)blocks);
  SAPR_LAUNCH((viterbi_approx_kernel<D, S>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, a.stream, a.feats,
              a.offsets, a.order, a.n_utts, a.n_tiles, a.W, pv.prm32, pv.hgc, pv.log_start, pv.log_trans, pv.wconst,
              ascore, aeps);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

size_t workspace_bytes(int64_t n_utts, int W, int S, int max_T, int topology) {
  const int64_t n_slots = round_up(n_utts > 0 ? n_utts : 1, kBlock);
  const size_t per = topology == SAPR_TOPO_BIDIAG ? sizeof(uint32_t) : static_cast<size_t>(S);
  return static_cast<size_t>(W) * static_cast<size_t>(max_T > 0 ? max_T : 1) * n_slots * per;
}

}  // namespace
}  // namespace sapr

using namespace sapr;

extern "C" int sapr_viterbi_workspace_bytes(int64_t n_utts, int32_t W, int32_t S, int32_t max_T,
                                            int32_t topology, size_t *bytes) {
  SAPR_REQUIRE(bytes != nullptr, "bytes is NULL");
  SAPR_REQUIRE(n_utts >= 0 && W > 0 && S > 0 && max_T >= 0, "bad sizes");
  SAPR_REQUIRE(topology == SAPR_TOPO_DENSE || topology == SAPR_TOPO_BIDIAG, "bad topology");
  *bytes = workspace_bytes(n_utts, W, S, max_T, topology);
  return 0;
}

extern "C" int sapr_diag_pack_bytes(int32_t W, int32_t S, int32_t D, size_t *bytes) {
  SAPR_REQUIRE(bytes != nullptr && W > 0 && S > 0 && D > 0, "bad arguments");
  *bytes = pack_doubles(W, S, D) * sizeof(double) + 64;
  return 0;
}

extern "C" int sapr_diag_pack(const double *means, const double *vars, const double *gconst,
                              const double *log_start, const double *log_trans, int32_t W, int32_t S,
                              int32_t D, void *pack, size_t pack_bytes, int32_t *fast_div_ok,
                              void *stream) {
  SAPR_REQUIRE(W > 0 && S > 0 && D > 0, "bad sizes");
  SAPR_REQUIRE(means && vars && gconst && log_start && log_trans && pack, "NULL pointer argument");
  SAPR_REQUIRE(pack_bytes >= pack_doubles(W, S, D) * sizeof(double) + 64, "pack buffer too small");
  hipStream_t st = as_stream(stream);
  int *flag = reinterpret_cast<int *>(static_cast<double *>(pack) + pack_doubles(W, S, D));
  SAPR_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), st));
  SAPR_LAUNCH(diag_pack_kernel, dim3(64), dim3(256), 0, st, means, vars, gconst, log_start, log_trans,
                     W, S, D, static_cast<double *>(pack), flag);
  SAPR_LAUNCH(diag_pack_consts_kernel, dim3((W + 63) / 64), dim3(64), 0, st, means, vars, gconst, log_start,
              log_trans, W, S, D, static_cast<double *>(pack));
  SAPR_LAUNCH(diag_pack_center_kernel, dim3(1), dim3(64 * ((8 * gemm_groups(D) + 63) / 64)), 0, st, means, vars, W, S,
              D, static_cast<double *>(pack));
  const int64_t n_ent = static_cast<int64_t>(W) * S * 2 * gemm_groups(D);
  SAPR_LAUNCH(diag_pack_gemm_max_kernel, dim3(static_cast<unsigned>((n_ent + 255) / 256)), dim3(256), 0, st, means,
              vars, gconst, log_trans, W, S, D, static_cast<double *>(pack), flag);
  const int64_t n_gemm = static_cast<int64_t>(W) * gemm_rtiles(S) * gemm_kchunks(D) * 64;
  SAPR_LAUNCH(diag_pack_gemm_kernel, dim3(static_cast<unsigned>((n_gemm + 255) / 256)), dim3(256), 0, st, means, vars,
              gconst, log_trans, W, S, D, static_cast<double *>(pack), flag);
  SAPR_LAUNCH(diag_pack_gemm_consts_kernel, dim3((W + 63) / 64), dim3(64), 0, st, means, vars, gconst, log_trans, W, S,
              D, static_cast<double *>(pack));
  SAPR_HIP_TRY(hipGetLastError());
  int bad = 0;
  SAPR_HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, st));
  SAPR_HIP_TRY(hipStreamSynchronize(st));  // model preparation, not the data path
  if (fast_div_ok)
    *fast_div_ok = ((bad & 1) ? 0 : SAPR_PACK_FAST_DIV) | ((bad & 2) ? 0 : SAPR_PACK_BOUND_OK) |
                   ((bad & 6) ? 0 : SAPR_PACK_GEMM_OK);
  return 0;
}

extern "C" int sapr_viterbi_diag_scores(const float *feats, const int64_t *offsets,
                                        const int32_t *order, int64_t n_utts, int32_t D,
                                        int32_t max_T, const void *pack, int32_t W, int32_t S,
                                        int32_t topology, int32_t tie, int32_t sum_order,
                                        int32_t fast_div, void *workspace, size_t workspace_size,
                                        double *scores, int32_t *last_state, void *stream) {
  SAPR_REQUIRE(n_utts >= 0 && W > 0 && S > 0 && D > 0 && max_T >= 0, "bad sizes");
  SAPR_REQUIRE(topology == SAPR_TOPO_DENSE || topology == SAPR_TOPO_BIDIAG, "bad topology");
  SAPR_REQUIRE(tie == SAPR_TIE_LOW || tie == SAPR_TIE_HIGH, "bad tie-break");
  SAPR_REQUIRE(sum_order == SAPR_SUM_PAIRWISE || sum_order == SAPR_SUM_TVIEW, "bad sum_order");
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(feats && offsets && pack && scores && last_state && workspace, "NULL pointer argument");
  if (workspace_size < workspace_bytes(n_utts, W, S, max_T, topology))
    return fail(SAPR_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_size,
                workspace_bytes(n_utts, W, S, max_T, topology));
  const PackView pv = pack_view(pack, W, S, D);
  ScoreArgs a;
  a.feats = feats;
  a.offsets = offsets;
  a.order = order;
  a.n_utts = n_utts;
  a.n_tiles = (n_utts + kBlock - 1) / kBlock;
  a.n_slots = round_up(n_utts, kBlock);
  a.max_T = max_T > 0 ? max_T : 1;
  a.W = W;
  a.prm = pv.prm;
  a.gconst = pv.gconst;
  a.log_start = pv.log_start;
  a.log_trans = pv.log_trans;
  a.bp = workspace;
  a.scores = scores;
  a.last_state = last_state;
  a.stream = as_stream(stream);
  const int fast = (fast_div & SAPR_PACK_FAST_DIV) ? 1 : 0;
  if (D == 13 && S == 10) return launch_scores<13, 10>(a, topology, tie, sum_order, fast);
#ifndef SAPR_ONLY_13_10  // dev builds: -DSAPR_ONLY_13_10 compiles the benchmark shape only
  if (D == 13 && S == 18) return launch_scores<13, 18>(a, topology, tie, sum_order, fast);
  if (D == 39 && S == 10) return launch_scores<39, 10>(a, topology, tie, sum_order, fast);
  if (D == 39 && S == 18) return launch_scores<39, 18>(a, topology, tie, sum_order, fast);
#endif
  return fail(SAPR_ERR_UNSUPPORTED,
              "viterbi kernels are instantiated for (D,S) in {13,39}x{10,18}; got D=%d S=%d", D, S);
}

extern "C" int sapr_viterbi_backtrace(const int64_t *offsets, const int32_t *order, int64_t n_utts,
                                      int32_t max_T, int32_t W, int32_t S, int32_t topology,
                                      const void *workspace, size_t workspace_size,
                                      const double *scores, const int32_t *last_state,
                                      const int32_t *word_sel, int32_t *best_word,
                                      double *best_score, int32_t *path, void *stream) {
  SAPR_REQUIRE(n_utts >= 0 && W > 0 && S > 0 && max_T >= 0, "bad sizes");
  SAPR_REQUIRE(topology == SAPR_TOPO_DENSE || topology == SAPR_TOPO_BIDIAG, "bad topology");
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(offsets && scores && last_state && workspace, "NULL pointer argument");
  if (workspace_size < workspace_bytes(n_utts, W, S, max_T, topology))
    return fail(SAPR_ERR_WORKSPACE, "workspace too small");
  const int64_t n_slots = round_up(n_utts, kBlock);
  const int mt = max_T > 0 ? max_T : 1;
  dim3 grid(static_cast<unsigned>((n_utts + kBlock - 1) / kBlock)), block(kBlock);
  if (topology == SAPR_TOPO_BIDIAG)
    SAPR_LAUNCH(viterbi_backtrace_kernel<true>, grid, block, 0, as_stream(stream), offsets,
                       order, n_utts, n_slots, mt, W, S, workspace, scores, last_state, word_sel,
                       best_word, best_score, path);
  else
    SAPR_LAUNCH(viterbi_backtrace_kernel<false>, grid, block, 0, as_stream(stream), offsets,
                       order, n_utts, n_slots, mt, W, S, workspace, scores, last_state, word_sel,
                       best_word, best_score, path);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_viterbi_pruned_workspace_bytes(int64_t n_utts, int32_t W, int32_t S, int32_t max_T,
                                                   size_t *bytes) {
  SAPR_REQUIRE(bytes != nullptr, "bytes is NULL");
  SAPR_REQUIRE(n_utts >= 0 && W > 0 && S > 0 && S <= 32 && max_T >= 0, "bad sizes");
  *bytes = pruned_layout(n_utts, W, max_T).total;
  return 0;
}

extern "C" int sapr_viterbi_decode_pruned(const float *feats, const int64_t *offsets, const int32_t *order,
                                          int64_t n_utts, int32_t D, int32_t max_T, const void *pack, int32_t W,
                                          int32_t S, int32_t tie, int32_t sum_order, int32_t pack_flags,
                                          void *workspace, size_t workspace_size, int32_t *best_word,
                                          double *best_score, int32_t *path, void *stream) {
  SAPR_REQUIRE(n_utts >= 0 && W > 0 && S > 0 && D > 0 && max_T >= 0, "bad sizes");
  SAPR_REQUIRE(tie == SAPR_TIE_LOW || tie == SAPR_TIE_HIGH, "bad tie-break");
  SAPR_REQUIRE(sum_order == SAPR_SUM_PAIRWISE || sum_order == SAPR_SUM_TVIEW, "bad sum_order");
  if (!(pack_flags & SAPR_PACK_BOUND_OK))
    return fail(SAPR_ERR_UNSUPPORTED, "model pack is outside the bounding pass's domain (variances in "
                                      "[1e-20, 1e20]): use sapr_viterbi_diag_scores + sapr_viterbi_backtrace");
  if (S > 32) return fail(SAPR_ERR_UNSUPPORTED, "the pruned decoder covers the bidiagonal topology (S <= 32)");
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(feats && offsets && pack && workspace && best_word && best_score, "NULL pointer argument");
  const PrunedLayout L = pruned_layout(n_utts, W, max_T);
  if (workspace_size < L.total)
    return fail(SAPR_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_size, L.total);
  char *ws = static_cast<char *>(workspace);
  auto *cand_utt = reinterpret_cast<int32_t *>(ws + L.cand_utt);
  auto *cand_slot = reinterpret_cast<int32_t *>(ws + L.cand_slot);
  auto *ascore = reinterpret_cast<double *>(ws + L.ascore);
  auto *aeps = reinterpret_cast<double *>(ws + L.aeps);
  auto *scores = reinterpret_cast<double *>(ws + L.scores);
  auto *last = reinterpret_cast<int32_t *>(ws + L.last);
  auto *cnt = reinterpret_cast<int32_t *>(ws + L.cnt);
  const PackView pv = pack_view(pack, W, S, D);
  ScoreArgs a;
  a.feats = feats;
  a.offsets = offsets;
  a.order = order;
  a.n_utts = n_utts;
  a.n_tiles = (n_utts + kBlock - 1) / kBlock;
  a.n_slots = round_up(n_utts, kBlock);
  a.max_T = max_T > 0 ? max_T : 1;
  a.W = W;
  a.prm = pv.prm;
  a.gconst = pv.gconst;
  a.log_start = pv.log_start;
  a.log_trans = pv.log_trans;
  a.bp = ws + L.bp;
  a.scores = scores;
  a.last_state = last;
  a.stream = as_stream(stream);
  int rc;
  // pass A: float32 bounds
  if (D == 13 && S == 10) rc = launch_approx<13, 10>(a, pv, ascore, aeps, pack_flags);
#ifndef SAPR_ONLY_13_10
  else if (D == 13 && S == 18) rc = launch_approx<13, 18>(a, pv, ascore, aeps, pack_flags);
  else if (D == 39 && S == 10) rc = launch_approx<39, 10>(a, pv, ascore, aeps, pack_flags);
  else if (D == 39 && S == 18) rc = launch_approx<39, 18>(a, pv, ascore, aeps, pack_flags);
#endif
  else
    return fail(SAPR_ERR_UNSUPPORTED,
                "viterbi kernels are instantiated for (D,S) in {13,39}x{10,18}; got D=%d S=%d", D, S);
  if (rc) return rc;
  // pass B: candidate lists
  SAPR_HIP_TRY(hipMemsetAsync(cnt, 0, static_cast<size_t>(W) * sizeof(int32_t), a.stream));
  dim3 ugrid(static_cast<unsigned>(a.n_tiles)), block(kBlock);
  SAPR_LAUNCH(viterbi_select_kernel, ugrid, block, 0, a.stream, order, n_utts, a.n_slots, W, ascore, aeps, cnt,
              cand_utt, cand_slot);
  SAPR_HIP_TRY(hipGetLastError());
  // pass C: exact lattice + back-pointers over the lists
  a.cand_utt = cand_utt;
  a.cand_cnt = cnt;
  const int fast = (pack_flags & SAPR_PACK_FAST_DIV) ? 1 : 0;
  if (D == 13 && S == 10) rc = launch_scores<13, 10>(a, SAPR_TOPO_BIDIAG, tie, sum_order, fast);
#ifndef SAPR_ONLY_13_10
  else if (D == 13 && S == 18) rc = launch_scores<13, 18>(a, SAPR_TOPO_BIDIAG, tie, sum_order, fast);
  else if (D == 39 && S == 10) rc = launch_scores<39, 10>(a, SAPR_TOPO_BIDIAG, tie, sum_order, fast);
  else rc = launch_scores<39, 18>(a, SAPR_TOPO_BIDIAG, tie, sum_order, fast);
#endif
  if (rc) return rc;
  // pass D: arg-max + back-trace
  SAPR_LAUNCH(viterbi_backtrace_pruned_kernel, ugrid, block, 0, a.stream, offsets, order, n_utts, a.n_slots, a.max_T,
              W, reinterpret_cast<const uint32_t *>(ws + L.bp), scores, last, cand_slot, best_word, best_score, path);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

/* debugging / test access to the pruned decoder's intermediate arrays inside `workspace` (device pointers) */
extern "C" int sapr_viterbi_pruned_views(int64_t n_utts, int32_t W, int32_t max_T, void *workspace,
                                         double **approx_score, double **approx_eps, double **exact_score,
                                         int32_t **cand_slot, int32_t **cand_count) {
  SAPR_REQUIRE(workspace != nullptr && n_utts >= 0 && W > 0, "bad arguments");
  const PrunedLayout L = pruned_layout(n_utts, W, max_T);
  char *ws = static_cast<char *>(workspace);
  if (approx_score) *approx_score = reinterpret_cast<double *>(ws + L.ascore);
  if (approx_eps) *approx_eps = reinterpret_cast<double *>(ws + L.aeps);
  if (exact_score) *exact_score = reinterpret_cast<double *>(ws + L.scores);
  if (cand_slot) *cand_slot = reinterpret_cast<int32_t *>(ws + L.cand_slot);
  if (cand_count) *cand_count = reinterpret_cast<int32_t *>(ws + L.cnt);
  return 0;
}
