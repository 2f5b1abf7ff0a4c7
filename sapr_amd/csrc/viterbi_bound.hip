// Bounding passes of the pruned decoder (pass A of sapr_viterbi_decode_pruned): float32 emission sums on the vector
// ALU or on the matrix cores, with a rigorous interval around every word's exact Viterbi score.
#include <algorithm>
#include <cstdlib>

#include "viterbi_shared.h"

namespace sapr {
namespace {

using namespace emission;

// ---------------------------------------------------------------------------------------
// Pruned decoder.  decoder.py:35-49 returns only the best word, its score and its state path, so the
// exact lattice is needed for the words that can still be the arg-max.  Pass A bounds every word's score:
//
//   ascore[u][w]  the Viterbi score with float32 emission sums (3 float32 VALU instructions per (state, dim)
//                 instead of 7 float64 ones, no back-pointers), lattice recursion in float64;
//   aeps[u][w]    a bound on |ascore - exact score|, accumulated alongside.  Error sources, with u32 = 2^-24,
//                 u = 2^-53, q = a state's float32 sum, C = sum_d mean^2/var (per state, precomputed):
//                   float(mean), float(1/var), x - mean, square, D-term fma chain:  |q - Q| <= (D + 6) u32 Q + 1.1 u32 C
//                   (the mean's rounding enters as 2 |x-mean| |mean| u32 / var <= u32 (Q_d + C_d));
//                   the exact kernel's own float64 rounding of Q: <= 20 u Q;
//                   b = -0.5 (gconst + Q): one rounding each side; the lattice: <= 2 additions per frame on
//                   each side, each within u of the running magnitude.
//                 With M = sum over frames and states of q (per frame in float32, frames in float64):
//                   eps = 2 * [ u32 ((D + 7) / 2 M + 0.6 T Cmax)        (10 M at 13 dimensions, 23 M at 39)
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
    const double e32 = u32 * (0.5 * (D + 7) * mag + 0.6 * Td * cmax);  // D-term chain: (D + 6) u32 Q on q, half on b
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
// r_j = lt_(j-1)j - sg_(j-1); dividing the cumulated weights R_j = r_1 + .. + r_j out of the column
// (v[j] = u[j] - R_j) removes them from the recursion altogether: v[j] = max(v[j-1], v[j]) + (b_j + sg_j), two
// float32 instructions per state and frame (sapr_diag_pack stores R; a -inf forward weight inside the reachable
// chain clears PACK_GEMM_OK, an unreachable tail of padding states is given zero rows and left out of the final
// maximum).  A state without a self-loop (the reference's entry state, hmmlearn_hmm.py:45-78) has its own candidate
// turned into a NaN, which v_max_f32 drops.
//
// Lanes and tiles: see the two kernels below (round 2's kernel gave every word its own 16-row tiles with states along
// the accumulator registers and fetched u[4q-1] by ds_bpermute; it is gone).
//
// Interval.  Let R = sum_k |P_k phi_k| for a (frame, state), in log-density units.  The computed value differs
// from the real-number one by at most cacc * 2^-24 * R + A, cacc = 36 + 68 KC:
//   feature centring, squaring and the float32 rounding of P                          4
//   phi as two truncated halves (2^-20), P as two rounded halves (2^-22), lo*lo (2^-21) 28
//   33 additions per MFMA, each allowed a whole ulp: the two small MFMAs (sums <= 2^-9 R) then the KC leading ones
//   A = 2^-14 2^-g (T max_j sum_k |P_jk 2^g| + sum_t sum_k |slot value|): a half below 2^-14 may be flushed
// With A2 = sum y x'^2, Q = quadratic form >= 0 and |2 y mu' x'| <= y x'^2 / 2 + 2 y mu'^2:  A2 <= 2 Q + 2 c0 and
// R <= 3 |value| + 3 c0 + 2 |gconst| + 4 |sg_j|.  Errors add up along a path, one state per frame: the kernels bound
// sum_t |value| along the two paths that matter (below).  The fp64 terms are as for the VALU kernel.  A slot value
// beyond the largest half (v_cvt_pkrtz saturates silently) or any non-finite arithmetic makes eps non-finite, which
// keeps the word.
// ---------------------------------------------------------------------------------------
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

__device__ __forceinline__ float nan_where(float v, unsigned long long mask) {
  unsigned b = __builtin_bit_cast(unsigned, v);
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(0x7FC00000u), "s"(mask));
  return __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float max_drop_nan(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
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

// ---------------------------------------------------------------------------------------
// pass A on the matrix cores, DENSE layout (round 3, second half): the states of WP words are laid back to back
// along the MFMA's N axis — row g = 16 tau + lane % 16 of tile tau is state g % S of word g / S — instead of one
// 16-row tile (or two) per word, and the operand roles are swapped: A = phi (M = 16 utterances), B = P (N = 16
// states), so a lane holds ONE state for the four utterances 4 q .. 4 q + 3 (q = lane / 16).  What that buys:
//   * no padding rows: 110 states of an 11-word, 10-state vocabulary are 7 tiles instead of 11 (21 MFMAs per frame
//     instead of 33, 28 column registers instead of 44), 36 states of two 18-state words 3 tiles instead of 4;
//   * the chain predecessor v[j-1] is the neighbouring LANE: one DPP row_shr:1 (plus a row_ror:1 of the tile below
//     for lane 0), where the tile-per-word layout needed a ds_bpermute per word and frame;
//   * one operand build and one feature read per frame for all WP words.
// Word starts (g % S == 0) take -inf instead of the neighbour (one v_cndmask_b32 on a compile-time lane mask).
// States without a self-loop: own candidate -> NaN, dropped by v_max_f32; the usual case — only the entry state of
// each word, which has no predecessor either — needs that in frame 1 only (the state is -inf ever after), a model
// with such a state inside the chain takes the `generic` instantiation of the step in every frame.
//
// Interval.  As above per (frame, state): |computed - real| <= cacc 2^-24 R + A, R <= 3 |value| + K_w.  Errors add up
// ALONG A PATH, so what is needed is M(pi) = sum_t |e_pi(t)| for two paths only: pi_A, the best path of the computed
// lattice, and pi_E, the best path of the exact one.  An emission is e = -(gconst_j + Q) / 2 + sg_j with Q >= 0, so
// with tau = max(0, max_j (-gconst_j / 2 + sg_j)) — a constant of the word — |e| <= 2 tau - e, hence
// M(pi) <= 2 T tau - sum_t e_pi(t), and the sum of a path's emissions is its score minus its start and forward
// weights: M(pi) <= 2 T tau - score(pi) + |start| + |R| terms.  Both paths score at least best - err(pi_A) exactly
// (pi_E beats pi_A there), which a second evaluation of the formula with M + 2 eps absorbs.  That replaces
// sum_t max_j |e_j(t)| of round 2's tile-per-word kernel — dominated by the worst-matching state of every frame — by
// the magnitude along the paths that matter: smaller intervals, fewer exact lattices, and no magnitude bookkeeping
// in the time loop.
// ---------------------------------------------------------------------------------------
__host__ __device__ constexpr unsigned long long dense_first_mask(int S, int tau) {  // word-start lanes, lane 0 left out
  unsigned long long m = 0;
  for (int r = 1; r < 16; ++r)
    if ((16 * tau + r) % S == 0) m |= 1ull << r;
  return m * 0x0001000100010001ull;
}
__host__ __device__ constexpr int dense_tiles(int S, int WP) { return (WP * S + 15) / 16; }
__device__ __forceinline__ float cnd_f32(float a, float b, unsigned long long mask) {  // mask ? b : a, uniform mask
  float r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
  return r;
}

// ---------------------------------------------------------------------------------------
// P streamed from LDS (the whole vocabulary in ONE pass at any shape).  Fragments kept in registers cap a pass at 7
// tiles for 13 dimensions and at 2 for 39 (24 registers per tile): at (39, 18) every word would repeat the 96-slot
// operand build, the most expensive part of a frame (that variant of this kernel was measured and removed, see
// launch_approx).  Here a workgroup of NW wavefronts gathers the dense fragments of all WP words once into LDS (78 KB
// for 11 x 18 states x 96 slots) and every wavefront walks the tiles of a frame from the top down, two fragment sets
// in flight (ds_read_b128, lane-contiguous: conflict-free), nine MFMAs and one column update per tile: one operand
// build and one feature read per frame, 4 registers of lattice column per tile.  Wavefronts are persistent and take
// 16-utterance tiles in a strided loop; nothing but the read-only LDS tables is shared, no workgroup barrier after
// the prologue.
// ---------------------------------------------------------------------------------------
template <int B, int E, class F>
__device__ __forceinline__ void sfor(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    sfor<B + 1, E>(f);
  }
}
__device__ __forceinline__ void wave_fence_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int D, int S, int WP, int NW>
struct BoundLds {
  static constexpr int KC = gemm_kchunks(D), NT = dense_tiles(S, WP), NG = 16 * NT;
  static constexpr size_t frag = static_cast<size_t>(NT) * KC * 2 * 64 * 16;
  static constexpr size_t tab = frag, tab_bytes = 2 * 4 * KC * 8 * 4;
  static constexpr size_t start = tab + tab_bytes, start_bytes = NG * 4;
  static constexpr size_t fin = start + start_bytes, fin_bytes = static_cast<size_t>(NW) * 4 * NG * 4;
  // kDma: the next frame's feature rows arrive by global_load_lds_dwordx4 in wave-private staging slots
  // [NW][KC][2][64 lanes x 16 bytes] instead of registers.  Measured: it frees 8 KC registers across the tile loop,
  // which is what lets 13 tiles of lattice column (11 words x 18 states) live without spilling — 4.79 ms of decode at
  // (39, 18) against 5.01 in two passes — and costs an LDS round trip at the top of every frame, which makes every
  // shape that fits anyway slower (1.11 -> 1.20 ms at (13, 10)); it is on only where the tiles would spill.
  static constexpr bool kDma = NT * (8 * KC + 4) > 256;
  static constexpr size_t stage = (fin + fin_bytes + 15) / 16 * 16, stage_bytes = kDma ? static_cast<size_t>(NW) * KC * 2 * 1024 : 0;
  static constexpr size_t total = stage + stage_bytes;
};

template <int D, int S, int WP, int NW, int WPE>
__global__ __launch_bounds__(NW *kWave) __attribute__((amdgpu_waves_per_eu(WPE))) void viterbi_bound_lds_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
    int64_t n_utts, int32_t W, const uint4 *__restrict__ gfrag, const float *__restrict__ gctr,
    const double *__restrict__ gkw, const double *__restrict__ gR, const double *__restrict__ log_start,
    const double *__restrict__ log_trans, const double *__restrict__ wconst, const double *__restrict__ hgc,
    double *__restrict__ ascore, double *__restrict__ aeps) {
  using L = BoundLds<D, S, WP, NW>;
  constexpr int G = gemm_groups(D), KC = L::KC, RT = gemm_rtiles(S), NT = L::NT, NG = L::NG, iC = D % 8, G8 = 8 * G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 *s_frag = reinterpret_cast<uint4 *>(smem);
  float *s_fa = reinterpret_cast<float *>(smem + L::tab);      // [4 q][KC][8]
  float *s_ctra = s_fa + 4 * KC * 8;
  float *s_start = reinterpret_cast<float *>(smem + L::start);  // [NG]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave, col = lane & 15, q = lane >> 4;
  float *s_fin = reinterpret_cast<float *>(smem + L::fin) + static_cast<size_t>(wave) * 4 * NG;  // [4 q][NG], wave-private
  unsigned char *s_stage = smem + L::stage + static_cast<size_t>(wave) * KC * 2 * 1024;        // wave-private (kDma)
  constexpr bool kDma = L::kDma;
  const int w0 = static_cast<int>(blockIdx.y) * WP;
  const int nw = W - w0 < WP ? W - w0 : WP;
  const int nt_valid = (nw * S + 15) / 16;
  const double up = gkw[2 * W], down = gkw[2 * W + 1];  // 2^g, 2^-g

  // ---- workgroup prologue: dense B fragments, slot factors, start values -> LDS ----
  for (int idx = tid; idx < NT * KC * 2 * 64; idx += NW * kWave) {
    const int l = idx & 63, p = (idx >> 6) & 1, c = (idx >> 7) % KC, tau = (idx >> 7) / KC;
    const int g = 16 * tau + (l & 15), wl = g / S, j = g - wl * S;
    uint4 x = make_uint4(0u, 0u, 0u, 0u);
    if (wl < nw) x = gfrag[(((static_cast<int64_t>(w0 + wl) * RT + j / 16) * KC + c) * 2 + p) * kWave + (j % 16) + 16 * (l >> 4)];
    s_frag[idx] = x;
  }
  for (int idx = tid; idx < 4 * KC * 8; idx += NW * kWave) {
    const int i = idx & 7, c = (idx >> 3) % KC, qq = (idx >> 3) / KC;
    const int g = 4 * c + qq;
    const int half = g < G ? 0 : (g < 2 * G ? 1 : 2);
    const int gg = half == 2 ? 0 : g - (half == 1 ? G : 0);
    const int f = 8 * gg + i;
    const bool ok = half < 2 && f < D;
    const float a = ok ? gctr[(half == 0 ? G8 : 2 * G8) + f] : 0.0f;
    s_fa[idx] = a;
    s_ctra[idx] = ok ? gctr[f] * a : 0.0f;  // exact: a is a power of two
  }
  for (int g = tid; g < NG; g += NW * kWave) {
    const int wl = g / S, j = g - wl * S;
    s_start[g] = wl < nw ? static_cast<float>((log_start[static_cast<int64_t>(w0 + wl) * S + j] - gR[static_cast<int64_t>(w0 + wl) * S + j]) * up)
                         : -__builtin_huge_valf();
  }
  // per-lane slot geometry of the operand build (as in the kernels above)
  int fbase[KC];
  float onev[KC];
  bool sq[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int g = 4 * c + q;
    const int half = g < G ? 0 : (g < 2 * G ? 1 : 2);
    const int gg = half == 2 ? 0 : g - (half == 1 ? G : 0);
    sq[c] = half == 0;
    onev[c] = (half == 0 && gg == D / 8) ? 1024.0f : 0.0f;
    fbase[c] = 8 * gg;
  }
  unsigned nsbits = 0u;  // bit tau: this lane's state of tile tau has no self-loop (a register, not NT mask pairs)
  bool inner_noself = false;
#pragma unroll
  for (int tau = 0; tau < NT; ++tau) {
    const int g = 16 * tau + col, wl = g / S, j = g - wl * S;
    const bool ns = wl < nw && log_trans[(static_cast<int64_t>(w0 + (wl < nw ? wl : 0)) * S + j) * S + j] == neg_inf();
    nsbits |= ns ? 1u << tau : 0u;
    inner_noself = inner_noself || __ballot(ns && j != 0) != 0ull;
  }
  __syncthreads();  // the only workgroup barrier

  const float ninf = -__builtin_huge_valf(), qnan = __builtin_nanf("");
  const int64_t n_floats = offsets[n_utts] * D;
  const int64_t n_tiles = (n_utts + 15) / 16;
  const uint4 *fr_lane = s_frag + lane;
  const float4 *fa_q = reinterpret_cast<const float4 *>(s_fa + q * KC * 8);
  const float4 *ctra_q = reinterpret_cast<const float4 *>(s_ctra + q * KC * 8);

  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * NW + wave; tile < n_tiles; tile += static_cast<int64_t>(gridDim.x) * NW) {
    const int64_t slot = tile * 16 + col;
    const bool live = slot < n_utts;
    const int u = live ? (order ? order[slot] : static_cast<int>(slot)) : 0;
    const int64_t beg = live ? offsets[u] : 0;
    const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
    const int Tw = wave_max_i32(T);
    const int Tmin = -wave_max_i32(-T);
    int Ti[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Ti[i] = __shfl(T, 4 * q + i);
    float v[NT][4];
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) {
      const float sv = s_start[16 * tau + col];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[tau][i] = sv;
    }
    float bigsum = 0.0f;

    // A lane reads the eight consecutive floats of its group with two 16-byte loads; slots past the frame's D values
    // (the next frame's data) are multiplied by zero.  Only the tile that holds the LAST frame of the feature buffer
    // can run past its end: that one takes guarded single loads (a wavefront-uniform choice, made once per tile).
    float xr[KC][8];
    const float *pc[KC];
    bool tile_safe = true;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      pc[c] = feats + beg * D + fbase[c];
      tile_safe = tile_safe && (beg + (T > 0 ? T - 1 : 0)) * D + fbase[c] + 8 <= n_floats;  // idle lanes read frame 0 of the buffer
    }
    tile_safe = __all(tile_safe);
    auto load = [&](int t) {
      const int tt = t < T ? t : (T > 0 ? T - 1 : 0);
      if (tile_safe) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const float *p = pc[c] + tt * D;
          const FeatQuad v0 = *reinterpret_cast<const FeatQuad *>(p);
          const FeatQuad v1 = *reinterpret_cast<const FeatQuad *>(p + 4);
          xr[c][0] = v0.a, xr[c][1] = v0.b, xr[c][2] = v0.c, xr[c][3] = v0.d;
          xr[c][4] = v1.a, xr[c][5] = v1.b, xr[c][6] = v1.c, xr[c][7] = v1.d;
        }
      } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const float *p = pc[c] + tt * D;
#pragma unroll
          for (int i = 0; i < 8; ++i) xr[c][i] = (T > 0 && fbase[c] + i < D) ? p[i] : 0.0f;
        }
      }
    };
    // kDma: the same rows one frame ahead straight into the wavefront's LDS staging slots (global_load_lds_dwordx4:
    // lane l's 16 bytes land at base + 16 l, scripts/ubench/lds_dma_probe.hip), issued right after the current frame's
    // copy has been read out; the unsafe tile goes through registers and ds_write
    auto prefetch = [&](int t) {
      const int tt = t < T ? t : (T > 0 ? T - 1 : 0);
      if (tile_safe) {
        if (T > 0) {  // (idle lanes fetch nothing: their slots keep whatever they held, their rows are never read out)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const float *p = pc[c] + tt * D;
          __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)(s_stage + (2 * c) * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(p + 4, (__attribute__((address_space(3))) void *)(s_stage + (2 * c + 1) * 1024), 16, 0, 0);
        }
        }
      } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const float *p = pc[c] + tt * D;
          float4 lo4, hi4;
          lo4.x = (T > 0 && fbase[c] + 0 < D) ? p[0] : 0.0f;
          lo4.y = (T > 0 && fbase[c] + 1 < D) ? p[1] : 0.0f;
          lo4.z = (T > 0 && fbase[c] + 2 < D) ? p[2] : 0.0f;
          lo4.w = (T > 0 && fbase[c] + 3 < D) ? p[3] : 0.0f;
          hi4.x = (T > 0 && fbase[c] + 4 < D) ? p[4] : 0.0f;
          hi4.y = (T > 0 && fbase[c] + 5 < D) ? p[5] : 0.0f;
          hi4.z = (T > 0 && fbase[c] + 6 < D) ? p[6] : 0.0f;
          hi4.w = (T > 0 && fbase[c] + 7 < D) ? p[7] : 0.0f;
          reinterpret_cast<float4 *>(s_stage + (2 * c) * 1024)[lane] = lo4;
          reinterpret_cast<float4 *>(s_stage + (2 * c + 1) * 1024)[lane] = hi4;
        }
      }
    };
    auto step = [&](auto first_c, auto generic_c, auto uniform_c, int t) {
      constexpr bool first = decltype(first_c)::value, generic = decltype(generic_c)::value,
                     uniform = decltype(uniform_c)::value;
      if constexpr (kDma) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the previous frame's prefetch has landed
        wave_fence_lds();
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const float4 a4 = reinterpret_cast<const float4 *>(s_stage + (2 * c) * 1024)[lane];
          const float4 b4 = reinterpret_cast<const float4 *>(s_stage + (2 * c + 1) * 1024)[lane];
          xr[c][0] = a4.x, xr[c][1] = a4.y, xr[c][2] = a4.z, xr[c][3] = a4.w;
          xr[c][4] = b4.x, xr[c][5] = b4.y, xr[c][6] = b4.z, xr[c][7] = b4.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wave_fence_lds();
        if (t + 1 < Tw) prefetch(t + 1);
      }
      u32x4 fr[2][KC][2];
      auto fetch = [&](auto tau_c, auto buf_c) {
        constexpr int tau = decltype(tau_c)::value, buf = decltype(buf_c)::value;
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const uint4 x = fr_lane[((tau * KC + c) * 2 + p) * 64];
            fr[buf][c][p] = u32x4{x.x, x.y, x.z, x.w};
          }
      };
      u32x4 bh[KC], bl[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float4 fa0 = fa_q[2 * c], fa1 = fa_q[2 * c + 1], ct0 = ctra_q[2 * c], ct1 = ctra_q[2 * c + 1];
        const float fa[8] = {fa0.x, fa0.y, fa0.z, fa0.w, fa1.x, fa1.y, fa1.z, fa1.w};
        const float ct[8] = {ct0.x, ct0.y, ct0.z, ct0.w, ct1.x, ct1.y, ct1.z, ct1.w};
        float ph[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xv = __builtin_fmaf(xr[c][i], fa[i], -ct[i]);
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
          asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(h2));
          asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(b) : "v"(h2));
          bh[c][e] = h2;
          bl[c][e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // (the two fragment sets are fetched here, not under the operand build: 48 registers the build needs)
      fetch(std::integral_constant<int, NT - 1>{}, std::integral_constant<int, (NT - 1) & 1>{});
      if constexpr (NT > 1) fetch(std::integral_constant<int, NT - 2>{}, std::integral_constant<int, (NT - 2) & 1>{});
      if constexpr (!kDma) {
        if (t + 1 < Tw) load(t + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      // tiles from the top down: the predecessor reads of tile tau (its own lane - 1, lane 15 of tile tau - 1) are
      // still frame t - 1 when it is updated.  Software pipeline: the MFMA chain of tile tau - 1 is issued BEFORE the
      // column update of tile tau, whose vector work then runs under it, and the fragments of tile tau - 2 are fetched
      // into the buffer tile tau's chain has just consumed — two fragment sets and two accumulators in flight.
      f32x4 accs[2];
      auto chain = [&](auto tau_c) {
        constexpr int tau = decltype(tau_c)::value, buf = tau & 1;
        if (tau < nt_valid) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < KC; ++c) {
            acc = mfma_f16(bl[c], fr[buf][c][0], acc);
            acc = mfma_f16(bh[c], fr[buf][c][1], acc);
          }
#pragma unroll
          for (int c = 0; c < KC; ++c) acc = mfma_f16(bh[c], fr[buf][c][0], acc);
          accs[buf] = acc;
        }
      };
      chain(std::integral_constant<int, NT - 1>{});
      sfor<0, NT>([&](auto k_c) {
        constexpr int tau = NT - 1 - decltype(k_c)::value, buf = tau & 1;
        if constexpr (tau > 0) chain(std::integral_constant<int, tau - 1>{});
        if constexpr (tau > 1 && !(SAPR_MFMA_ABL & 8)) fetch(std::integral_constant<int, tau - 2>{}, std::integral_constant<int, tau & 1>{});
        if (tau < nt_valid) {
          const f32x4 acc = accs[buf];
          if constexpr (SAPR_MFMA_ABL & 2) {  // timing ablation (wrong results): no column update
#pragma unroll
            for (int i = 0; i < 4; ++i) v[tau][i] += acc[i];
          } else {
          constexpr unsigned long long fm = dense_first_mask(S, tau);
          constexpr bool head0 = (16 * tau) % S == 0;
          float nv[4];
          if constexpr (first) {
#pragma unroll
            for (int i = 0; i < 4; ++i) nv[i] = v[tau][i] + acc[i];
          } else {
            int wrap[4];
            float pred[4], self[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              wrap[i] = __float_as_int(ninf);
              if constexpr (tau > 0 && !head0)
                wrap[i] = __builtin_amdgcn_mov_dpp(__float_as_int(v[tau > 0 ? tau - 1 : 0][i]), 0x121, 0xf, 0xf, true);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
              pred[i] = __int_as_float(__builtin_amdgcn_update_dpp(wrap[i], __float_as_int(v[tau][i]), 0x111, 0xf, 0xf, false));
            if constexpr (fm != 0ull) {
#pragma unroll
              for (int i = 0; i < 4; ++i) pred[i] = cnd_f32(pred[i], ninf, fm);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              self[i] = v[tau][i];
              if constexpr (generic) self[i] = (nsbits >> tau) & 1u ? qnan : self[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pred[i] = max_drop_nan(pred[i], self[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) nv[i] = pred[i] + acc[i];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (uniform)
              v[tau][i] = nv[i];
            else
              v[tau][i] = t < Ti[i] ? nv[i] : v[tau][i];
          }
          }
        }
        // keep the fragment reads two tiles deep: left alone, the scheduler hoists all NT tiles' reads to the top of
        // the frame and spills what they return
        __builtin_amdgcn_sched_barrier(0);
      });
    };

    if constexpr (kDma) {
      if (Tw > 0) prefetch(0);
    } else {
      load(0);
    }
    if (Tw > 0) step(std::true_type{}, std::false_type{}, std::false_type{}, 0);
    int t = 1;
    if (Tw > 1) {
      step(std::false_type{}, std::true_type{}, std::false_type{}, 1);
      t = 2;
    }
    if (inner_noself) {
      for (; t < Tw; ++t) step(std::false_type{}, std::true_type{}, std::false_type{}, t);
    } else {
      for (; t < Tmin; ++t) step(std::false_type{}, std::false_type{}, std::true_type{}, t);
      for (; t < Tw; ++t) step(std::false_type{}, std::false_type{}, std::false_type{}, t);
    }

    // ---- scores and intervals: lane (row q, word lane % 16) takes the row's four utterances in turn ----
    double phi_sum = static_cast<double>(bigsum);
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) phi_sum += __shfl_xor(phi_sum, off);
    constexpr double u32 = 5.9604644775390625e-08, u64 = 1.1102230246251565e-16;
    constexpr double cacc = 36.0 + 68.0 * KC;
    const int wl = col;
    const bool mine = wl < nw;
    const int w = w0 + (mine ? wl : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double phik = __shfl(phi_sum, 4 * q + i);
      const int ui = __shfl(u, 4 * q + i);
      const bool livei = __shfl(live ? 1 : 0, 4 * q + i) != 0;
      wave_fence_lds();
#pragma unroll
      for (int tau = 0; tau < NT; ++tau) s_fin[q * NG + 16 * tau + col] = v[tau][i];
      wave_fence_lds();
      if (mine && livei) {
        double best = neg_inf(), tau_l = 0.0;
        for (int j = 0; j < S; ++j) {
          double sg = log_trans[(static_cast<int64_t>(w) * S + j) * S + j];
          if (sg == neg_inf()) sg = 0.0;
          const double top = hgc[static_cast<int64_t>(w) * S + j] + sg;  // an emission never exceeds -gconst / 2 + sg
          tau_l = (top > tau_l || top != top) ? top : tau_l;
          // back to u = v + R; the unreachable tail states (R = -inf there) drop out
          const double d = static_cast<double>(s_fin[q * NG + wl * S + j]) * down + gR[(static_cast<int64_t>(W) + w) * S + j] - sg;
          best = (d > best || d != d) ? d : best;
        }
        const double *wc4 = wconst + static_cast<int64_t>(w) * 4;
        const double lts = wc4[2], lss = wc4[3], Td = static_cast<double>(Ti[i]);
        const double m_raw = 2.0 * Td * tau_l - best + lss + 7.0 * lts;
        const double m0 = (m_raw > 0.0 || m_raw != m_raw) ? m_raw : 0.0;
        auto interval = [&](double mm) {
          const double span = 3.0 * mm + Td * gkw[w];
          const double e32 = cacc * u32 * 1.001 * span + 0x1p-14 * 1.01 * (Td * gkw[W + w] + 8.0 * phik) * down;
          const double e_lat = u32 * 1.01 * (Td + 1.0) * (mm + 2.0 * lts + lss);
          const double e64 = (8.0 * Td + 16.0) * u64 * (span + Td * lts + lss);
          return 2.0 * (e32 + e_lat + e64) + Td * 1e-14 + 1e-30;
        };
        const double eps0 = interval(m0);
        const double eps = interval(m0 + 2.0 * eps0);
        ascore[static_cast<int64_t>(ui) * W + w] = Ti[i] > 0 ? best : neg_inf();
        aeps[static_cast<int64_t>(ui) * W + w] = Ti[i] > 0 ? eps : 0.0;
      }
    }
    wave_fence_lds();
  }
}

template <int D, int S, int WP, int NW, int WPE>
int launch_bound_lds(const ScoreArgs &a, const PackView &pv, double *ascore, double *aeps) {
  using L = BoundLds<D, S, WP, NW>;
  static_assert(WP <= 16, "one finishing lane per word of a pass");
  const auto kern = viterbi_bound_lds_kernel<D, S, WP, NW, WPE>;
  // per launch, not once: the attribute belongs to the current device's copy of the kernel
  SAPR_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(L::total)));
  int dev = 0, cus = 256;
  SAPR_HIP_TRY(hipGetDevice(&dev));
  SAPR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int64_t n_tiles = (a.n_utts + 15) / 16;
  const int per_cu = std::max(1, std::min(static_cast<int>((160 * 1024) / L::total), (4 * WPE) / NW));
  int64_t gx = std::min<int64_t>((n_tiles + NW - 1) / NW, static_cast<int64_t>(cus) * per_cu);
  if (gx < 1) gx = 1;
  const int n_pass = (a.W + WP - 1) / WP;
  SAPR_LAUNCH((viterbi_bound_lds_kernel<D, S, WP, NW, WPE>), dim3(static_cast<unsigned>(gx), static_cast<unsigned>(n_pass)),
              dim3(NW * kWave), L::total, a.stream, a.feats, a.offsets, a.order, a.n_utts, a.W, pv.gfrag, pv.gctr,
              pv.gkw, pv.gR, pv.log_start, pv.log_trans, pv.wconst, pv.hgc, ascore, aeps);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

template <int D, int S>
int launch_approx(const ScoreArgs &a, const PackView &pv, double *ascore, double *aeps, int pack_flags) {
  if constexpr (S <= 32) {
    if (pack_flags & SAPR_PACK_GEMM_OK) {  // callers clear the bit to keep pass A on the vector ALU
      // P streamed from LDS, the vocabulary in one pass.  Measured on MI355X, 100 000 utterances x 11 words, whole
      // pruned decode [the same layout with register-resident fragments and as many words per pass as 256 registers
      // hold, removed since]: (13, 10) 1.09 ms [1.13], (13, 18) 1.80 [1.85], (39, 10) 2.88 [3.89], (39, 18) 4.79 with
      // the feature rows prefetched into LDS (BoundLds::kDma; 5.61 with a register prefetch, whose 13 tiles of lattice
      // column spill; 5.01 in two passes of 6 + 5 words: developer switch SAPR_BOUND_WC=6) [7.98].
      const char *env = std::getenv("SAPR_BOUND_WC");
      const int want = env ? std::atoi(env) : 0;
      if constexpr (D <= 16 && S <= 16) return launch_bound_lds<D, S, 11, 4, 3>(a, pv, ascore, aeps);
      else if constexpr (D <= 16) return launch_bound_lds<D, S, 11, 4, 2>(a, pv, ascore, aeps);
      else if constexpr (S <= 16) return launch_bound_lds<D, S, 11, 8, 2>(a, pv, ascore, aeps);
      else {
        if (want == 6) return launch_bound_lds<D, S, 6, 8, 2>(a, pv, ascore, aeps);
        return launch_bound_lds<D, S, 11, 8, 2>(a, pv, ascore, aeps);
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

}  // namespace

int launch_approx_13_10(const ScoreArgs &a, const emission::PackView &pv, double *ascore, double *aeps, int pack_flags) {
  return launch_approx<13, 10>(a, pv, ascore, aeps, pack_flags);
}
int launch_approx_13_18(const ScoreArgs &a, const emission::PackView &pv, double *ascore, double *aeps, int pack_flags) {
  return launch_approx<13, 18>(a, pv, ascore, aeps, pack_flags);
}
int launch_approx_39_10(const ScoreArgs &a, const emission::PackView &pv, double *ascore, double *aeps, int pack_flags) {
  return launch_approx<39, 10>(a, pv, ascore, aeps, pack_flags);
}
int launch_approx_39_18(const ScoreArgs &a, const emission::PackView &pv, double *ascore, double *aeps, int pack_flags) {
  return launch_approx<39, 18>(a, pv, ascore, aeps, pack_flags);
}
}  // namespace sapr
