// Bounding passes of the pruned decoder (pass A of sapr_viterbi_decode_pruned): float32 emission sums on the vector
// ALU or on the matrix cores, with a rigorous interval around every word's exact Viterbi score.
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
// turned into a NaN, which v_max_f32 drops (one v_cndmask_b32 for position 0 of each quarter; a model with such a
// state elsewhere in the chain is bounded by the kernel above).
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

// wavefronts per SIMD the register allocation aims at: the time loop is one dependent chain per wavefront (operand
// build -> MFMA -> column update -> next frame, with a feature load to wait for), so few words per wavefront at a
// higher occupancy can beat many words at two wavefronts per SIMD
__host__ __device__ constexpr int approx_wpe(int D, int S, int WC) {
  const int per_word = 4 * gemm_rtiles(S) * (2 * gemm_kchunks(D) + 2) + 1;  // fragments, column, accumulators
  const int regs = per_word * WC + 24 * gemm_kchunks(D) + 40;
  return regs <= 120 ? 4 : (regs <= 160 ? 3 : 2);
}
template <int D, int S, int WC>
// an explicit waves_per_eu also makes the compiler put the MFMA results in VGPRs (no v_accvgpr_read per use)
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(approx_wpe(D, S, WC)))) void viterbi_approx_mfma_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
    int64_t n_utts, int32_t W, const uint4 *__restrict__ gfrag, const float *__restrict__ gctr,
    const double *__restrict__ gkw, const double *__restrict__ gR, const double *__restrict__ log_start,
    const double *__restrict__ log_trans, const double *__restrict__ wconst, double *__restrict__ ascore,
    double *__restrict__ aeps) {
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
  // The lattice column runs in FLOAT32 with the forward weights divided out (round 3; float64 with a weight per
  // state before): v[j] = u[j] - R_j, R_j = sum_(i<=j) r_i (sapr_diag_pack, PackView::gR), turns
  // u[j] = max(u[j-1] + r_j, u[j]) + e_j into v[j] = max(v[j-1], v[j]) + e_j — max, add per state and frame at the
  // float32 rate, no weight registers, no float32 -> float64 conversion of the MFMA results.  That is what lets ONE
  // wavefront pass hold the whole 11-word vocabulary at (13, 10): one operand build and one feature read per frame.
  // The roundings enter the interval as e_lat below.
  float uu[WC][NS];
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
        // units of 2^-g; -inf where the state cannot start a path (and for the rows past S)
        uu[wc][rt * 4 + i] = j < S ? static_cast<float>((ls[j] - gR[static_cast<int64_t>(w) * S + j]) * up)
                                   : -__builtin_huge_valf();
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
    // lattice value of the state just below this lane's first one, per row tile: state 16 rt + 4 q - 1 lives in
    // lane - 16 (same tile, position 3) or, for q == 0 and rt > 0, in lane + 48 of the tile below
    float p3[WC][RT];
    if constexpr (!first) {
#pragma unroll
      for (int wc = 0; wc < WC; ++wc)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          p3[wc][rt] = __shfl_up(uu[wc][rt * 4 + 3], 16);
          if (rt == 0) p3[wc][rt] = q == 0 ? -__builtin_huge_valf() : p3[wc][rt];  // state 0 has no predecessor
          if constexpr (RT > 1) {
            if (rt > 0) {
              const float wrap = __shfl(uu[wc][(rt - 1) * 4 + 3], (lane + 48) & 63);
              p3[wc][rt] = q == 0 ? wrap : p3[wc][rt];
            }
          }
        }
    }
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
            for (int i = 0; i < 4; ++i) uu[wc][rt * 4 + i] += a[i];
          } else {
#pragma unroll
            for (int i = 3; i >= 0; --i) {
              const int k = rt * 4 + i;
              const float pred = i == 0 ? p3[wc][rt] : uu[wc][k - 1];
              const float self = i == 0 ? nan_where(uu[wc][k], noself0[wc][rt]) : uu[wc][k];
              uu[wc][k] = max_drop_nan(pred, self) + a[i];
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
      // back to u = v + R; the unreachable tail states (R = -inf there) drop out
      const double d = static_cast<double>(uu[wc][k]) * down +
                       (j < S ? gR[(static_cast<int64_t>(W) + w) * S + j] : neg_inf()) - sg;
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
      // float32 lattice: along a path every frame rounds once (+ emission; the max is exact), within 2^-24 of a
      // magnitude <= |log_start| + |R| + sum_t max|emission| <= lss + 2 lts + m; the start value was rounded once
      const double e_lat = u32 * 1.01 * (Td + 1.0) * (m + 2.0 * lts + lss);
      const double e64 = (8.0 * Td + 16.0) * u64 * (span + Td * lts + lss);
      ascore[u * W + w] = T > 0 ? best : neg_inf();
      aeps[u * W + w] = T > 0 ? 2.0 * (e32 + e_lat + e64) + Td * 1e-14 + 1e-30 : 0.0;
    }
  }
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
// lattice, and pi_E, the best path of the exact one.  With tau >= max(0, every computed emission of the utterance)
// (one v_max3_i32 per two tiles and frame), |e| <= 2 tau - e, hence M(pi) <= 2 T tau - sum_t e_pi(t), and the sum of
// a path's emissions is its lattice value minus its start value: M(pi_A) <= 2 T tau - best + |start| + |R| terms;
// pi_E's computed value is within err(pi_A) + err(pi_E) of best (it beats pi_A exactly), which a second evaluation
// of the formula with M + 2 eps absorbs.  That replaces sum_t max_j |e_j(t)| of the tile-per-word kernel — dominated
// by the worst-matching state of every frame — by the magnitude along the paths that matter: smaller intervals,
// fewer exact lattices, and no magnitude bookkeeping in the time loop.
// ---------------------------------------------------------------------------------------
__host__ __device__ constexpr unsigned long long dense_first_mask(int S, int tau) {  // word-start lanes, lane 0 left out
  unsigned long long m = 0;
  for (int r = 1; r < 16; ++r)
    if ((16 * tau + r) % S == 0) m |= 1ull << r;
  return m * 0x0001000100010001ull;
}
__host__ __device__ constexpr int dense_tiles(int S, int WP) { return (WP * S + 15) / 16; }
__host__ __device__ constexpr int dense_wpe(int D, int S, int WP) {
  const int regs = dense_tiles(S, WP) * (8 * gemm_kchunks(D) + 8) + 32 * gemm_kchunks(D) + 40;
  return regs <= 120 ? 4 : (regs <= 160 ? 3 : 2);
}
__device__ __forceinline__ float cnd_f32(float a, float b, unsigned long long mask) {  // mask ? b : a, uniform mask
  float r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
  return r;
}

template <int D, int S, int WP>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(dense_wpe(D, S, WP)))) void viterbi_bound_dense_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets, const int32_t *__restrict__ order,
    int64_t n_utts, int32_t W, const uint4 *__restrict__ gfrag, const float *__restrict__ gctr,
    const double *__restrict__ gkw, const double *__restrict__ gR, const double *__restrict__ log_start,
    const double *__restrict__ log_trans, const double *__restrict__ wconst, double *__restrict__ ascore,
    double *__restrict__ aeps) {
  constexpr int G = gemm_groups(D), KC = gemm_kchunks(D), RT = gemm_rtiles(S), NT = dense_tiles(S, WP), iC = D % 8;
  constexpr int NG = 16 * NT;
  __shared__ double s_d[16][NG + 1];
  __shared__ double s_phi[16];
  __shared__ float s_tau[16];
  const int lane = threadIdx.x, col = lane & 15, q = lane >> 4;
  const int n_pass = (W + WP - 1) / WP;
  const int64_t tile = blockIdx.x / n_pass;
  const int w0 = static_cast<int>(blockIdx.x - tile * n_pass) * WP;
  const int nw = W - w0 < WP ? W - w0 : WP;
  // operand-build side: this lane's utterance is `col`
  const int64_t slot = tile * 16 + col;
  const bool live = slot < n_utts;
  const int64_t u = live ? (order ? static_cast<int64_t>(order[slot]) : slot) : 0;
  const int64_t beg = live ? offsets[u] : 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
  const int Tw = wave_max_i32(T);
  const int Tmin = -wave_max_i32(-T);
  const int64_t n_floats = offsets[n_utts] * D;
  // lattice side: this lane's state rows, for the utterances 4 q + i
  int Ti[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) Ti[i] = __shfl(T, 4 * q + i);

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
  float bigsum = 0.0f;

  // B fragments of this pass's rows, gathered from the per-word fragments sapr_diag_pack stores (row j % 16 of row
  // tile j / 16 of word w, same k group); rows past the pass's words are zero
  u32x4 bfr[NT][KC][2];
  float v[NT][4];
  unsigned long long noself[NT];
  bool inner_noself = false;
#pragma unroll
  for (int tau = 0; tau < NT; ++tau) {
    const int g = 16 * tau + col, wl = g / S, j = g - wl * S;
    const bool valid = wl < nw;
    const int w = w0 + (valid ? wl : 0);
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const uint4 x = gfrag[(((static_cast<int64_t>(w) * RT + j / 16) * KC + c) * 2 + p) * kWave + (j % 16) + 16 * q];
        bfr[tau][c][p] = valid ? u32x4{x.x, x.y, x.z, x.w} : u32x4{0u, 0u, 0u, 0u};
      }
    const float sv = valid ? static_cast<float>((log_start[static_cast<int64_t>(w) * S + j] - gR[static_cast<int64_t>(w) * S + j]) * up)
                           : -__builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < 4; ++i) v[tau][i] = sv;
    const bool ns = valid && log_trans[(static_cast<int64_t>(w) * S + j) * S + j] == neg_inf();
    noself[tau] = __ballot(ns);
    inner_noself = inner_noself || __ballot(ns && j != 0) != 0ull;
  }
  int runmax[4] = {0, 0, 0, 0};  // tau: max(+0, every computed emission) of utterance 4 q + i over this lane's rows, as bits
  const float ninf = -__builtin_huge_valf(), qnan = __builtin_nanf("");

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

  // FIRST: frame 0 (start + emission); GENERIC: states without a self-loop are honoured in this frame; UNIFORM: every
  // utterance of the wavefront has frame t
  auto step = [&](auto first_c, auto generic_c, auto uniform_c, int t) {
    constexpr bool first = decltype(first_c)::value, generic = decltype(generic_c)::value,
                   uniform = decltype(uniform_c)::value;
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
        // residual of the truncated half, exact in float32: one v_fma_mix_f32 each (half source, float addend)
        asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(h2));
        asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(b) : "v"(h2));
        bh[c][e] = h2;
        bl[c][e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
      }
    }
    if (t + 1 < Tw) load(t + 1);
    f32x4 acc[NT];
#pragma unroll
    for (int tau = 0; tau < NT; ++tau) acc[tau] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KC; ++c) {  // small products first: phi_lo P_hi, phi_hi P_lo, then phi_hi P_hi
#pragma unroll
      for (int tau = 0; tau < NT; ++tau) acc[tau] = mfma_f16(bl[c], bfr[tau][c][0], acc[tau]);
#pragma unroll
      for (int tau = 0; tau < NT; ++tau) acc[tau] = mfma_f16(bh[c], bfr[tau][c][1], acc[tau]);
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int tau = NT - 1; tau >= 0; --tau) acc[tau] = mfma_f16(bh[c], bfr[tau][c][0], acc[tau]);  // updated first, done first
    // descending tiles: the predecessor reads (this tile's lane - 1, the tile below's lane 15) are still frame t - 1.
    // The four utterances of a tile are four independent chains (fetch, gate, max, add); they are written stage by
    // stage so that the compiler interleaves them instead of running one dependent chain through one register.
#pragma unroll
    for (int tau = NT - 1; tau >= 0; --tau) {
      constexpr unsigned long long kNoMask = 0ull;
      const unsigned long long fm = dense_first_mask(S, tau);
      const bool head0 = (16 * tau) % S == 0;  // lane 0 of the tile starts a word
      float nv[4];
      if constexpr (first) {
#pragma unroll
        for (int i = 0; i < 4; ++i) nv[i] = v[tau][i] + acc[tau][i];
      } else {
        int wrap[4];
        float pred[4], self[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          wrap[i] = __float_as_int(ninf);
          if (tau > 0 && !head0)
            wrap[i] = __builtin_amdgcn_mov_dpp(__float_as_int(v[tau > 0 ? tau - 1 : 0][i]), 0x121, 0xf, 0xf, true);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          pred[i] = __int_as_float(__builtin_amdgcn_update_dpp(wrap[i], __float_as_int(v[tau][i]), 0x111, 0xf, 0xf, false));
        if (fm != kNoMask) {
#pragma unroll
          for (int i = 0; i < 4; ++i) pred[i] = cnd_f32(pred[i], ninf, fm);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          self[i] = v[tau][i];
          if constexpr (generic) self[i] = cnd_f32(self[i], qnan, noself[tau]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) pred[i] = max_drop_nan(pred[i], self[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) nv[i] = pred[i] + acc[tau][i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (uniform)
          v[tau][i] = nv[i];
        else
          v[tau][i] = t < Ti[i] ? nv[i] : v[tau][i];
      }
    }
    // tau as an INTEGER maximum of the bit patterns: runmax >= +0, a negative float is a negative integer, positive
    // floats order like their patterns (v_max3_i32; a positive NaN wins and poisons eps, which keeps the word).  Not
    // inline assembly: the compiler must see these reads of the MFMA results to place their wait states.
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int tau = 0; tau < NT; ++tau) {
        const int b = __float_as_int(acc[tau][i]);
        runmax[i] = runmax[i] > b ? runmax[i] : b;
      }
  };

  load(0);
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

  // ---- final values -> LDS, then one lane per (utterance, word) ----
#pragma unroll
  for (int tau = 0; tau < NT; ++tau) {
    const int g = 16 * tau + col, wl = g / S, j = g - wl * S;
    const bool valid = wl < nw;
    const int w = w0 + (valid ? wl : 0);
    double sg = log_trans[(static_cast<int64_t>(w) * S + j) * S + j];
    if (sg == neg_inf()) sg = 0.0;
    const double rf = valid ? gR[(static_cast<int64_t>(W) + w) * S + j] : neg_inf();  // -inf: unreachable tail
#pragma unroll
    for (int i = 0; i < 4; ++i) s_d[4 * q + i][g] = static_cast<double>(v[tau][i]) * down + rf - sg;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = runmax[i];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int o = __shfl_xor(m, off);
      m = m > o ? m : o;
    }
    if (col == 0) s_tau[4 * q + i] = __int_as_float(m);
  }
  double phi_sum = static_cast<double>(bigsum);
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) phi_sum += __shfl_xor(phi_sum, off);
  if (q == 0) s_phi[col] = phi_sum;
  __syncthreads();

  constexpr double u32 = 5.9604644775390625e-08, u64 = 1.1102230246251565e-16;
  constexpr double cacc = 36.0 + 68.0 * KC;
  for (int p = lane; p < 16 * WP; p += kWave) {
    const int k = p & 15, wl = p >> 4;
    const int Tk = __shfl(T, k);  // (p & 15 == lane & 15: the value is the lane's own; kept as a shuffle for clarity)
    const int64_t uk = u;
    if (wl >= nw || !live) continue;
    const int w = w0 + wl;
    double best = neg_inf();
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const double d = s_d[k][wl * S + j];
      best = (d > best || d != d) ? d : best;
    }
    const double *wc4 = wconst + static_cast<int64_t>(w) * 4;
    const double lts = wc4[2], lss = wc4[3], Td = static_cast<double>(Tk);
    const double tau_l = static_cast<double>(s_tau[k]) * down;
    // M(pi_A) <= 2 T tau - (best - start - R terms); |start| <= lss, |R_j|, |R_j - sg_j| spreads <= a few lts
    const double m_raw = 2.0 * Td * tau_l - best + lss + 7.0 * lts;
    const double m0 = (m_raw > 0.0 || m_raw != m_raw) ? m_raw : 0.0;  // a NaN score must reach eps (fmax would drop it)
    auto interval = [&](double m) {
      const double span = 3.0 * m + Td * gkw[w];
      const double e32 = cacc * u32 * 1.001 * span + 0x1p-14 * 1.01 * (Td * gkw[W + w] + 8.0 * s_phi[k]) * down;
      const double e_lat = u32 * 1.01 * (Td + 1.0) * (m + 2.0 * lts + lss);
      const double e64 = (8.0 * Td + 16.0) * u64 * (span + Td * lts + lss);
      return 2.0 * (e32 + e_lat + e64) + Td * 1e-14 + 1e-30;
    };
    const double eps0 = interval(m0);
    const double eps = interval(m0 + 2.0 * eps0);  // pi_E's computed value lies within err(pi_A) + err(pi_E) of best
    ascore[uk * W + w] = Tk > 0 ? best : neg_inf();
    aeps[uk * W + w] = Tk > 0 ? eps : 0.0;
  }
}

template <int D, int S, int WP>
int launch_bound_dense(const ScoreArgs &a, const PackView &pv, double *ascore, double *aeps) {
  const int64_t blocks = (a.n_utts + 15) / 16 * ((a.W + WP - 1) / WP);
  if (blocks > 0x7fffffffLL) return fail(SAPR_ERR_ARG, "grid too large (%lld blocks)", (long long)blocks);
  SAPR_LAUNCH((viterbi_bound_dense_kernel<D, S, WP>), dim3(static_cast<unsigned>(blocks)), dim3(kWave), 0, a.stream,
              a.feats, a.offsets, a.order, a.n_utts, a.W, pv.gfrag, pv.gctr, pv.gkw, pv.gR, pv.log_start,
              pv.log_trans, pv.wconst, ascore, aeps);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

template <int D, int S, int WC>
int launch_approx_mfma(const ScoreArgs &a, const PackView &pv, double *ascore, double *aeps) {
  const int64_t blocks = (a.n_utts + 15) / 16 * ((a.W + WC - 1) / WC);
  if (blocks > 0x7fffffffLL) return fail(SAPR_ERR_ARG, "grid too large (%lld blocks)", (long long)blocks);
  SAPR_LAUNCH((viterbi_approx_mfma_kernel<D, S, WC>), dim3(static_cast<unsigned>(blocks)), dim3(kWave), 0, a.stream,
              a.feats, a.offsets, a.order, a.n_utts, a.W, pv.gfrag, pv.gctr, pv.gkw, pv.gR, pv.log_start,
              pv.log_trans, pv.wconst, ascore, aeps);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

template <int D, int S>
int launch_approx(const ScoreArgs &a, const PackView &pv, double *ascore, double *aeps, int pack_flags) {
  if constexpr (S <= 32) {
    if (pack_flags & SAPR_PACK_GEMM_OK) {  // callers clear the bit to keep pass A on the vector ALU
      // words per wavefront pass: as many as 256 registers hold (two wavefronts per SIMD); the operand build and
      // the feature read are shared by the words of a pass
      // SAPR_BOUND_WC (developer switch): force another instantiated word count
      const char *env = std::getenv("SAPR_BOUND_WC");
      const int want = env ? std::atoi(env) : 0;
      // SAPR_BOUND_LAYOUT=tile (developer switch): the tile-per-word kernel of round 2; default: the dense layout,
      // WP words per wavefront pass (SAPR_BOUND_WC picks the other instantiated count)
      const char *lay = std::getenv("SAPR_BOUND_LAYOUT");
      if (!(lay && lay[0] == 't')) {
        if constexpr (D <= 16 && S <= 16) {
          if (want && want <= 6) return launch_bound_dense<D, S, 6>(a, pv, ascore, aeps);
          if (want == 8) return launch_bound_dense<D, S, 8>(a, pv, ascore, aeps);
          return launch_bound_dense<D, S, 11>(a, pv, ascore, aeps);
        } else if constexpr (D <= 16) {
          if (want && want <= 3) return launch_bound_dense<D, S, 3>(a, pv, ascore, aeps);
          return launch_bound_dense<D, S, 7>(a, pv, ascore, aeps);
        } else if constexpr (S <= 16) {
          if (want == 1) return launch_bound_dense<D, S, 1>(a, pv, ascore, aeps);
          return launch_bound_dense<D, S, 3>(a, pv, ascore, aeps);
        } else {
          // measured at (39, 18), 100 000 x 11 words: one word per pass (2 tiles, 255 registers, no spill) 7.9 ms of
          // decode, two words (3 tiles, 52 dwords of scratch) 8.9
          if (want >= 2) return launch_bound_dense<D, S, 2>(a, pv, ascore, aeps);
          return launch_bound_dense<D, S, 1>(a, pv, ascore, aeps);
        }
      }
      // Measured on MI355X at (13, 10), 100 000 utterances x 11 words (round 3, float32 weight-free lattice):
      // 3 words / 4 wavefronts per SIMD 0.79 ms, 4 / 3 0.71, 6 / 2 0.75, 11 / 2 (spilling) 0.74 — the pass is bound by
      // its MFMA + column-update instruction count (33 MFMAs and ~210 VALU per frame and 16 utterances whatever the
      // chunking), not by the operand build the chunking repeats.
      if constexpr (D <= 16 && S <= 16) {
        const int wc = want ? want : 4;
        if (wc >= 11) return launch_approx_mfma<D, S, 11>(a, pv, ascore, aeps);
        if (wc >= 6) return launch_approx_mfma<D, S, 6>(a, pv, ascore, aeps);
        if (wc >= 4) return launch_approx_mfma<D, S, 4>(a, pv, ascore, aeps);
        return launch_approx_mfma<D, S, 3>(a, pv, ascore, aeps);
      } else if constexpr (D <= 16) {
        if ((want ? want : 4) >= 4) return launch_approx_mfma<D, S, 4>(a, pv, ascore, aeps);
        return launch_approx_mfma<D, S, 2>(a, pv, ascore, aeps);
      } else if constexpr (S <= 16) {
        if ((want ? want : 3) >= 3) return launch_approx_mfma<D, S, 3>(a, pv, ascore, aeps);
        return launch_approx_mfma<D, S, 2>(a, pv, ascore, aeps);
      } else {
        if ((want ? want : 1) >= 2) return launch_approx_mfma<D, S, 2>(a, pv, ascore, aeps);
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
