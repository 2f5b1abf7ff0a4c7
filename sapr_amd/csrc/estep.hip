// Forward-backward (Baum-Welch E-step) and forward scoring for diagonal-Gaussian HMMs on gfx950.
//
// Replaces, for a whole batch of utterances at once, what hmmlearn does per sequence inside
// GaussianHMM.fit / .score as the reference calls them (hmmlearn_hmm.py:103-104):
//   _compute_log_likelihood (stats.py)           -> frame_log_densities (emission.h)
//   _hmmc.cpp forward_log / backward_log         -> fb_forward_kernel / fb_backward_kernel
//   _compute_posteriors_log, compute_log_xi_sum  -> fb_backward_kernel
//   _accumulate_sufficient_statistics (hmm.py)   -> fb_backward_kernel + fb_obs_kernel
// followed by a fixed-order (deterministic, atomics-free) reduction over the utterances of each
// word model.  CPU restatement: oracle/hmmlearn_oracle.py (accumulate / forward_log / ...).
//
// Mapping: as in viterbi.hip one lane owns one utterance and a workgroup (256 utterances) is
// homogeneous in its word model, so model parameters are wavefront-uniform scalars.  The two
// lattices a sequence needs twice (log-densities b[t][s] and forward values) are parked in HBM in
// a lane-contiguous layout [t][s][slot] — every access is a coalesced 512-byte row per wavefront —
// and the posteriors overwrite the forward lattice in place for the Σγx / Σγx² pass.
//
// Arithmetic is float64 throughout; exp/log come from the device math library, so results agree
// with the CPU evaluation to ~1e-13 relative (tests use 1e-9), not bit for bit.
#include "emission_quick.h"

namespace sapr {
namespace {

#include "lse_unit.h"

using namespace emission;

constexpr int kBlock = 256;

__host__ __device__ inline int64_t round_up64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// _hmmc.cpp logsumexp over the two candidates of a bidiagonal column/row: log(exp(a - m) + exp(b - m)) + m with
// m = max(a, b).  One of the two exponentials is exp(0) = 1, so this is m + log1p(exp(-|a - b|)): one exponential
// instead of two, same value to the last bit or two (the E-step is compared at 1e-9, §4.3).  Round 4: the exponential,
// the reciprocal and the logarithm come from ONE short chain (lse_unit.h: 58 float64 instructions where the library's
// exp, two IEEE divisions and the log1p series took ~105) and the call is branch-free — an infinite maximum (an
// unreachable state, or an overflow) runs the arithmetic on d = 0 and is selected away — so that the nine independent
// calls of a frame interleave instead of forming one dependent chain each behind its own divergent branch.
__device__ __forceinline__ double lse2(double a, double b) {
  // both -inf: a - b is NaN and v_max_f64 returns its other operand, 0 — the arithmetic then runs on d = 0 and the
  // infinite maximum absorbs the finite log 2 (no select on the way: a v_cndmask_b32 that takes its mask from vcc is
  // the slowest vector instruction of this chip, scripts/ubench/mix_rate)
  const double m = __builtin_fmax(a, b), d = __builtin_fmax(__builtin_fabs(a - b), 0.0);
  double e, inv, l1p;
  lse2_terms(d, &e, &inv, &l1p);
  return m + l1p;
}

// the same, together with the share of the SECOND argument in the sum, exp(b - result) = 1 / (1 + e) or e / (1 + e):
// the backward pass of the bidiagonal E-step is a smoothing recursion over these shares (fb_smooth_obs_kernel).  An
// unreachable state (both arguments -inf) gets the share 1/2 of d = 0: its posterior is 0 whatever the share.
__device__ __forceinline__ double lse2_share(double a, double b, double &share_b) {
  const double m = __builtin_fmax(a, b), d = __builtin_fmax(__builtin_fabs(a - b), 0.0);
  double e, inv, l1p;
  lse2_terms(d, &e, &inv, &l1p);
  share_b = b >= a ? inv : e * inv;
  return m + l1p;
}

// _hmmc.cpp logaddexp
__device__ __forceinline__ double logaddexp(double a, double b) {
  const double m = a > b ? a : b;
  double e, inv, l1p;
  lse2_terms(isinf(m) ? 0.0 : fabs(b - a), &e, &inv, &l1p);
  const double r = m + l1p;
  return a == neg_inf() ? b : (b == neg_inf() ? a : r);
}

template <int S>
__device__ __forceinline__ double lse_all(const double (&v)[S]) {
  double m = v[0];
#pragma unroll
  for (int i = 1; i < S; ++i) m = v[i] > m ? v[i] : m;
  if (isinf(m)) return m;
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < S; ++i) acc += exp_unit(v[i] - m);
  return log(acc) + m;
}

// -------------------------------------------------------------------------------------------
// E-step emission: lat_b[t][s][slot] for every frame, on a grid of (utterance tile, chunk of kEmitFrames frames).
// The lattice recursions are sequential in t and a training batch has one word model per utterance (~1.5
// wavefronts per SIMD in fb_forward_kernel); the log-densities have no such dependence, so they are computed
// here at full occupancy (13 x the workgroups) and the forward kernel only reads them back.
// -------------------------------------------------------------------------------------------
constexpr int kEmitFrames = 8;

// Features in slot-major order, feat_t[t][d][slot]: a lane reads ITS utterance's frame, so from the (T, D)
// concatenation a wavefront's load touches 64 different rows (64+ cache lines for 52 bytes each) and the light
// kernels of the E-step (emission at three instructions per term, the gamma-weighted sums) were bound by the
// texture addresser, not by arithmetic.  Staged once per batch (the features do not change between EM
// iterations: SAPR_ESTEP_STAGED), every later read is one coalesced 256-byte row per wavefront.
template <int D>
__global__ __launch_bounds__(kBlock) void fb_stage_kernel(const float *__restrict__ feats,
                                                          const int64_t *__restrict__ offsets,
                                                          const int32_t *__restrict__ slot_utt, int64_t n_slots,
                                                          int32_t n_fc, float *__restrict__ feat_t) {
  const int64_t tile = blockIdx.x / n_fc;
  const int t_beg = static_cast<int>(blockIdx.x - tile * n_fc) * kEmitFrames;
  const int64_t slot = tile * kBlock + threadIdx.x;
  const int64_t u = slot_utt[slot];
  if (u < 0) return;
  const int64_t beg = offsets[u];
  const int T = static_cast<int>(offsets[u + 1] - beg);
  for (int t = t_beg; t < t_beg + kEmitFrames && t < T; ++t) {
    float x[D];
    load_frame_f32<D>(feats + (beg + t) * D, x);
#pragma unroll
    for (int d = 0; d < D; ++d) feat_t[(static_cast<int64_t>(t) * D + d) * n_slots + slot] = x[d];
  }
}

template <int D, int S, int NF>
__global__ __launch_bounds__(kBlock) void fb_emit_kernel(
    const float *__restrict__ feat_t, const int64_t *__restrict__ offsets, const int32_t *__restrict__ slot_utt,
    const int32_t *__restrict__ tile_model, int64_t n_slots, int32_t n_fc, const double4 *__restrict__ prm_all,
    const double *__restrict__ gconst, double *__restrict__ lat_b) {
  static_assert(kEmitFrames % NF == 0, "whole walks");
  const int64_t tile = blockIdx.x / n_fc;
  const int t_beg = static_cast<int>(blockIdx.x - tile * n_fc) * kEmitFrames;
  const int w = tile_model[tile];
  const int64_t slot = tile * kBlock + threadIdx.x;
  const int64_t u = slot_utt[slot];
  const bool live = u >= 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - offsets[u]) : 0;
  const int Tw = wave_max_i32(T);
  const double4 *__restrict__ prm = prm_all + static_cast<int64_t>(w) * S * D;
  const double *__restrict__ gc = gconst + static_cast<int64_t>(w) * S;
  using XT = std::conditional_t<(D >= 39), float, double>;
  for (int t0 = t_beg; t0 < t_beg + kEmitFrames && t0 < Tw; t0 += NF) {
    if (t0 < T) {
      XT x[NF][D];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int t = t0 + f < T ? t0 + f : T - 1;  // frames past the end: recomputed, not stored
        const float *row = feat_t + (static_cast<int64_t>(t) * D) * n_slots + slot;
#pragma unroll
        for (int d = 0; d < D; ++d) x[f][d] = static_cast<XT>(row[static_cast<int64_t>(d) * n_slots]);
      }
      // one running store pointer per frame, advanced state by state (kept opaque: as loop-invariant scalar
      // offsets the NF * S row addresses would cost 2 SGPRs each and spill)
      double *out[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) out[f] = lat_b + (static_cast<int64_t>(t0 + f) * S) * n_slots + slot;
      frame_log_densities_quick<D, S, NF>(x, prm, gc, [&](auto, int f, double b) {
        if (t0 + f < T) *out[f] = b;
        out[f] += n_slots;
        asm volatile("" : "+v"(out[f]));
      });
    }
  }
}

// PREB: the log-densities are already in lat_b (fb_emit_kernel); otherwise they are evaluated here, in numpy's
// operation order (GaussianHMM.score: sapr_forward_diag), and stored when lat_b is given
// QEMIT (bidiagonal E-step): the log-densities are evaluated here, four frames per walk over the SGPR-resident
// parameters, from the slot-major feature copy and in fb_emit_kernel's quick form (same operations: the same values) —
// no log-density lattice: 0.8 GB written by the emission grid and read back here per 100 000 utterances fall away
template <int D, int S, bool BIDIAG, bool FASTDIV, bool PREB, bool QEMIT = false>
__global__ __launch_bounds__(kBlock) void fb_forward_kernel(
    const float *__restrict__ feats, const int64_t *__restrict__ offsets,
    const int32_t *__restrict__ slot_utt, const int32_t *__restrict__ tile_model, int64_t n_slots,
    const double4 *__restrict__ prm_all, const double *__restrict__ gconst,
    const double *__restrict__ log_start, const double *__restrict__ log_trans,
    double *__restrict__ lat_b, double *__restrict__ lat_f, double *__restrict__ loglik,
    const float *__restrict__ feat_t = nullptr) {
  const int64_t tile = blockIdx.x;
  const int w = tile_model[tile];
  const int64_t slot = tile * kBlock + threadIdx.x;
  const int64_t u = slot_utt[slot];
  const bool live = u >= 0;
  const int64_t beg = live ? offsets[u] : 0;
  const int T = live ? static_cast<int>(offsets[u + 1] - beg) : 0;
  const int Tw = wave_max_i32(T);

  const double4 *__restrict__ prm = prm_all + static_cast<int64_t>(w) * S * D;
  const double *__restrict__ gc = gconst + static_cast<int64_t>(w) * S;
  const double *__restrict__ ls = log_start + static_cast<int64_t>(w) * S;
  const double *__restrict__ lt = log_trans + static_cast<int64_t>(w) * S * S;
  const float *__restrict__ xp = feats + beg * D;

  double fwd[S];
#pragma unroll
  for (int s = 0; s < S; ++s) fwd[s] = ls[s];

  // one frame of the recursion from its log-densities
  auto step = [&](int t, const double (&b)[S]) {
    if (t == 0) {
#pragma unroll
      for (int j = 0; j < S; ++j) fwd[j] += b[j];
    } else if constexpr (BIDIAG) {
      if (lat_f) {  // E-step: keep the share of each state's mass that STAYED (the rest came from j - 1)
        const int64_t row = static_cast<int64_t>(t) * S;
#pragma unroll
        for (int j = S - 1; j >= 1; --j) {
          double stay;
          fwd[j] = lse2_share(fwd[j - 1] + lt[(j - 1) * S + j], fwd[j] + lt[j * S + j], stay) + b[j];
          lat_f[(row + j) * n_slots + slot] = stay;
        }  // (state 0 has no predecessor: its share is the constant 1 and is not stored)
      } else {
#pragma unroll
        for (int j = S - 1; j >= 1; --j)
          fwd[j] = lse2(fwd[j - 1] + lt[(j - 1) * S + j], fwd[j] + lt[j * S + j]) + b[j];
      }
      fwd[0] = (fwd[0] + lt[0]) + b[0];
    } else {
      double prev[S], work[S];
#pragma unroll
      for (int s = 0; s < S; ++s) prev[s] = fwd[s];
#pragma unroll
      for (int j = 0; j < S; ++j) {
#pragma unroll
        for (int i = 0; i < S; ++i) work[i] = prev[i] + lt[i * S + j];
        fwd[j] = lse_all<S>(work) + b[j];
      }
    }
    if constexpr (!BIDIAG) {
      if (lat_f) {
        const int64_t row = static_cast<int64_t>(t) * S;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          if constexpr (!PREB) lat_b[(row + j) * n_slots + slot] = b[j];
          lat_f[(row + j) * n_slots + slot] = fwd[j];
        }
      }
    }
  };
  // bidiagonal E-step: the last forward row (the posteriors of the last frame start from it) goes where that frame's
  // log-densities were — nothing reads them after the forward pass
  auto finish = [&]() {
    if constexpr (BIDIAG) {
      if (lat_f && T > 0) {
        const int64_t row = static_cast<int64_t>(T - 1) * S;
#pragma unroll
        for (int j = 0; j < S; ++j) lat_b[(row + j) * n_slots + slot] = fwd[j];
      }
    }
  };

  if constexpr (QEMIT) {
    static_assert(BIDIAG && !PREB, "the fused form of the bidiagonal E-step");
    using XT = std::conditional_t<(D >= 39), float, double>;
#ifndef SAPR_FWD_NF
#define SAPR_FWD_NF ((D >= 39 || S > 10) ? 2 : 4)
#endif
    constexpr int NF = SAPR_FWD_NF;  // (256 registers: two wavefronts per SIMD)
    for (int t0 = 0; t0 < Tw; t0 += NF) {
      if (t0 < T) {
        XT xq[NF][D];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int t = t0 + f < T ? t0 + f : T - 1;  // frames past the end: evaluated, not used
          const float *row = feat_t + (static_cast<int64_t>(t) * D) * n_slots + slot;
#pragma unroll
          for (int d = 0; d < D; ++d) xq[f][d] = static_cast<XT>(row[static_cast<int64_t>(d) * n_slots]);
        }
        double bq[NF][S];
        frame_log_densities_quick<D, S, NF>(xq, prm, gc, [&](auto jc, int f, double bv) {
          bq[f][decltype(jc)::value] = bv;
        });
#pragma unroll
        for (int f = 0; f < NF; ++f)
          if (t0 + f < T) step(t0 + f, bq[f]);
      }
    }
    finish();
    if (live) loglik[u] = T > 0 ? lse_all<S>(fwd) : 0.0;
    return;
  }

  if constexpr (PREB) {
    double b[S], nb[S];
    if (T > 0) {
#pragma unroll
      for (int j = 0; j < S; ++j) b[j] = lat_b[static_cast<int64_t>(j) * n_slots + slot];
    }
    for (int t = 0; t < Tw; ++t) {
      if (t < T) {
        if (t + 1 < T) {  // next frame's row: in flight behind this frame's exp / log work
          const int64_t row = static_cast<int64_t>(t + 1) * S;
#pragma unroll
          for (int j = 0; j < S; ++j) nb[j] = lat_b[(row + j) * n_slots + slot];
        }
        step(t, b);
#pragma unroll
        for (int j = 0; j < S; ++j) b[j] = nb[j];
      }
    }
    finish();
    if (live) loglik[u] = T > 0 ? lse_all<S>(fwd) : 0.0;
    return;
  }

  if constexpr (FASTDIV) {
    // two frames per walk over the parameters (emission.h EmitLoop2): a training batch gives every utterance ONE
    // word model, so this grid holds ~1.5 wavefronts per SIMD and the scalar loads' latency would show
    using XT = std::conditional_t<(D >= 39), float, double>;
    XT xa[D], xb[D];
    auto load2 = [&](const float *p, XT (&dst)[D]) {
      if constexpr (D >= 39)
        load_frame_f32<D>(p, dst);
      else
        load_frame<D>(p, dst);
    };
    for (int t = 0; t < Tw; t += 2) {
      if (t < T) {
        const bool two = (t + 1) < T;
        load2(xp + static_cast<int64_t>(t) * D, xa);
        load2(xp + static_cast<int64_t>(two ? t + 1 : t) * D, xb);
        double ba[S], bb[S];
        frame_log_densities2<D, S, false>(xa, xb, prm, gc, ba, bb);
        step(t, ba);
        if (two) step(t + 1, bb);
      }
    }
  } else {
    double x[D];
    for (int t = 0; t < Tw; ++t) {
      if (t < T) {
        load_frame<D>(xp + static_cast<int64_t>(t) * D, x);
        double b[S];
        frame_log_densities<D, S, false, false>(x, prm, gc, b);
        step(t, b);
      }
    }
  }
  finish();
  if (live) loglik[u] = T > 0 ? lse_all<S>(fwd) : 0.0;
}

// -------------------------------------------------------------------------------------------
// The backward half of the E-step.  utt_stats[u] / wave_stats[w] = {nobs, logprob, start[S], trans[S][S], post[S]}
//
// Bidiagonal models: a smoothing recursion (round 3b).  base.py computes log beta (a logsumexp per state and frame),
// the posteriors as softmax(fwd + bwd) and log xi = fwd + log a + b + bwd - logprob.  For a bidiagonal model all of it
// follows from the forward pass: the share of state j's forward mass at frame t that stayed in j (fb_forward_kernel
// stores it in the lattice the forward values used to occupy; state 0 has no predecessor: constant 1, not stored) is
// P(q_(t-1) = j | q_t = j, x_1..t), the rest came from j - 1:
//     xi_t(j -> j) = gamma_t(j) stay_t(j),   xi_t(j-1 -> j) = gamma_t(j) - xi_t(j -> j),
//     gamma_(t-1)(i) = xi_t(i -> i) + xi_t(i -> i+1)
// — two multiplications and two additions per state and frame where the recursion on beta took two exponentials and a
// log1p, no log-density lattice to read, the same sums to rounding (statistics are compared at 1e-9).  The posteriors
// of the last frame are softmax(fwd_(T-1)) as in the reference (NaN when no state is reachable).
// -------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ void softmax_last_row(const double *__restrict__ lat_b, int64_t n_slots, int64_t slot, int T,
                                                 double (&g)[S]) {
  double lg[S];
#pragma unroll
  for (int s = 0; s < S; ++s) lg[s] = lat_b[(static_cast<int64_t>(T - 1) * S + s) * n_slots + slot];
  double mx = lg[0];
#pragma unroll
  for (int s = 1; s < S; ++s) mx = lg[s] > mx ? lg[s] : mx;
  double den = 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    lg[s] = exp_unit(lg[s] - mx);  // all -inf: NaN, as exp(lg - (-inf)) is in the reference
    den += lg[s];
  }
  const double inv = 1.0 / den;
#pragma unroll
  for (int s = 0; s < S; ++s) g[s] = lg[s] * inv;
}

// one step t -> t-1 of the smoothing recursion: g = gamma_t on entry, gamma_(t-1) on exit; xs += the xi terms of the step
// ([i] = (i,i), [S+i-1] = (i-1,i)); st[1..S-1] = the stay shares of frame t
template <int S>
__device__ __forceinline__ void smooth_step(double (&g)[S], const double (&st)[S], double (&xs)[2 * S]) {
  double x_stay[S], x_move[S];
  x_stay[0] = g[0];
  x_move[0] = 0.0;
  xs[0] += x_stay[0];
#pragma unroll
  for (int i = 1; i < S; ++i) {
    x_stay[i] = g[i] * st[i];
    x_move[i] = g[i] - x_stay[i];
    xs[i] += x_stay[i];
    xs[S + i - 1] += x_move[i];
  }
#pragma unroll
  for (int i = 0; i < S; ++i) g[i] = x_stay[i] + (i + 1 < S ? x_move[i + 1] : 0.0);
}

// The smoothing recursion alone, posteriors written over the shares for fb_obs_kernel (round 3b's form: 0.8 GB of
// posteriors written and read back per 100 000 utterances; kept behind SAPR_ESTEP_OBS=split for comparison)
template <int S>
__global__ __launch_bounds__(kBlock) void fb_smooth_kernel(
    const int64_t *__restrict__ offsets, const int32_t *__restrict__ slot_utt, int64_t n_slots,
    const double *__restrict__ lat_b, double *__restrict__ lat_f, const double *__restrict__ loglik,
    double *__restrict__ utt_stats) {
  constexpr int K = 2 + S + S * S + S;
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t u = slot_utt[slot];
  if (u < 0) return;
  const int T = static_cast<int>(offsets[u + 1] - offsets[u]);
  double *__restrict__ out = utt_stats + u * K;
  if (T <= 0) {
    for (int k = 0; k < K; ++k) out[k] = 0.0;
    return;
  }
  double g[S], post[S], xs[2 * S];
#pragma unroll
  for (int i = 0; i < 2 * S; ++i) xs[i] = 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) post[s] = 0.0;
  softmax_last_row<S>(lat_b, n_slots, slot, T, g);
  double st[S];  // stay shares of the step into frame t: one row ahead of the recursion
  st[0] = 1.0;
  if (T > 1) {
#pragma unroll
    for (int s = 1; s < S; ++s) st[s] = lat_f[(static_cast<int64_t>(T - 1) * S + s) * n_slots + slot];
  }
  for (int t = T - 1; t >= 0; --t) {
    double nst[S];
    nst[0] = 1.0;
    if (t >= 2) {
#pragma unroll
      for (int s = 1; s < S; ++s) nst[s] = lat_f[(static_cast<int64_t>(t - 1) * S + s) * n_slots + slot];
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      post[s] += g[s];
      lat_f[(static_cast<int64_t>(t) * S + s) * n_slots + slot] = g[s];  // gamma replaces the shares in place
    }
    if (t == 0) break;
    smooth_step<S>(g, st, xs);
#pragma unroll
    for (int s = 0; s < S; ++s) st[s] = nst[s];
  }
  out[0] = 1.0;
  out[1] = loglik[u];
#pragma unroll
  for (int s = 0; s < S; ++s) out[2 + s] = g[s];  // stats['start'] += posteriors[0]
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) out[2 + S + i * S + j] = 0.0;
  if (T > 1) {  // stats['trans'] += exp(log_xi_sum)   (skipped for one-frame sequences, base.py)
#pragma unroll
    for (int i = 0; i < S; ++i) {
      out[2 + S + i * S + i] = xs[i];
      if (i + 1 < S) out[2 + S + i * S + i + 1] = xs[S + i];
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) out[2 + S + S * S + s] = post[s];
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// -------------------------------------------------------------------------------------------
// Smoothing recursion AND the posterior-weighted observation sums in one pass (round 4): the posteriors are consumed
// where they are produced and never reach HBM — 0.8 GB written by the smoothing pass and 0.8 GB (+ the features, once
// per chunk of states) read back by fb_obs_kernel per 100 000 utterances fall away, and with them one of the E-step's
// four lattice kernels.
//
// stats['obs'] = sum_t gamma_t^T x_t and stats['obs**2'] = sum_t gamma_t^T x_t^2 are matrix products whose inner
// dimension runs over (utterance, frame) — with a lane per utterance that dimension lies ACROSS the lanes, and 2 S D
// float64 accumulators per lane (260 at 13 x 10) do not fit a register file, which is why fb_obs_kernel is a pass of
// its own over chunks of two states.  On the float64 matrix cores the accumulators are shared by the wavefront:
// v_mfma_f64_16x16x4_f64 with M = state, N = feature dimension, K = four utterances of the wavefront, the same frame —
// 16 instructions cover the 64 utterances, two N tiles (x | x^2) 32 per frame, 16 accumulator registers per lane.  The
// operands must change from lane = utterance to lane = (row, k): every frame each lane writes its S posteriors
// (float64) and D feature values (float32) into the wavefront's own 13 KB of LDS and reads back one posterior and one
// feature value per MFMA.  Row strides of 68 doubles / 68 floats make both directions conflict-free (writes: lanes
// contiguous; reads: row stride = 8 resp. 4 banks, k neighbours 2 resp. 1 bank apart).  Column D of the feature tile
// is the constant 1, so sum_t gamma_t (stats['post']) falls out of the same product.  x^2 is rounded to float32 before
// it is widened, as numpy squares the float32 feature array.
//
// One wavefront per workgroup (64 utterance slots, a quarter of a tile): no workgroup barrier anywhere, and 1 563
// workgroups spread over 256 CUs where 391 four-wavefront ones left half of them with twice the work of the rest.
// Outputs per WAVEFRONT: wave_stats[wq][K] (the lanes' rows summed by a fixed xor butterfly) and wave_obs[wq][2][S D];
// fb_reduce_kernel adds the wavefronts of a word in order — fixed shapes throughout: bit-reproducible statistics.
// -------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int kGs = 68;  // row stride of the posterior rows in LDS (doubles)
constexpr int kXs = 68;  // row stride of the feature rows in LDS (floats)

template <int D, int S>
__global__ __launch_bounds__(64) void fb_smooth_obs_kernel(
    const float *__restrict__ feat_t, const int64_t *__restrict__ offsets, const int32_t *__restrict__ slot_utt,
    int64_t n_slots, const double *__restrict__ lat_b, const double *__restrict__ lat_f,
    const double *__restrict__ loglik, double *__restrict__ wave_stats, double *__restrict__ wave_obs) {
  constexpr int MT = (S + 15) / 16, NT = D / 16 + 1, K = 2 + S + S * S + S, pairs = S * D;
  static_assert(D % 16 != 0, "column D of the last feature tile carries the constant 1 (sum of the posteriors)");
  static_assert(K <= MT * 16 * kGs, "the statistics row is staged in the posterior rows' LDS");
  __shared__ double s_g[MT * 16][kGs];
  __shared__ float s_x[NT * 16][kXs];
  const int lane = threadIdx.x, n = lane & 15, k = lane >> 4;
  const int64_t wq = blockIdx.x;
  const int64_t slot = wq * 64 + lane;
  const int64_t u = slot_utt[slot];
  const int T = u >= 0 ? static_cast<int>(offsets[u + 1] - offsets[u]) : 0;
  const int Tw = wave_max_i32(T);

#pragma unroll
  for (int r = S; r < MT * 16; ++r) s_g[r][lane] = 0.0;  // A rows past the last state
#pragma unroll
  for (int r = D; r < NT * 16; ++r) s_x[r][lane] = r == D ? 1.0f : 0.0f;

  f64x4 acc1[MT][NT], acc2[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc1[mt][nt] = acc2[mt][nt] = f64x4{0.0, 0.0, 0.0, 0.0};
  double g[S], xs[2 * S];
#pragma unroll
  for (int s = 0; s < S; ++s) g[s] = 0.0;  // a lane's posteriors are 0 until the recursion reaches ITS last frame
#pragma unroll
  for (int i = 0; i < 2 * S; ++i) xs[i] = 0.0;

  // rows of frames t, t-1 in registers, t-2 in flight: nothing but these loads stands between the kernel and HBM.
  // Lanes whose utterance has ended (or not begun: t >= T) load defined addresses of undefined content — masked below.
  double st0[S], st1[S];
  float x0[D], x1[D];
  auto load_rows = [&](int t, double (&st)[S], float (&x)[D]) {
    st[0] = 1.0;
#pragma unroll
    for (int s = 1; s < S; ++s) st[s] = lat_f[(static_cast<int64_t>(t) * S + s) * n_slots + slot];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = feat_t[(static_cast<int64_t>(t) * D + d) * n_slots + slot];
  };
  if (Tw >= 1) load_rows(Tw - 1, st0, x0);
  if (Tw >= 2) load_rows(Tw - 2, st1, x1);

  for (int t = Tw - 1; t >= 0; --t) {
    double st2[S];
    float x2[D];
    if (t >= 2) load_rows(t - 2, st2, x2);
    const bool active = t < T;
    if (t == T - 1) softmax_last_row<S>(lat_b, n_slots, slot, T, g);
    // lane = utterance -> LDS
#pragma unroll
    for (int s = 0; s < S; ++s) s_g[s][lane] = g[s];
#pragma unroll
    for (int d = 0; d < D; ++d) s_x[d][lane] = active ? x0[d] : 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // LDS -> lane = (row n, utterance 4 q + k) of the MFMA operands
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      double a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = s_g[mt * 16 + n][4 * q + k];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float xb = s_x[nt * 16 + n][4 * q + k];
        const double b1 = static_cast<double>(xb), b2 = static_cast<double>(xb * xb);  // float32 square, then widened
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc1[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b1, acc1[mt][nt], 0, 0, 0);
          acc2[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b2, acc2[mt][nt], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (active && t >= 1) smooth_step<S>(g, st0, xs);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      st0[s] = st1[s];
      st1[s] = st2[s];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      x0[d] = x1[d];
      x1[d] = x2[d];
    }
  }

  // the wavefront's statistics row, staged in LDS (the posterior rows are free now) and stored in one sweep
  double *s_out = &s_g[0][0];
  for (int i = lane; i < K; i += 64) s_out[i] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool live = T > 0;
  {
    const double nobs = wave_sum_f64(live ? 1.0 : 0.0), lp = wave_sum_f64(live ? loglik[u] : 0.0);
    if (lane == 0) {
      s_out[0] = nobs;
      s_out[1] = lp;
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {  // stats['start'] += posteriors[0]: g holds gamma_0 of every live lane
    const double v = wave_sum_f64(g[s]);
    if (lane == 0) s_out[2 + s] = v;
  }
#pragma unroll
  for (int i = 0; i < S; ++i) {  // stats['trans'] += exp(log_xi_sum) (one-frame sequences never took a step: zeros)
    const double v = wave_sum_f64(xs[i]);
    if (lane == 0) s_out[2 + S + i * S + i] = v;
    if (i + 1 < S) {
      const double m = wave_sum_f64(xs[S + i]);
      if (lane == 0) s_out[2 + S + i * S + i + 1] = m;
    }
  }
  // C/D of the float64 MFMA: column = lane & 15, row = (lane >> 4) + 4 * register
  double *__restrict__ obs = wave_obs + wq * 2 * pairs;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = mt * 16 + k + 4 * r;
      if (s < S) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int d = nt * 16 + n;
          if (d < D) {
            obs[s * D + d] = acc1[mt][nt][r];
            obs[pairs + s * D + d] = acc2[mt][nt][r];
          } else if (d == D) {
            s_out[2 + S + S * S + s] = acc1[mt][nt][r];  // the constant column: sum_t gamma_t(s)
          }
        }
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int i = lane; i < K; i += 64) wave_stats[wq * K + i] = s_out[i];
}

// -------------------------------------------------------------------------------------------
// Dense models: _hmmc.cpp backward_log + _compute_posteriors_log + compute_log_xi_sum as written (log domain, the S*S
// log xi sums kept in the output row itself; slow path, rarely used); posteriors replace the forward lattice in place
// -------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(kBlock) void fb_backward_dense_kernel(
    const int64_t *__restrict__ offsets, const int32_t *__restrict__ slot_utt,
    const int32_t *__restrict__ tile_model, int64_t n_slots, const double *__restrict__ log_trans,
    const double *__restrict__ lat_b, double *__restrict__ lat_f, const double *__restrict__ loglik,
    double *__restrict__ utt_stats) {
  constexpr int K = 2 + S + S * S + S;
  const int64_t tile = blockIdx.x;
  const int w = tile_model[tile];
  const int64_t slot = tile * kBlock + threadIdx.x;
  const int64_t u = slot_utt[slot];
  if (u < 0) return;
  const int T = static_cast<int>(offsets[u + 1] - offsets[u]);
  const double *__restrict__ lt = log_trans + static_cast<int64_t>(w) * S * S;
  double *__restrict__ out = utt_stats + u * K;
  if (T <= 0) {
    for (int k = 0; k < K; ++k) out[k] = 0.0;
    return;
  }
  const double logprob = loglik[u];
  double bwd[S], post[S], fw[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    bwd[s] = 0.0;
    post[s] = 0.0;
  }
  for (int k = 0; k < S * S; ++k) out[2 + S + k] = neg_inf();
#pragma unroll
  for (int s = 0; s < S; ++s) fw[s] = lat_f[(static_cast<int64_t>(T - 1) * S + s) * n_slots + slot];
  for (int t = T - 1; t >= 0; --t) {
    // posteriors of frame t (base.py _compute_posteriors_log: row soft-max of fwd + bwd).  exp(lg - logsumexp(lg))
    // is evaluated as exp(lg - max) / sum: the S exponentials of the logsumexp are the numerators
    double lg[S];
#pragma unroll
    for (int s = 0; s < S; ++s) lg[s] = fw[s] + bwd[s];
    double mx = lg[0];
#pragma unroll
    for (int s = 1; s < S; ++s) mx = lg[s] > mx ? lg[s] : mx;
    double den = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      lg[s] = exp_unit(lg[s] - mx);  // all -inf: NaN, as exp(lg - (-inf)) is in the reference
      den += lg[s];
    }
    const double inv = 1.0 / den;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const double g = lg[s] * inv;
      post[s] += g;
      lat_f[(static_cast<int64_t>(t) * S + s) * n_slots + slot] = g;  // gamma replaces fwd in place
      if (t == 0) out[2 + s] = g;                                      // stats['start'] += posteriors[0]
    }
    if (t == 0) break;
    // step to t-1: needs b[t][.] and fwd[t-1][.]
    double bt[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      bt[s] = lat_b[(static_cast<int64_t>(t) * S + s) * n_slots + slot];
      fw[s] = lat_f[(static_cast<int64_t>(t - 1) * S + s) * n_slots + slot];
    }
    double nb[S], work[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
#pragma unroll
      for (int j = 0; j < S; ++j) {
        work[j] = lt[i * S + j] + bt[j] + bwd[j];
        const double lx = fw[i] + lt[i * S + j] + bt[j] + bwd[j] - logprob;
        out[2 + S + i * S + j] = logaddexp(out[2 + S + i * S + j], lx);
      }
      nb[i] = lse_all<S>(work);
    }
#pragma unroll
    for (int i = 0; i < S; ++i) bwd[i] = nb[i];
  }
  out[0] = 1.0;
  out[1] = logprob;
  // stats['trans'] += exp(log_xi_sum)   (skipped for one-frame sequences, base.py)
  for (int k = 0; k < S * S; ++k) out[2 + S + k] = T > 1 ? exp(out[2 + S + k]) : 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) out[2 + S + S * S + s] = post[s];
}

// -------------------------------------------------------------------------------------------
// stats['obs'] += posteriors.T @ X ; stats['obs**2'] += posteriors.T @ X**2 — per tile partials.
// X**2 is rounded to float32 first, exactly as numpy evaluates it on the float32 feature array.
//
// Same mapping as the trellis kernels: one lane per utterance slot.  A workgroup handles SC states
// of one tile; every lane walks its own utterance once (gamma[t][s][slot] is a coalesced row per
// wavefront, the frame is 4*D contiguous bytes) with 2*SC*D float64 accumulators in registers, then
// the 256 lanes are combined in a FIXED order (xor-butterfly inside the wavefront, wavefronts 0..3
// in sequence): deterministic, no atomics.  The SC-chunks of one tile sit on one XCD back to back
// (same decode as viterbi.hip) so the features they all re-read come from that XCD's L2; features are read
// from the slot-major copy (fb_stage_kernel), one coalesced row per wavefront and dimension.
// -------------------------------------------------------------------------------------------
constexpr int kXcd = 8;

template <int D, int SC>
__global__ __launch_bounds__(kBlock) void fb_obs_kernel(const float *__restrict__ feat_t,
                                                        const int64_t *__restrict__ offsets,
                                                        const int32_t *__restrict__ slot_utt, int64_t n_tiles,
                                                        int64_t n_slots, int S, int n_chunks,
                                                        const double *__restrict__ gamma,
                                                        double *__restrict__ tile_obs) {
  __shared__ double part[kBlock / 64][2 * SC * D];
  const int64_t id = blockIdx.x;
  const int xcd = static_cast<int>(id % kXcd);
  const int64_t k = id / kXcd;
  const int64_t tile = (k / n_chunks) * kXcd + xcd;
  const int s0 = static_cast<int>(k % n_chunks) * SC;
  if (tile >= n_tiles) return;  // whole workgroup (grid is padded to a multiple of 8 tiles)
  const int64_t slot = tile * kBlock + threadIdx.x;
  const int64_t u = slot_utt[slot];
  const int T = u >= 0 ? static_cast<int>(offsets[u + 1] - offsets[u]) : 0;
  const float *__restrict__ xp = feat_t + slot;  // slot-major rows (fb_stage_kernel)

  double o1[SC][D], o2[SC][D];
#pragma unroll
  for (int c = 0; c < SC; ++c)
#pragma unroll
    for (int d = 0; d < D; ++d) o1[c][d] = o2[c][d] = 0.0;

  for (int t = 0; t < T; ++t) {
    float xf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xf[d] = xp[(static_cast<int64_t>(t) * D + d) * n_slots];
    double g[SC];
#pragma unroll
    for (int c = 0; c < SC; ++c)
      g[c] = (s0 + c < S) ? gamma[(static_cast<int64_t>(t) * S + s0 + c) * n_slots + slot] : 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double x1 = static_cast<double>(xf[d]);
      const double x2 = static_cast<double>(xf[d] * xf[d]);  // float32 square, then widened
#pragma unroll
      for (int c = 0; c < SC; ++c) {
        o1[c][d] = __builtin_fma(g[c], x1, o1[c][d]);
        o2[c][d] = __builtin_fma(g[c], x2, o2[c][d]);
      }
    }
  }

  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
#pragma unroll
  for (int c = 0; c < SC; ++c)
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double a1 = wave_sum_f64(o1[c][d]), a2 = wave_sum_f64(o2[c][d]);
      if (lane == 0) {
        part[wave][c * D + d] = a1;
        part[wave][SC * D + c * D + d] = a2;
      }
    }
  __syncthreads();
  const int pairs = S * D;
  for (int i = threadIdx.x; i < 2 * SC * D; i += kBlock) {
    const int which = i / (SC * D), r = i - which * SC * D;
    const int c = r / D, d = r - c * D;
    if (s0 + c < S) {
      double acc = part[0][i];
#pragma unroll
      for (int wv = 1; wv < kBlock / 64; ++wv) acc += part[wv][i];
      tile_obs[(tile * 2 + which) * pairs + (s0 + c) * D + d] = acc;
    }
  }
}

// fixed-order reduction, level 1: the K per-utterance statistics of one tile's utterances -> tile_stats[tile][K].
// Coalesced over k; the 256 rows of a tile are added as two runs of 128 (one per half of the workgroup, each in row
// order with eight loads in flight) whose sums are then added — a fixed shape, so the statistics stay bit-reproducible;
// one lane per statistic walking all 256 rows with one load in flight took 0.12 ms per 100 000 utterances.
__global__ __launch_bounds__(kBlock) void fb_tile_reduce_kernel(const int32_t *__restrict__ slot_utt, int K,
                                                                const double *__restrict__ utt_stats,
                                                                double *__restrict__ tile_stats) {
  static_assert(kBlock == 256, "two runs of 128 rows, 128 statistics per sweep");
  const int64_t tile = blockIdx.x;
  __shared__ int32_t us[kBlock];
  __shared__ double upper[128];
  us[threadIdx.x] = slot_utt[tile * kBlock + threadIdx.x];
  __syncthreads();
  const int col = static_cast<int>(threadIdx.x & 127), run = static_cast<int>(threadIdx.x >> 7);
  for (int k0 = 0; k0 < K; k0 += 128) {  // (uniform trip count: the barriers below are reached by every thread)
    const int k = k0 + col;
    double acc = 0.0;
    if (k < K) {
      for (int j0 = run * 128; j0 < run * 128 + 128; j0 += 8) {
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int64_t u = us[j0 + i];
          v[i] = u >= 0 ? utt_stats[u * K + k] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (us[j0 + i] >= 0) acc += v[i];
      }
    }
    if (run == 1 && k < K) upper[col] = acc;
    __syncthreads();
    if (run == 0 && k < K) tile_stats[tile * K + k] = acc + upper[col];
    __syncthreads();
  }
}

// level 2: stats[w] = {nobs, logprob, start[S], trans[S][S], post[S], obs[S][D], obs2[S][D]} summed over the word's
// partial rows in order; a tile contributes `sub` consecutive rows (1: per-tile rows of the split path, 4: the
// per-wavefront rows of fb_smooth_obs_kernel).  Eight rows in flight per lane, added in row order.
__global__ void fb_reduce_kernel(const int32_t *__restrict__ model_tile_off, int W, int S, int D, int sub,
                                 const double *__restrict__ tile_stats, const double *__restrict__ tile_obs,
                                 double *__restrict__ stats) {
  const int K = 2 + S + S * S + S;
  const int pairs = S * D;
  const int Kw = K + 2 * pairs;
  const int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (idx >= static_cast<int64_t>(W) * Kw) return;
  const int w = static_cast<int>(idx / Kw), k = static_cast<int>(idx - static_cast<int64_t>(w) * Kw);
  const int64_t r0 = static_cast<int64_t>(model_tile_off[w]) * sub, r1 = static_cast<int64_t>(model_tile_off[w + 1]) * sub;
  const double *__restrict__ src = k < K ? tile_stats + k : tile_obs + (k - K);
  const int64_t stride = k < K ? K : 2 * pairs;
  double acc = 0.0;
  int64_t r = r0;
  for (; r + 8 <= r1; r += 8) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[(r + i) * stride];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
  }
  for (; r < r1; ++r) acc += src[r * stride];
  stats[idx] = acc;
}

struct FbArgs {
  const float *feats;
  const int64_t *offsets;
  const int32_t *slot_utt, *tile_model;
  int64_t n_tiles, n_slots;
  PackView pv;
  double *lat_b, *lat_f, *loglik;
  float *feat_t;  // slot-major feature copy inside the E-step workspace (NULL: forward scoring only)
  bool staged;    // feat_t already holds this batch
  hipStream_t stream;
};

template <int D, int S>
int launch_forward(const FbArgs &a, int topology, int fast, int max_T) {
  dim3 grid(static_cast<unsigned>(a.n_tiles)), block(kBlock);
  if (a.lat_b) {  // E-step: emission at full occupancy first, then the recursion over the stored rows
    constexpr int NF = D >= 39 ? 2 : 4;
    const int n_fc = ((max_T > 0 ? max_T : 1) + kEmitFrames - 1) / kEmitFrames;
    const int64_t blocks = a.n_tiles * n_fc;
    if (blocks > 0x7fffffffLL) return fail(SAPR_ERR_ARG, "grid too large (%lld blocks)", (long long)blocks);
    if (!a.staged)
      SAPR_LAUNCH((fb_stage_kernel<D>), dim3(static_cast<unsigned>(blocks)), block, 0, a.stream, a.feats, a.offsets,
                  a.slot_utt, a.n_slots, n_fc, a.feat_t);
    static const bool fused = [] {  // SAPR_ESTEP_EMIT=grid: the emission grid + the forward pass over its lattice
      const char *e = getenv("SAPR_ESTEP_EMIT");
      return !(e && e[0] == 'g');
    }();
    if (topology == SAPR_TOPO_BIDIAG && fused && a.lat_f) {
      SAPR_LAUNCH((fb_forward_kernel<D, S, true, false, false, true>), grid, block, 0, a.stream, a.feats, a.offsets,
                  a.slot_utt, a.tile_model, a.n_slots, a.pv.prm, a.pv.gconst, a.pv.log_start, a.pv.log_trans, a.lat_b,
                  a.lat_f, a.loglik, a.feat_t);
      SAPR_HIP_TRY(hipGetLastError());
      return 0;
    }
    SAPR_LAUNCH((fb_emit_kernel<D, S, NF>), dim3(static_cast<unsigned>(blocks)), block, 0, a.stream, a.feat_t,
                a.offsets, a.slot_utt, a.tile_model, a.n_slots, n_fc, a.pv.prm, a.pv.gconst, a.lat_b);
    SAPR_HIP_TRY(hipGetLastError());
    if (topology == SAPR_TOPO_BIDIAG)
      SAPR_LAUNCH((fb_forward_kernel<D, S, true, false, true>), grid, block, 0, a.stream, a.feats, a.offsets,
                  a.slot_utt, a.tile_model, a.n_slots, a.pv.prm, a.pv.gconst, a.pv.log_start, a.pv.log_trans, a.lat_b,
                  a.lat_f, a.loglik);
    else
      SAPR_LAUNCH((fb_forward_kernel<D, S, false, false, true>), grid, block, 0, a.stream, a.feats, a.offsets,
                  a.slot_utt, a.tile_model, a.n_slots, a.pv.prm, a.pv.gconst, a.pv.log_start, a.pv.log_trans, a.lat_b,
                  a.lat_f, a.loglik);
    SAPR_HIP_TRY(hipGetLastError());
    return 0;
  }
#define SAPR_FWD(BD, FD)                                                                               \
  SAPR_LAUNCH((fb_forward_kernel<D, S, BD, FD, false>), grid, block, 0, a.stream, a.feats, a.offsets, \
                     a.slot_utt, a.tile_model, a.n_slots, a.pv.prm, a.pv.gconst, a.pv.log_start,     \
                     a.pv.log_trans, a.lat_b, a.lat_f, a.loglik)
  if (topology == SAPR_TOPO_BIDIAG) {
    if (fast)
      SAPR_FWD(true, true);
    else
      SAPR_FWD(true, false);
  } else {
    if (fast)
      SAPR_FWD(false, true);
    else
      SAPR_FWD(false, false);
  }
#undef SAPR_FWD
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

template <int S>
int launch_backward(const FbArgs &a, int topology, double *utt_stats) {
  dim3 grid(static_cast<unsigned>(a.n_tiles)), block(kBlock);
  if (topology == SAPR_TOPO_BIDIAG)
    SAPR_LAUNCH((fb_smooth_kernel<S>), grid, block, 0, a.stream, a.offsets, a.slot_utt, a.n_slots, a.lat_b, a.lat_f,
                a.loglik, utt_stats);
  else
    SAPR_LAUNCH((fb_backward_dense_kernel<S>), grid, block, 0, a.stream, a.offsets, a.slot_utt,
                       a.tile_model, a.n_slots, a.pv.log_trans, a.lat_b, a.lat_f, a.loglik, utt_stats);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

// bidiagonal models: smoothing recursion + observation sums in one pass, one wavefront per workgroup
template <int D, int S>
int launch_smooth_obs(const FbArgs &a, double *wave_stats, double *wave_obs) {
  SAPR_LAUNCH((fb_smooth_obs_kernel<D, S>), dim3(static_cast<unsigned>(a.n_tiles * (kBlock / 64))), dim3(64), 0,
              a.stream, a.feat_t, a.offsets, a.slot_utt, a.n_slots, a.lat_b, a.lat_f, a.loglik, wave_stats, wave_obs);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

size_t fb_ws_bytes(int64_t n_tiles, int S, int D, int max_T, int64_t n_utts) {
  const int64_t n_slots = n_tiles * kBlock;
  const size_t lat = static_cast<size_t>(max_T > 0 ? max_T : 1) * S * n_slots * sizeof(double);
  const size_t us = static_cast<size_t>(n_utts > 0 ? n_utts : 1) * (2 + S + S * S + S) * sizeof(double);
  // partial rows: one per WAVEFRONT (fb_smooth_obs_kernel), i.e. kBlock / 64 per tile
  const size_t to = static_cast<size_t>(n_tiles > 0 ? n_tiles : 1) * (kBlock / 64) * 2 * S * D * sizeof(double);
  const size_t ts = static_cast<size_t>(n_tiles > 0 ? n_tiles : 1) * (kBlock / 64) * (2 + S + S * S + S) * sizeof(double);
  const size_t ft = static_cast<size_t>(max_T > 0 ? max_T : 1) * D * n_slots * sizeof(float);
  return 2 * lat + us + to + ts + ft + 256;
}

}  // namespace
}  // namespace sapr

using namespace sapr;

extern "C" int sapr_fb_workspace_bytes(int64_t n_utts, int64_t n_tiles, int32_t S, int32_t D, int32_t max_T,
                                       size_t *bytes) {
  SAPR_REQUIRE(bytes && n_utts >= 0 && n_tiles >= 0 && S > 0 && D > 0 && max_T >= 0, "bad arguments");
  *bytes = fb_ws_bytes(n_tiles, S, D, max_T, n_utts);
  return 0;
}

extern "C" int sapr_stats_width(int32_t S, int32_t D, int32_t *width) {
  SAPR_REQUIRE(width && S > 0 && D > 0, "bad arguments");
  *width = 2 + S + S * S + S + 2 * S * D;
  return 0;
}

// forward log-likelihood of every utterance under its tile's model (GaussianHMM.score)
extern "C" int sapr_forward_diag(const float *feats, const int64_t *offsets, const int32_t *slot_utt,
                                 const int32_t *tile_model, int64_t n_tiles, int32_t D, const void *pack,
                                 int32_t W, int32_t S, int32_t topology, int32_t fast_div, double *loglik,
                                 void *stream) {
  SAPR_REQUIRE(n_tiles >= 0 && W > 0 && S > 0 && D > 0, "bad sizes");
  SAPR_REQUIRE(topology == SAPR_TOPO_DENSE || topology == SAPR_TOPO_BIDIAG, "bad topology");
  if (n_tiles == 0) return 0;
  SAPR_REQUIRE(feats && offsets && slot_utt && tile_model && pack && loglik, "NULL pointer argument");
  FbArgs a;
  a.feats = feats;
  a.offsets = offsets;
  a.slot_utt = slot_utt;
  a.tile_model = tile_model;
  a.n_tiles = n_tiles;
  a.n_slots = n_tiles * kBlock;
  a.pv = pack_view(pack, W, S, D);
  a.lat_b = nullptr;
  a.lat_f = nullptr;
  a.feat_t = nullptr;
  a.staged = false;
  a.loglik = loglik;
  a.stream = as_stream(stream);
  const int fast = fast_div ? 1 : 0;
  if (D == 13 && S == 10) return launch_forward<13, 10>(a, topology, fast, 0);
#ifndef SAPR_ONLY_13_10
  if (D == 13 && S == 18) return launch_forward<13, 18>(a, topology, fast, 0);
  if (D == 39 && S == 10) return launch_forward<39, 10>(a, topology, fast, 0);
  if (D == 39 && S == 18) return launch_forward<39, 18>(a, topology, fast, 0);
#endif
  return fail(SAPR_ERR_UNSUPPORTED, "trellis kernels are instantiated for (D,S) in {13,39}x{10,18}; got D=%d S=%d",
              D, S);
}

// one E-step over a batch: stats[W][width], loglik[n_utts]
extern "C" int sapr_estep_diag(const float *feats, const int64_t *offsets, const int32_t *slot_utt,
                               const int32_t *tile_model, const int32_t *model_tile_off, int64_t n_utts,
                               int64_t n_tiles, int32_t D, int32_t max_T, const void *pack, int32_t W,
                               int32_t S, int32_t topology, int32_t fast_div, void *workspace,
                               size_t workspace_size, double *loglik, double *stats, void *stream) {
  SAPR_REQUIRE(n_utts >= 0 && n_tiles >= 0 && W > 0 && S > 0 && D > 0 && max_T >= 0, "bad sizes");
  SAPR_REQUIRE(topology == SAPR_TOPO_DENSE || topology == SAPR_TOPO_BIDIAG, "bad topology");
  SAPR_REQUIRE(feats && offsets && slot_utt && tile_model && model_tile_off && pack && workspace && loglik && stats,
               "NULL pointer argument");
  if (workspace_size < fb_ws_bytes(n_tiles, S, D, max_T, n_utts))
    return fail(SAPR_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_size,
                fb_ws_bytes(n_tiles, S, D, max_T, n_utts));
  const int64_t n_slots = n_tiles * kBlock;
  const size_t lat_elems = static_cast<size_t>(max_T > 0 ? max_T : 1) * S * n_slots;
  double *lat_b = static_cast<double *>(workspace);
  double *lat_f = lat_b + lat_elems;
  double *utt_stats = lat_f + lat_elems;
  const int K = 2 + S + S * S + S;
  double *tile_obs = utt_stats + static_cast<size_t>(n_utts > 0 ? n_utts : 1) * K;
  double *tile_stats = tile_obs + static_cast<size_t>(n_tiles > 0 ? n_tiles : 1) * (kBlock / 64) * 2 * S * D;
  float *feat_t = reinterpret_cast<float *>(tile_stats + static_cast<size_t>(n_tiles > 0 ? n_tiles : 1) * (kBlock / 64) * K);
  FbArgs a;
  a.feats = feats;
  a.offsets = offsets;
  a.slot_utt = slot_utt;
  a.tile_model = tile_model;
  a.n_tiles = n_tiles;
  a.n_slots = n_slots;
  a.pv = pack_view(pack, W, S, D);
  a.lat_b = lat_b;
  a.lat_f = lat_f;
  a.feat_t = feat_t;
  a.staged = (fast_div & SAPR_ESTEP_STAGED) != 0;
  a.loglik = loglik;
  a.stream = as_stream(stream);
  const int fast = (fast_div & SAPR_PACK_FAST_DIV) ? 1 : 0;
  int rc = 0, sub = 1;  // partial rows per tile handed to the last reduction
  if (n_tiles > 0) {
    if (D == 13 && S == 10)
      rc = launch_forward<13, 10>(a, topology, fast, max_T);
#ifndef SAPR_ONLY_13_10
    else if (D == 13 && S == 18)
      rc = launch_forward<13, 18>(a, topology, fast, max_T);
    else if (D == 39 && S == 10)
      rc = launch_forward<39, 10>(a, topology, fast, max_T);
    else if (D == 39 && S == 18)
      rc = launch_forward<39, 18>(a, topology, fast, max_T);
#endif
    else
      rc = fail(SAPR_ERR_UNSUPPORTED,
                "trellis kernels are instantiated for (D,S) in {13,39}x{10,18}; got D=%d S=%d", D, S);
    if (rc) return rc;
    static const bool split = [] {  // SAPR_ESTEP_OBS=split: round 3's smoothing pass + fb_obs_kernel over its lattice
      const char *e = getenv("SAPR_ESTEP_OBS");
      return e && e[0] == 's';
    }();
    if (topology == SAPR_TOPO_BIDIAG && !split) {
      sub = kBlock / 64;
      if (D == 13 && S == 10)
        rc = launch_smooth_obs<13, 10>(a, tile_stats, tile_obs);
#ifndef SAPR_ONLY_13_10
      else if (D == 13)
        rc = launch_smooth_obs<13, 18>(a, tile_stats, tile_obs);
      else if (S == 10)
        rc = launch_smooth_obs<39, 10>(a, tile_stats, tile_obs);
      else
        rc = launch_smooth_obs<39, 18>(a, tile_stats, tile_obs);
#endif
      if (rc) return rc;
    } else {
      rc = S == 10 ? launch_backward<10>(a, topology, utt_stats) : launch_backward<18>(a, topology, utt_stats);
      if (rc) return rc;
      const int64_t tiles_pad = round_up64(n_tiles, kXcd);
      if (D == 13) {
        constexpr int SC = 2;
        const int n_chunks = (S + SC - 1) / SC;
        SAPR_LAUNCH((fb_obs_kernel<13, SC>), dim3(static_cast<unsigned>(tiles_pad * n_chunks)), dim3(kBlock), 0,
                    a.stream, feat_t, offsets, slot_utt, n_tiles, n_slots, S, n_chunks, lat_f, tile_obs);
      } else {  // D == 39 (launch_forward rejected everything else)
        constexpr int SC = 1;
        SAPR_LAUNCH((fb_obs_kernel<39, SC>), dim3(static_cast<unsigned>(tiles_pad * S)), dim3(kBlock), 0, a.stream,
                    feat_t, offsets, slot_utt, n_tiles, n_slots, S, S, lat_f, tile_obs);
      }
      SAPR_HIP_TRY(hipGetLastError());
      SAPR_LAUNCH(fb_tile_reduce_kernel, dim3(static_cast<unsigned>(n_tiles)), dim3(kBlock), 0, a.stream, slot_utt, K,
                  utt_stats, tile_stats);
      SAPR_HIP_TRY(hipGetLastError());
    }
  }
  const int Kw = K + 2 * S * D;
  const int64_t total = static_cast<int64_t>(W) * Kw;
  SAPR_LAUNCH(fb_reduce_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, a.stream,
                     model_tile_off, W, S, D, sub, tile_stats, tile_obs, stats);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Flat start of the hmmlearn-style models (hmmlearn_hmm.py:83-94): np.mean / np.var over axis 0 of the
// concatenated float32 features.  numpy reduces a C-contiguous float32 (N, D) array along axis 0 by adding
// row after row in float32 (no pair-wise scheme on the slow axis), so each column is ONE sequential float32
// chain over all N frames — reproduced here bit for bit, which is why this is a one-workgroup kernel: the
// chain is the critical path (N x one float32 add); the other three wavefronts only keep it fed, staging the
// next tile of rows into LDS while wavefront 0 adds the current one.
//   center == NULL:  out[d] = sum_r x[r][d]                       (np.sum(X, axis=0))
//   center != NULL:  out[d] = sum_r RN32(RN32(x[r][d] - c[d])^2)  (the sum inside np.var: x = arr - mean; x*x)
// ------------------------------------------------------------------------------------------------------
namespace sapr {
namespace {
constexpr int kColsumLdsFloats = 6144;  // per buffer (24 KB); two buffers

__global__ __launch_bounds__(256) void colsum_f32_kernel(const float *__restrict__ x, int64_t n_rows, int D,
                                                         const float *__restrict__ center, float *__restrict__ out) {
  __shared__ float buf[2][kColsumLdsFloats];
  const int tid = threadIdx.x;
  const int tile_rows = kColsumLdsFloats / D;
  const int64_t n_tiles = (n_rows + tile_rows - 1) / tile_rows;
  float acc = 0.0f;
  const float c = (center && tid < D) ? center[tid] : 0.0f;
  auto stage = [&](int64_t tile, int b, int first_thread, int n_threads) {
    const int64_t r0 = tile * tile_rows;
    const int64_t rows = (r0 + tile_rows <= n_rows) ? tile_rows : (n_rows - r0);
    const int64_t n = rows * D;
    const float *src = x + r0 * D;
    for (int64_t i = tid - first_thread; i < n; i += n_threads) buf[b][i] = src[i];
  };
  if (n_tiles > 0) stage(0, 0, 0, 256);
  __syncthreads();
  for (int64_t tile = 0; tile < n_tiles; ++tile) {
    const int b = static_cast<int>(tile & 1);
    if (tid >= 64) {  // wavefronts 1-3: next tile -> the other buffer
      if (tile + 1 < n_tiles) stage(tile + 1, b ^ 1, 64, 192);
    } else if (tid < D) {  // wavefront 0, one lane per column: the sequential chain
      const int64_t r0 = tile * tile_rows;
      const int rows = static_cast<int>((r0 + tile_rows <= n_rows) ? tile_rows : (n_rows - r0));
      const float *p = buf[b] + tid;
      int r = 0;
      for (; r + 16 <= rows; r += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = p[(r + i) * D];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (center) {
            const float dlt = v[i] - c;
            acc += dlt * dlt;  // -ffp-contract=off: the product is rounded before the add, as in numpy
          } else {
            acc += v[i];
          }
        }
      }
      for (; r < rows; ++r) {
        const float v = p[r * D];
        if (center) {
          const float dlt = v - c;
          acc += dlt * dlt;
        } else {
          acc += v;
        }
      }
    }
    __syncthreads();
  }
  if (tid < D) out[tid] = acc;
}
}  // namespace
}  // namespace sapr

extern "C" int sapr_colsum_f32(const float *x, int64_t n_rows, int32_t D, const float *center, float *out,
                               void *stream) {
  SAPR_REQUIRE(x && out && n_rows >= 0 && D > 0 && D <= 64, "bad arguments (D <= 64)");
  SAPR_LAUNCH(sapr::colsum_f32_kernel, dim3(1), dim3(256), 0, sapr::as_stream(stream), x, n_rows, D, center, out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}
