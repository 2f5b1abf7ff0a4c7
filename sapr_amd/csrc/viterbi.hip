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
#include "viterbi_shared.h"

namespace sapr {
namespace {

using namespace emission;

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
  // ... and the eight states of a step leave as two 16-byte stores (a lane's path is contiguous in t; only
  // dword-aligned: unaligned dwordx4 is legal on gfx9) instead of eight 4-byte stores, each of which touches 64 cache
  // lines per wavefront
  constexpr int kAhead = 8;
  int t = T - 1;
  for (; t >= kAhead; t -= kAhead) {
    uint32_t bits[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) bits[k] = bp[static_cast<int64_t>(t - k) * n_slots];
    int st[kAhead];  // st[k] = state of frame t - k - 1
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
      s -= static_cast<int>((bits[k] >> s) & 1u);
      st[k] = s;
    }
    struct __attribute__((packed, aligned(4))) Quad {
      int32_t a, b, c, d;
    };
    Quad *dst = reinterpret_cast<Quad *>(path + beg + t - kAhead);  // frames t - 8 .. t - 1, ascending
    dst[0] = Quad{st[7], st[6], st[5], st[4]};
    dst[1] = Quad{st[3], st[2], st[1], st[0]};
  }
  for (; t >= 1; --t) {  // the first frames (fewer than eight left)
    s -= static_cast<int>((bp[static_cast<int64_t>(t) * n_slots] >> s) & 1u);
    path[beg + t - 1] = s;
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
    // as in viterbi_backtrace_pruned_kernel: the words' addresses do not depend on the state chain (eight loads in
    // flight) and a lane's path is contiguous in t (eight states leave as two 16-byte stores)
    constexpr int kAhead = 8;
    int t = T - 1;
    for (; t >= kAhead; t -= kAhead) {
      uint32_t bits[kAhead];
#pragma unroll
      for (int k = 0; k < kAhead; ++k) bits[k] = bp[static_cast<int64_t>(t - k) * n_slots];
      int st[kAhead];  // st[k] = state of frame t - k - 1
#pragma unroll
      for (int k = 0; k < kAhead; ++k) {
        s -= static_cast<int>((bits[k] >> s) & 1u);
        st[k] = s;
      }
      struct __attribute__((packed, aligned(4))) Quad {
        int32_t a, b, c, d;
      };
      Quad *dst = reinterpret_cast<Quad *>(path + beg + t - kAhead);  // frames t - 8 .. t - 1, ascending
      dst[0] = Quad{st[7], st[6], st[5], st[4]};
      dst[1] = Quad{st[3], st[2], st[1], st[0]};
    }
    for (; t >= 1; --t) {
      s -= static_cast<int>((bp[static_cast<int64_t>(t) * n_slots] >> s) & 1u);
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
                                        double *__restrict__ blob, int *__restrict__ flag) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const PackView pv = pack_view(blob, W, S, D);
  double cmax = 0.0, gcs = 0.0, lts = 0.0, lss = 0.0;
  bool off_band = false;  // a transition other than i -> i, i -> i + 1: not the bidiagonal topology (flag bit 3)
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
      if (s2 != s && s2 != s + 1 && !(l == neg_inf())) off_band = true;  // NaN counts as a transition
    }
  }
  if (off_band) atomicOr(flag, 8);
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

// First state of word w's unreachable TAIL: every state j >= J0 has start probability 0 and nothing leads into J0
// (trellis.kernel_states pads small models that way).  The matrix-core bounding lattice divides the forward weights
// out of its column (viterbi_bound.hip), which needs them finite: tail states get a zero row of P, weight 0, and are
// left out of the final maximum instead.  S when the chain has no such tail.
__device__ int unreachable_tail(const double *__restrict__ log_start, const double *__restrict__ log_trans, int S,
                                int w) {
  int j0 = S;
  for (int j = S - 1; j >= 1; --j) {
    if (log_start[static_cast<int64_t>(w) * S + j] != neg_inf()) break;
    if (log_trans[(static_cast<int64_t>(w) * S + (j - 1)) * S + j] == neg_inf()) j0 = j;
  }
  return j0;
}

// entry (state j of word w, slot i of group g) of P with the slot's feature factor divided out:
//   squared slot  -y/2 / a_sq^2     linear slot  y mu' / a_lin     constant slot (phi = 1024)  [-(c0 + gconst)/2 + sg] / 1024
// *noself_elsewhere: the state has no self-loop and sits at a chain position the kernel has no mask for
__device__ double gemm_entry(const double *__restrict__ means, const double *__restrict__ vars,
                             const double *__restrict__ gconst, const double *__restrict__ log_trans,
                             const PackView &pv, int W, int S, int D, int w, int j, int g, int i, bool *noself_elsewhere,
                             int tail /* unreachable_tail(w), computed once by the caller */) {
  const int G = gemm_groups(D), G8 = 8 * G;
  if (j >= S || j >= tail) return 0.0;
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
  const int tail = unreachable_tail(pv.log_start, log_trans, S, w);
  for (int i = 0; i < 8; ++i) {
    const double v = fabs(gemm_entry(means, vars, gconst, log_trans, pv, W, S, D, w, j, g, i, &elsewhere, tail));
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
  const int tail = unreachable_tail(pv.log_start, log_trans, S, w);
  for (int i = 0; i < 8; ++i) {
    const float v32 = static_cast<float>(gemm_entry(means, vars, gconst, log_trans, pv, W, S, D, w, j, g, i, &elsewhere, tail) * scale);
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
                                             int W, int S, int D, double *__restrict__ blob, int *__restrict__ bad) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= W) return;
  const PackView pv = pack_view(blob, W, S, D);
  const double scale = pv.gkw[2 * W];
  // forward weights divided out of the bounding lattice: R_j = sum_(1<=i<=j) (lt_(i-1)i - sg_(i-1)); a -inf inside
  // the reachable chain has no finite R: the matrix-core pass is then off (flag bit 2)
  const int j0 = unreachable_tail(pv.log_start, log_trans, S, w);
  {
    double *R = const_cast<double *>(pv.gR) + static_cast<int64_t>(w) * S;
    double *Rf = R + static_cast<int64_t>(W) * S;
    double acc = 0.0;
    for (int j = 0; j < S; ++j) {
      if (j >= 1 && j < j0) {
        double sg = log_trans[(static_cast<int64_t>(w) * S + (j - 1)) * S + (j - 1)];
        if (sg == neg_inf()) sg = 0.0;
        const double r = log_trans[(static_cast<int64_t>(w) * S + (j - 1)) * S + j] - sg;
        if (!(fabs(r) <= 1e300)) atomicOr(bad, 4);
        acc += r;
      }
      R[j] = acc;
      Rf[j] = j < j0 ? acc : neg_inf();
    }
  }
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
        row += fabs(gemm_entry(means, vars, gconst, log_trans, pv, W, S, D, w, s, g, i, &unused, j0)) * scale;
    psum = nan_max(psum, row);
  }
  const_cast<double *>(pv.gkw)[w] = k;
  const_cast<double *>(pv.gkw)[W + w] = psum;
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
  // a pack for the E-step only (a Baum-Welch loop builds one per iteration and never decodes with it): the operands
  // of the pruned decoder's bounding pass are left out and its flags stay clear
  const bool exact_only = fast_div_ok && (*fast_div_ok & SAPR_PACK_EXACT_ONLY);
  int *flag = reinterpret_cast<int *>(static_cast<double *>(pack) + pack_doubles(W, S, D));
  SAPR_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), st));
  SAPR_LAUNCH(diag_pack_kernel, dim3(64), dim3(256), 0, st, means, vars, gconst, log_start, log_trans,
                     W, S, D, static_cast<double *>(pack), flag);
  SAPR_LAUNCH(diag_pack_consts_kernel, dim3((W + 63) / 64), dim3(64), 0, st, means, vars, gconst, log_start,
              log_trans, W, S, D, static_cast<double *>(pack), flag);
  if (!exact_only) {
  SAPR_LAUNCH(diag_pack_center_kernel, dim3(1), dim3(64 * ((8 * gemm_groups(D) + 63) / 64)), 0, st, means, vars, W, S,
              D, static_cast<double *>(pack));
  const int64_t n_ent = static_cast<int64_t>(W) * S * 2 * gemm_groups(D);
  SAPR_LAUNCH(diag_pack_gemm_max_kernel, dim3(static_cast<unsigned>((n_ent + 255) / 256)), dim3(256), 0, st, means,
              vars, gconst, log_trans, W, S, D, static_cast<double *>(pack), flag);
  const int64_t n_gemm = static_cast<int64_t>(W) * gemm_rtiles(S) * gemm_kchunks(D) * 64;
  SAPR_LAUNCH(diag_pack_gemm_kernel, dim3(static_cast<unsigned>((n_gemm + 255) / 256)), dim3(256), 0, st, means, vars,
              gconst, log_trans, W, S, D, static_cast<double *>(pack), flag);
  SAPR_LAUNCH(diag_pack_gemm_consts_kernel, dim3((W + 63) / 64), dim3(64), 0, st, means, vars, gconst, log_trans, W, S,
              D, static_cast<double *>(pack), flag);
  }
  SAPR_HIP_TRY(hipGetLastError());
  int bad = 0;
  SAPR_HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, st));
  SAPR_HIP_TRY(hipStreamSynchronize(st));  // model preparation, not the data path
  if (fast_div_ok) {
    *fast_div_ok = ((bad & 1) ? 0 : SAPR_PACK_FAST_DIV) | ((bad & 2) ? 0 : SAPR_PACK_BOUND_OK) |
                   ((bad & 6) ? 0 : SAPR_PACK_GEMM_OK) | ((bad & 8) ? 0 : SAPR_PACK_BIDIAG);
    if (exact_only) *fast_div_ok = (*fast_div_ok & ~(SAPR_PACK_BOUND_OK | SAPR_PACK_GEMM_OK)) | SAPR_PACK_EXACT_ONLY;
  }
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
  SAPR_REQUIRE(sum_order == SAPR_SUM_PAIRWISE || sum_order == SAPR_SUM_TVIEW || sum_order == SAPR_SUM_SEQ, "bad sum_order");
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
  a.seq_all = sum_order == SAPR_SUM_SEQ ? 1 : 0;
  const int fast = (fast_div & SAPR_PACK_FAST_DIV) ? 1 : 0;
  if (D == 13 && S == 10) return launch_scores_13_10(a, topology, tie, sum_order, fast);
  if (D == 13 && S == 18) return launch_scores_13_18(a, topology, tie, sum_order, fast);
  if (D == 39 && S == 10) return launch_scores_39_10(a, topology, tie, sum_order, fast);
  if (D == 39 && S == 18) return launch_scores_39_18(a, topology, tie, sum_order, fast);
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
  SAPR_REQUIRE(sum_order == SAPR_SUM_PAIRWISE || sum_order == SAPR_SUM_TVIEW || sum_order == SAPR_SUM_SEQ, "bad sum_order");
  if (!(pack_flags & SAPR_PACK_BOUND_OK))
    return fail(SAPR_ERR_UNSUPPORTED, "model pack is outside the bounding pass's domain (variances in "
                                      "[1e-20, 1e20]): use sapr_viterbi_diag_scores + sapr_viterbi_backtrace");
  if (S > 32 || !(pack_flags & SAPR_PACK_BIDIAG))
    return fail(SAPR_ERR_UNSUPPORTED, "the pruned decoder covers the bidiagonal topology only (transitions i -> i and "
                                      "i -> i + 1, S <= 32; sapr_diag_pack reports SAPR_PACK_BIDIAG): use "
                                      "sapr_viterbi_diag_scores + sapr_viterbi_backtrace");
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
  a.seq_all = sum_order == SAPR_SUM_SEQ ? 1 : 0;
  int rc;
  // pass A: float32 bounds
  if (D == 13 && S == 10) rc = launch_approx_13_10(a, pv, ascore, aeps, pack_flags);
  else if (D == 13 && S == 18) rc = launch_approx_13_18(a, pv, ascore, aeps, pack_flags);
  else if (D == 39 && S == 10) rc = launch_approx_39_10(a, pv, ascore, aeps, pack_flags);
  else if (D == 39 && S == 18) rc = launch_approx_39_18(a, pv, ascore, aeps, pack_flags);
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
  if (D == 13 && S == 10) rc = launch_scores_13_10(a, SAPR_TOPO_BIDIAG, tie, sum_order, fast);
  else if (D == 13 && S == 18) rc = launch_scores_13_18(a, SAPR_TOPO_BIDIAG, tie, sum_order, fast);
  else if (D == 39 && S == 10) rc = launch_scores_39_10(a, SAPR_TOPO_BIDIAG, tie, sum_order, fast);
  else rc = launch_scores_39_18(a, SAPR_TOPO_BIDIAG, tie, sum_order, fast);
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

