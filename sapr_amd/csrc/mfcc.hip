// Batched MFCC front-end on gfx950 (MI355X): PCM in HBM -> frame-major MFCC (+delta, +delta-delta).
//
// Replaces librosa.feature.mfcc as the reference calls it (mfcc_extract.py:15-23):
//   center=True zero padding -> periodic-Hamming window (win_length, centred in n_fft) -> rFFT ->
//   |X|^2 -> Slaney mel filterbank -> 10*log10(max(1e-10, .)) -> clip at (utterance max - top_db)
//   -> orthonormal DCT-II -> first n_mfcc coefficients,
// plus BASELINE.json's north-star additions (no reference semantics): pre-emphasis and
// Savitzky-Golay delta / delta-delta.  CPU restatement: oracle/mfcc_oracle.py.
//
// One 256-thread workgroup owns one utterance (so the utterance-global top_db maximum needs no
// second pass over HBM: the log-mel matrix stays in LDS) and walks it in tiles of frames:
//
//   FFT      n_fft = 2*R*R real points are packed into an (R*R)-point complex FFT; R lanes
//            cooperate on a frame (R = 16: 4 frames per wavefront; R = 32: 2), each lane holding
//            R complex points in registers: in-lane radix-2 DIF of size R (compile-time twiddles),
//            twiddle, one R x R transpose through LDS (padded rows, conflict-free), in-lane DIF
//            again, then the real-FFT untangle with the conjugate partner fetched across lanes (DPP
//            mirror + rotate inside the 16-lane row for R = 16, ds_bpermute for R = 32).
//   mel      power tile P[frame][bin] in LDS -> v_mfma_f32_16x16x4_f32 with the filterbank as the
//            A operand, restricted to each 16-mel tile's band of non-zero bins (the triangular
//            filters are banded, so ~1.1x n_bins K-steps instead of n_mtiles x n_bins).
//   log      on the accumulator registers, written to the LDS log-mel matrix; running maximum.
//   DCT      second (tiny) MFMA over the clipped log-mel matrix.
//   deltas   9-tap Savitzky-Golay along t out of LDS (scipy mode="interp" edge polynomials).
//   store    frame-major [T][D_out] float32, fully coalesced.
//
// HBM traffic is the algorithmic minimum: PCM in once (raw buffer loads into registers, one tile
// ahead, then the LDS stage buffer all frames of the tile read from), features out once.  fp32 throughout (librosa's dtype flow is float32
// after the FFT); this file alone is built with FMA contraction enabled.
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#include "sapr_common.h"

#pragma clang fp contract(fast)

// SAPR_ABLATE (developer builds only, scripts/experiments/ablate_mfcc.sh): bit mask of phases to SKIP, to see
// which work the kernel's time follows.  Results are wrong by construction; never set in a product build.
//   1 in-lane FFTs   2 LDS transposes   4 untangle arithmetic   8 mel MFMA + log   16 workgroup barriers in the tile loop
//   32 utterance epilogue (max, DCT, deltas, store)   64 sample + window reads   128 pass-A twiddles   256 PCM staging
//   512 the filterbank MFMAs only (operand reads, log and stores stay)   1024 the log / store after the MFMAs only
#ifndef SAPR_ABLATE
#define SAPR_ABLATE 0
#endif

namespace sapr {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;
constexpr double kPi = 3.14159265358979323846264338327950288;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// compile-time twiddles
// ------------------------------------------------------------------------------------------
constexpr double c_sin_taylor(double x) {  // |x| <= pi/2
  double term = x, sum = x;
  for (int n = 1; n < 20; ++n) {
    term *= -x * x / ((2.0 * n) * (2.0 * n + 1.0));
    sum += term;
  }
  return sum;
}
constexpr double c_cos_taylor(double x) {  // |x| <= pi/2
  double term = 1.0, sum = 1.0;
  for (int n = 1; n < 20; ++n) {
    term *= -x * x / ((2.0 * n - 1.0) * (2.0 * n));
    sum += term;
  }
  return sum;
}
// cos / sin of 2*pi*j/L for 0 <= j < L/2 (first two quadrants)
constexpr double c_cos2pi(int j, int L) {
  if (4 * j <= L) return c_cos_taylor(2.0 * kPi * j / L);
  return -c_cos_taylor(2.0 * kPi * (L / 2 - j) / L);
}
constexpr double c_sin2pi(int j, int L) {
  if (4 * j <= L) return c_sin_taylor(2.0 * kPi * j / L);
  return c_sin_taylor(2.0 * kPi * (L / 2 - j) / L);
}

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// (r + i*im) *= exp(-2*pi*i*J/L), 0 <= J < L/2, trivial cases folded at compile time
template <int L, int J>
__device__ __forceinline__ void mul_twiddle(float &r, float &i) {
  if constexpr (J == 0) {
  } else if constexpr (4 * J == L) {  // * (-i)
    const float t = r;
    r = i;
    i = -t;
  } else if constexpr (8 * J == L) {  // * (1 - i)/sqrt2
    constexpr float k = 0.70710678118654752440f;
    const float t = (r + i) * k, u = (i - r) * k;
    r = t;
    i = u;
  } else if constexpr (8 * J == 3 * L) {  // * (-1 - i)/sqrt2
    constexpr float k = 0.70710678118654752440f;
    const float t = (i - r) * k, u = -(r + i) * k;
    r = t;
    i = u;
  } else {
    constexpr float c = static_cast<float>(c_cos2pi(J, L));
    constexpr float s = static_cast<float>(c_sin2pi(J, L));
    const float t = r * c + i * s, u = i * c - r * s;
    r = t;
    i = u;
  }
}

constexpr int bitrev(int v, int bits) {
  int r = 0;
  for (int b = 0; b < bits; ++b) r |= ((v >> b) & 1) << (bits - 1 - b);
  return r;
}
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

// in-lane radix-2 decimation-in-frequency FFT of size R over register arrays:
// natural-order input, X[k] is left at index bitrev(k).
template <int R, int H>
__device__ __forceinline__ void dif_stages(float (&re)[R], float (&im)[R]) {
  if constexpr (H >= 1) {
    static_for<0, R / (2 * H)>([&](auto blk_c) {
      constexpr int blk = decltype(blk_c)::value * 2 * H;
      static_for<0, H>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        constexpr int a = blk + j, b = a + H;
        const float ar = re[a], ai = im[a], br = re[b], bi = im[b];
        re[a] = ar + br;
        im[a] = ai + bi;
        float dr = ar - br, di = ai - bi;
        mul_twiddle<2 * H, j>(dr, di);
        re[b] = dr;
        im[b] = di;
      });
    });
    dif_stages<R, H / 2>(re, im);
  }
}
template <int R>
__device__ __forceinline__ void fft_inlane(float (&re)[R], float (&im)[R]) {
  dif_stages<R, R / 2>(re, im);
}

// ------------------------------------------------------------------------------------------
// plan (device tables + scalars), passed to the kernel by value
// ------------------------------------------------------------------------------------------
struct MfccDev {
  int n_fft, win_length, hop, n_mels, n_mfcc, d_out, deltas;
  int n_mtiles, lm_stride, t_pad, r_lo, r_hi, n_bins, total_ks, mel_in_lds;
  int span0, span_len, stage_floats, ksr;  // staged PCM span of a tile; register-fragment K-steps (0 = off)
  int two_pass;                            // 1: log-mel goes to HBM, mfcc_finish_kernel does clip/DCT/deltas
  float preemph, top_db, amin;
  const float *window;     // [n_fft], already scaled by 0.5 (folds the real-FFT untangle's 1/2)
  const float2 *tw_ab;     // [R][R]: exp(-2*pi*i*k1*l/(R*R)) at [k1*R + l]
  const float2 *tw_u;      // [R*R]:  exp(-i*pi*k/(R*R))
  const float *mel_frag;   // [total_ks][64] MFMA A fragments of the banded filterbank
  const unsigned *mel_frag_bf16;  // [n_mtiles][3 chunks][hi, lo][4 regs][64]: bf16 A fragments (BMEL kernels)
  int mel_bf16;            // 1: the register-fragment kernels use the split-bf16 product
  const int *mel_tiles;    // [n_mtiles][4]: {first mel, mel count (<=16), first bin (mult. of 4), first K-step}
  const float *dct_frag;   // [n_mels_pad/4][64] MFMA A fragments of the DCT rows
  const float *delta_tab;  // [2][9][9]: per order: row 0 interior taps, rows 1-4 head, 5-8 tail
  // wave-private core (mfcc_wave.h): 0 = not eligible, else quads of filterbank steps per MFMA block
  int wave_s4;
  const float *wave_a;     // [wave_s4][64][4] A operands of v_mfma_f32_4x4x1_16b_f32
  const int *wave_blk;     // [16][4] per block: first bin, head mel or -1, next block continues, next but one
};

struct MfccPlan {
  MfccDev dev;
  int R;
  size_t lds_bytes;
  void *buffer;  // one device allocation holding every table
  int wave_rlo = 0, wave_rhi = 16;  // wave-private core: window support rows as template arguments
  size_t wave_lds = 0;
  int wave_blocks_per_cu = 0;       // resident workgroups per CU (occupancy query at plan creation)
};

__host__ __device__ inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

template <int R>
struct Cfg {
  static constexpr int kNc = R * R;
  static constexpr int kNfft = 2 * R * R;
  static constexpr int kFpw = kWave / R;        // frames per wavefront per FFT set
  static constexpr int kTile = kWaves * kFpw;   // frames per tile (16 or 8)
  static constexpr int kRowPad = R + 1;         // transpose scratch row (floats)
  static constexpr int kScratchPerGroup = R * kRowPad;  // floats (one plane: re, then im)
  static constexpr int kPStride = kNc + 2;      // floats per frame row of the power tile (== 2 mod 32)
  // bf16 filterbank product (BMEL): rows of packed {bf16 hi, bf16 lo} words, == 4 mod 64 so that the
  // four ds_read_b64 of a B fragment (address = row * stride + 2 * kgroup + 8 * i) touch every bank once
  static constexpr int kPStrideB = kNc + 4;
  static constexpr int kPTail = 128;            // zeroed floats after the last row (K padding reads)
  // rows of the power tile.  The filterbank MFMA is 16 frames wide; the 2048-point core fills 8 frames per tile, and
  // (round 4) columns 8..15 re-read rows 0..7 instead of eight rows of zeros — their results were never stored — which
  // takes 33 KB off the workgroup's LDS: two-pass plans of the reference preset fit two workgroups per CU (69 KB)
  static constexpr int kPRows = kTile < 16 ? kTile : 16;
  // R == 16, float32 product ("own rows"): a wavefront's four power rows live INSIDE its own transpose scratch
  // (4 x 258 <= 4 x 272 floats), so no other wavefront's data is overwritten when it stores its powers and the
  // workgroup barrier between the FFT and the untangle phase is not needed.  Region stride == 8 mod 32 and row
  // stride == 2 mod 32 keep the filterbank's B-operand reads (16 frames x 2 k-groups per pass) conflict-free.
  static constexpr int kRegion = 1096;          // floats per wavefront region (>= kFpw * kScratchPerGroup = 1088)
};

// LDS carve, shared by host (sizing) and device (pointers); every offset is a multiple of 16
struct LdsLayout {
  int win, twab, twu, mel, dct, u, stage, lm, red, total;
};
template <int R>
__host__ __device__ inline LdsLayout lds_layout(int t_pad, int lm_stride, int total_ks, int mel_in_lds,
                                                int n_mels, int stage_floats) {
  using C = Cfg<R>;
  LdsLayout L;
  int o = 0;
  L.win = o;
  o += C::kNfft * 4;
  L.twab = o;
  o += R * R * 8;
  L.twu = o;
  o += C::kNc * 8;
  L.mel = o;
  o += mel_in_lds ? total_ks * kWave * 4 : 0;
  L.dct = o;
  o += align_up(2 * 81 * 4, 16) + 16 * 16;  // delta taps + <=15 tiles + sentinel
  L.u = o;
  const int scratch = kWaves * C::kFpw * C::kScratchPerGroup * 4;
  int ptile = (C::kPRows * C::kPStrideB + C::kPTail) * 4;  // the wider of the two row strides
  if (R == 16) ptile = (kWaves * C::kRegion + C::kPTail) * 4 > ptile ? (kWaves * C::kRegion + C::kPTail) * 4 : ptile;
  const int outb = t_pad * 16 * 4;  // MFCC staging (cepstra of the whole utterance)
  (void)n_mels;
  int u = scratch > ptile ? scratch : ptile;
  u = u > outb ? u : outb;
  o += align_up(u, 16);
  L.stage = o;
  o += align_up(stage_floats * 4, 16);
  L.lm = o;
  o += align_up(t_pad * lm_stride * 4, 16);
  L.red = o;
  o += 64;
  L.total = o;
  return L;
}

struct __attribute__((packed, aligned(4))) f4u {
  float x, y, z, w;
};

constexpr int kStagePasses = 3;  // float4 chunks per thread per tile span (span <= 3072 samples)

// One tile's PCM span [gs, gs + span_len) -> registers (kStagePasses float4 per thread) with raw buffer
// loads against a descriptor of THIS utterance's samples: offsets before the first or past the last
// sample — librosa's center=True zero padding, the end of the signal, a chunk that straddles it, the
// y[-1] := 0 start-up of the pre-emphasis — come back as 0 from the hardware bounds check (checked per
// dword: scripts/ubench/bufload_probe.hip), so the utterance edges need no code of their own.  Nothing
// consumes the registers until they are written to the LDS stage buffer a whole tile later.  gs is a
// multiple of 4 samples (even hop, tiles of 8 / 16 frames), so a chunk never straddles sample 0.
template <bool PREEMPH>
struct StageRegs {
  f4u v[kStagePasses];
  float m[PREEMPH ? kStagePasses : 1];  // sample before each chunk (pre-emphasis)
};

template <bool PREEMPH>
__device__ __forceinline__ void stage_issue(__amdgpu_buffer_rsrc_t rsrc, int gs, int n_chunks, int tid,
                                            StageRegs<PREEMPH> &sr) {
#pragma unroll
  for (int p = 0; p < kStagePasses; ++p) {
    const int c = tid + kThreads * p;
    const int cc = c < n_chunks ? c : n_chunks - 1;  // tail threads re-read a valid chunk
    const int gi = gs + 4 * cc;
    sr.v[p] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(rsrc, gi * 4, 0, 0));
    if constexpr (PREEMPH) sr.m[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (gi - 1) * 4, 0, 0));
  }
}

// registers -> LDS stage buffer, applying y'[n] = y[n] - coef*y[n-1] on the way.  Samples outside
// the signal were already zeroed, but pre-emphasis must not leak the last real sample into the
// first padded one: positions >= n_samp stay zero.
template <bool PREEMPH>
__device__ __forceinline__ void stage_write(const StageRegs<PREEMPH> &sr, float *__restrict__ s_stage, int gs,
                                            int n_samp, int n_chunks, bool inside, int tid, float coef) {
#pragma unroll
  for (int p = 0; p < kStagePasses; ++p) {
    const int c = tid + kThreads * p;
    if (c < n_chunks) {
      f4u t = sr.v[p];
      if constexpr (PREEMPH) {
        const float m = sr.m[p];
        f4u o;
        o.x = t.x - coef * m;
        o.y = t.y - coef * t.x;
        o.z = t.z - coef * t.y;
        o.w = t.w - coef * t.z;
        if (!inside) {
          const int gi = gs + 4 * c;
          o.x = (gi >= 0 && gi < n_samp) ? o.x : 0.f;
          o.y = (gi + 1 >= 0 && gi + 1 < n_samp) ? o.y : 0.f;
          o.z = (gi + 2 >= 0 && gi + 2 < n_samp) ? o.z : 0.f;
          o.w = (gi + 3 >= 0 && gi + 3 < n_samp) ? o.w : 0.f;
        }
        t = o;
      }
      *reinterpret_cast<float4 *>(s_stage + 4 * c) = make_float4(t.x, t.y, t.z, t.w);
    }
  }
}

// Value of `v` in the lane that holds the conjugate partner: lane l of a frame's group reads lane
// (R - l) % R.  For R = 16 a group is one DPP row, and "mirror, then rotate right by one" is exactly
// that permutation — two VALU moves instead of a ds_bpermute, which costs the (busier) LDS pipe three
// times as much as a plain ds_read_b32 (scripts/ubench/lat_probe.hip).  R = 32 keeps the bpermute.
template <int R>
__device__ __forceinline__ float partner16(float v, int src_lane) {
  if constexpr (R == 16) {
    const int m = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140 /* row_mirror */, 0xf, 0xf, false);
    return __int_as_float(__builtin_amdgcn_update_dpp(0, m, 0x121 /* row_ror:1 */, 0xf, 0xf, false));
  } else {
    return __shfl(v, src_lane, kWave);
  }
}

#include "mfcc_wave.h"

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
// STAMP: diagnostic build only (sapr_mfcc_batch_stamped): per-phase s_memtime sums per wavefront.
#define SAPR_STAMP(slot)                                            \
  if constexpr (STAMP) {                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
    __builtin_amdgcn_s_waitcnt(0xc07f);                             \
    st_acc[slot] += now_ - st_last;                                 \
    st_last = now_;                                                 \
  }

// KSR > 0: the workgroup's four wavefronts each own ONE mel tile for the whole launch and keep its
// KSR MFMA A-fragments in registers (host guarantees n_mtiles <= 4 and <= KSR K-steps per tile);
// KSR == 0: fragments come from LDS (MEL_LDS) or L1/L2.
// BMEL (with KSR > 0): the filterbank product runs on v_mfma_f32_16x16x32_bf16 with both operands split
// into two bf16 words (x = hi + lo exactly to 16 significant bits): acc += Ah*Bh + Ah*Bl + Al*Bh, three
// 16-cycle MFMAs per 32 bins instead of eight 32-cycle float32 ones.  The power tile then holds packed
// {hi, lo} words, made where the power is computed.  |dMFCC| vs the float32 product: < 1e-4 (emulated in
// float64 and measured, tests/test_mfcc_gpu.py), inside the front-end's 3e-3 contract.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_hi_lo_bf16(float p) {
  const unsigned pb = __float_as_uint(p);
  const float hi = __uint_as_float(pb & 0xFFFF0000u);   // top 8 significant bits (truncation: p - hi is exact)
  const __bf16 lo = static_cast<__bf16>(p - hi);        // next 8, rounded
  return (pb & 0xFFFF0000u) | static_cast<unsigned>(__builtin_bit_cast(unsigned short, lo));
}

template <int R, bool PREEMPH, bool MEL_LDS, int KSR, bool STAMP = false, bool TWO_PASS = false, bool BMEL = false>
__global__ __launch_bounds__(kThreads, (R == 16 && KSR > 0) ? 3 : 1) void mfcc_kernel(
    const float *__restrict__ pcm, const int64_t *__restrict__ sample_offsets,
    const int64_t *__restrict__ frame_offsets, int64_t n_utts, MfccDev P, float *__restrict__ out,
    unsigned long long *__restrict__ stamps = nullptr, float *__restrict__ lm_out = nullptr,
    unsigned *__restrict__ gmax_out = nullptr) {
  unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = 0;
  if constexpr (STAMP) {
    st_last = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
  using C = Cfg<R>;
  constexpr bool kOwnRows = (R == 16) && !BMEL;  // power rows inside the owning wavefront's scratch region
  constexpr int kBits = ilog2(R);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const LdsLayout L = lds_layout<R>(TWO_PASS ? 0 : P.t_pad, P.lm_stride, P.total_ks, MEL_LDS ? 1 : 0, P.n_mels,
                                    P.stage_floats);
  float *s_win = reinterpret_cast<float *>(smem + L.win);
  float2 *s_twab = reinterpret_cast<float2 *>(smem + L.twab);
  float2 *s_twu = reinterpret_cast<float2 *>(smem + L.twu);
  float *s_mel = reinterpret_cast<float *>(smem + L.mel);
  float *s_dtab = reinterpret_cast<float *>(smem + L.dct);
  int *s_tiles = reinterpret_cast<int *>(s_dtab + align_up(2 * 81, 4));
  float *s_scr = reinterpret_cast<float *>(smem + L.u);
  float *s_pt = reinterpret_cast<float *>(smem + L.u);
  float *s_out = reinterpret_cast<float *>(smem + L.u);
  float *s_stage = reinterpret_cast<float *>(smem + L.stage);
  float *s_lm = reinterpret_cast<float *>(smem + L.lm);
  float *s_red = reinterpret_cast<float *>(smem + L.red);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);  // scalar: tile tables via s_load
  const int lane = tid % kWave;
  const int grp = lane / R;   // frame slot inside the wavefront
  const int l = lane % R;     // lane inside the frame's group
  const int q = lane >> 4;    // MFMA: k index (A/B), row quad (C/D)
  const int j16 = lane & 15;  // MFMA: row (A), column (B, C/D)

  // ---- tables -> LDS (once per workgroup; workgroups are persistent over utterances) ----
  for (int i = tid; i < C::kNfft; i += kThreads) s_win[i] = P.window[i];
  for (int i = tid; i < R * R; i += kThreads) s_twab[i] = P.tw_ab[i];
  for (int i = tid; i < C::kNc; i += kThreads) s_twu[i] = P.tw_u[i];
  if constexpr (MEL_LDS)
    for (int i = tid; i < P.total_ks * kWave; i += kThreads) s_mel[i] = P.mel_frag[i];
  for (int i = tid; i < 2 * 81; i += kThreads) s_dtab[i] = P.delta_tab[i];
  for (int i = tid; i < 4 * (P.n_mtiles + 1); i += kThreads) s_tiles[i] = P.mel_tiles[i];
  if constexpr (kOwnRows) {
    // words the tile loop never writes but the filterbank's K padding may read (times a zero coefficient) must
    // hold finite values: the padding column of the transpose rows, the tail of every region and the floats
    // behind the last one.  Everything written later (FFT intermediates, powers, cepstra) is finite.
    for (int i = tid; i < kWaves * C::kRegion + C::kPTail; i += kThreads) s_pt[i] = 0.f;
  }
  __syncthreads();

  // register-resident filterbank fragments of this wavefront's mel tile
  static_assert(!BMEL || (KSR == 24 && R == 16), "the bf16 product is laid out for 3 chunks of 32 bins");
  constexpr int kPS = BMEL ? C::kPStrideB : C::kPStride;  // row stride of the power tile
  // power row of frame slot f (0..15): float offset from s_pt
  auto row_off = [](int f) {
    return kOwnRows ? (f >> 2) * C::kRegion + (f & 3) * C::kPStride : (f & (C::kPRows - 1)) * kPS;
  };
  float afr[KSR > 0 ? KSR : 1];   // BMEL: the same 24 registers hold [chunk][hi, lo][4] packed bf16 pairs
  int my_mel0 = 0, my_mcnt = 0, my_kbeg = 0;
  if constexpr (KSR > 0) {
    const bool has = wave < P.n_mtiles;
    const int mt = has ? wave : 0;
    my_mel0 = s_tiles[4 * mt + 0];
    my_mcnt = has ? s_tiles[4 * mt + 1] : 0;
    my_kbeg = s_tiles[4 * mt + 2];
    const int ks0 = s_tiles[4 * mt + 3], nks = s_tiles[4 * (mt + 1) + 3] - ks0;
    if constexpr (BMEL) {
#pragma unroll
      for (int i = 0; i < KSR; ++i)
        afr[i] = has ? __uint_as_float(P.mel_frag_bf16[(mt * KSR + i) * kWave + lane]) : 0.f;
    } else {
#pragma unroll
      for (int ks = 0; ks < KSR; ++ks) afr[ks] = (has && ks < nks) ? P.mel_frag[(ks0 + ks) * kWave + lane] : 0.f;
    }
  }

  const float neg_floor = -3.0e38f;
  const int fslot = wave * C::kFpw + grp;  // column of the power tile this lane's group fills
  const int n_chunks = (P.span_len + 3) / 4;

  for (int64_t u = blockIdx.x; u < n_utts; u += gridDim.x) {
    const int64_t s_beg = sample_offsets[u];
    const int n_samp = static_cast<int>(sample_offsets[u + 1] - s_beg);
    const int64_t f_beg = frame_offsets[u];
    const int T = static_cast<int>(frame_offsets[u + 1] - f_beg);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pcm + s_beg), 0, n_samp * 4, 0x00020000 /* raw dword buffer */);
    float run_max = neg_floor;

    // does tile t0's staged span lie inside the signal (no padding; one extra sample for pre-emphasis)?
    auto span_gs = [&](int t0) { return t0 * P.hop + P.span0; };
    auto span_inside = [&](int t0) {
      const int gs = span_gs(t0);
      return gs - (PREEMPH ? 1 : 0) >= 0 && gs + 4 * n_chunks <= n_samp;
    };

    // prologue: tile 0 -> LDS stage, tile 1 -> registers
    StageRegs<PREEMPH> pre;
    bool pre_inside = span_inside(0);
    stage_issue<PREEMPH>(rsrc, span_gs(0), n_chunks, tid, pre);
    stage_write<PREEMPH>(pre, s_stage, span_gs(0), n_samp, n_chunks, pre_inside, tid, P.preemph);
    if (C::kTile < T) {
      pre_inside = span_inside(C::kTile);
      stage_issue<PREEMPH>(rsrc, span_gs(C::kTile), n_chunks, tid, pre);
    }
    __syncthreads();

    for (int tile0 = 0; tile0 < T; tile0 += C::kTile) {
      // =========================== FFT of this wavefront's frames ===========================
      float re[R], im[R];
      SAPR_STAMP(0)  // loop overhead / previous barrier
      {
        // samples of frame `fslot` of the staged span: row r of the n_fft-frame starts 2R*(r-r_lo)
        // floats into the frame's slice; rows outside the window support re-read a valid row.
        // (Fetching them one tile ahead, under the filterbank phase, was tried in round 2: the 32 carried
        // registers spill at the 168-VGPR budget of three wavefronts per SIMD and the kernel gets slower.)
        const float *fs = s_stage + fslot * P.hop + 2 * l;
        float2 ys[R], ws[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int rc = r < P.r_lo ? P.r_lo : (r >= P.r_hi ? P.r_hi - 1 : r);
          if constexpr (SAPR_ABLATE & 64) {
            ys[r] = make_float2(0.001f * (r + lane), 0.002f * (r - lane));
            ws[r] = make_float2(0.5f, 0.25f);
          } else {
            ys[r] = *reinterpret_cast<const float2 *>(fs + 2 * R * (rc - P.r_lo));
            ws[r] = *reinterpret_cast<const float2 *>(&s_win[2 * (R * r + l)]);
          }
        }
        // all 2R LDS reads are issued before anything waits on them (the scheduler otherwise pairs
        // each read with its multiply and exposes the LDS latency R times)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          re[r] = ys[r].x * ws[r].x;
          im[r] = ys[r].y * ws[r].y;
        }
      }
      SAPR_STAMP(1)  // stage + window reads
      // pass A: FFT over n1 (register index); result for k1 sits at bitrev(k1)
      if constexpr (!(SAPR_ABLATE & 1)) fft_inlane<R>(re, im);
      // twiddle W_{Nc}^{l*k1}, then the R x R transpose through LDS: scratch[group][k1][l], real
      // and imaginary planes one after the other (halves the scratch).  The exchange stays inside
      // one wavefront (a frame's R lanes), whose DS instructions execute in order: only the
      // compiler has to be told not to reorder across the plane boundaries.
      float *scr = s_scr + (kOwnRows ? wave * C::kRegion + grp * C::kScratchPerGroup
                                      : (wave * C::kFpw + grp) * C::kScratchPerGroup);
      if constexpr (!(SAPR_ABLATE & 128))
      static_for<0, R>([&](auto k1_c) {
        constexpr int k1 = decltype(k1_c)::value;
        constexpr int p = bitrev(k1, kBits);
        const float2 w = s_twab[k1 * R + l];
        const float tr = re[p] * w.x - im[p] * w.y;
        const float ti = re[p] * w.y + im[p] * w.x;
        re[p] = tr;
        im[p] = ti;
      });
      if constexpr (!(SAPR_ABLATE & 2)) {
      static_for<0, R>([&](auto k1_c) {
        constexpr int k1 = decltype(k1_c)::value;
        scr[k1 * C::kRowPad + l] = re[bitrev(k1, kBits)];
      });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int n2 = 0; n2 < R; ++n2) re[n2] = scr[l * C::kRowPad + n2];  // lane l now plays k1
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      static_for<0, R>([&](auto k1_c) {
        constexpr int k1 = decltype(k1_c)::value;
        scr[k1 * C::kRowPad + l] = im[bitrev(k1, kBits)];
      });
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int n2 = 0; n2 < R; ++n2) im[n2] = scr[l * C::kRowPad + n2];
      }
      // pass B: FFT over n2; Z[l + R*k2] sits at bitrev(k2)
      if constexpr (!(SAPR_ABLATE & 1)) fft_inlane<R>(re, im);

      SAPR_STAMP(2)  // FFT A + transpose + FFT B
      // every wavefront is done with the stage buffer and the transpose scratch.  Own-rows layout with the
      // staging moved behind barrier 2 (register-fragment kernels): nothing a wavefront does before barrier 2
      // touches memory another wavefront still reads, and the barrier is dropped.
      if constexpr (!(SAPR_ABLATE & 16) && !(kOwnRows && KSR > 0)) __syncthreads();
      SAPR_STAMP(3)  // barrier 1

      // next tile's samples (loaded a whole tile ago) -> LDS stage, then tile i+2's loads.  The register-fragment
      // kernels do both later, behind the filterbank MFMAs (whose latency the wavefront would otherwise sit out);
      // any point between barrier 1 (every wavefront has read this tile's samples) and barrier 3 is legal.
      auto stage_next = [&]() {
        if (!(SAPR_ABLATE & 256) && tile0 + C::kTile < T)
          stage_write<PREEMPH>(pre, s_stage, span_gs(tile0 + C::kTile), n_samp, n_chunks, pre_inside, tid, P.preemph);
        const int nt0 = tile0 + 2 * C::kTile;
        if (!(SAPR_ABLATE & 256) && nt0 < T) {  // two tiles of latency cover
          pre_inside = span_inside(nt0);
          stage_issue<PREEMPH>(rsrc, span_gs(nt0), n_chunks, tid, pre);
        }
      };
      if constexpr (KSR == 0) stage_next();

      // ===================== untangle to the real spectrum, power -> LDS ====================
      // X[k] = E + W_k O and X[Nc-k] = conj(E - W_k O) share E and W_k O, so each lane does the
      // R/2 bins k = l + R*k2 < Nc/2 and also writes the mirror bin Nc-k.
      {
        float *prow = s_pt + row_off(fslot);
        const int src_lane = (lane - l) + ((R - l) % R);
        auto put = [&](int k, float p) {
          if constexpr (BMEL)
            prow[k] = __uint_as_float(pack_hi_lo_bf16(p));
          else
            prow[k] = p;
        };
        float pr[R / 2], pi[R / 2];
        // conjugate partners first (all R cross-lane fetches in flight together): lane l > 0 needs
        // logical register R-1-k2 of lane R-l; lane 0 is its own partner with register (R-k2)%R
        float2 tw[R / 2];
        static_for<0, R / 2>([&](auto k2_c) {
          constexpr int k2 = decltype(k2_c)::value;
          constexpr int p_other = bitrev(R - 1 - k2, kBits);
          pr[k2] = partner16<R>(re[p_other], src_lane);
          pi[k2] = partner16<R>(im[p_other], src_lane);
          tw[k2] = s_twu[l + R * k2];
        });
        __builtin_amdgcn_sched_barrier(0);  // every partner fetch / twiddle read in flight before the first use
        static_for<0, R / 2>([&](auto k2_c) {
          constexpr int k2 = decltype(k2_c)::value;
          constexpr int pz = bitrev(k2, kBits);
          constexpr int p_self0 = bitrev((R - k2) % R, kBits);
          if constexpr (SAPR_ABLATE & 4) {
            put(l + R * k2, re[pz] + pr[k2] + tw[k2].x);
            put(C::kNc - (l + R * k2), im[pz] + pi[k2] + tw[k2].y);
            return;
          }
          const float prr = (l == 0) ? re[p_self0] : pr[k2];
          const float pii = (l == 0) ? im[p_self0] : pi[k2];
          const float zr = re[pz], zi = im[pz];
          const float er = zr + prr, ei = zi - pii;    // E (window carries the 1/2)
          const float o_r = zi + pii, o_i = prr - zr;  // O = (Z - conj Zp)/(2i)
          const float2 w = tw[k2];
          const float wr = w.x * o_r - w.y * o_i, wi = w.x * o_i + w.y * o_r;
          const float ar = er + wr, ai = ei + wi, br = er - wr, bi = ei - wi;
          const int k = l + R * k2;
          put(k, ar * ar + ai * ai);  // columns of frames >= T hold the last tile's leftovers: never stored
          put(C::kNc - k, br * br + bi * bi);  // k == 0: the Nyquist bin
        });
        if (l == 0) {  // the self-paired middle bin Nc/2: X = 2 Re Z' - i 2 Im Z'
          constexpr int pm = bitrev(R / 2, kBits);
          const float zr = re[pm], zi = im[pm];
          put(C::kNc / 2, 4.f * (zr * zr + zi * zi));
          prow[C::kNc + 1] = 0.f;
        }
        if constexpr (BMEL) {  // the row's pad words are read as K padding: keep them finite
          if (l < 2) prow[C::kNc + 2 + l] = 0.f;
        }
        if (!kOwnRows && tid < C::kPTail) s_pt[C::kPRows * kPS + tid] = 0.f;  // K padding read past the last row
      }
      SAPR_STAMP(4)  // stage write + untangle + power

      SAPR_STAMP(5)  // (staging moved: see stage_next)
      if constexpr (!(SAPR_ABLATE & 16)) __syncthreads();
      SAPR_STAMP(6)  // barrier 2

      if constexpr (SAPR_ABLATE & 8) {
        if (tid < 64) s_lm[(tile0 + (tid & 15)) * P.lm_stride + (tid >> 4)] = s_pt[tid * 7];
      } else {
      // ============================ mel filterbank on the MFMA ==============================
      if constexpr (KSR > 0) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BMEL) {
          // B fragment of chunk c, lane (frame j16, k-group q): element e <-> bin kbeg + 32c + 8(e/2) + 2q + e%2
          // (any k order works as long as the A fragments use the same one; this one makes each of the four
          // ds_read_b64 conflict-free).  Words are {hi, lo}: two byte-permutes split a pair of them.
          const float *brow = s_pt + j16 * kPS + my_kbeg + 2 * q;
          float2 w[3][4];
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) w[c][i] = *reinterpret_cast<const float2 *>(brow + 32 * c + 8 * i);
          __builtin_amdgcn_sched_barrier(0);  // all B reads in flight
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            u32x4 bh, bl, ah, al;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const unsigned w0 = __float_as_uint(w[c][i].x), w1 = __float_as_uint(w[c][i].y);
              bh[i] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);  // {hi(w1), hi(w0)}
              bl[i] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);  // {lo(w1), lo(w0)}
              ah[i] = __float_as_uint(afr[8 * c + i]);
              al[i] = __float_as_uint(afr[8 * c + 4 + i]);
            }
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Al = __builtin_bit_cast(bf16x8, al);
            const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bl = __builtin_bit_cast(bf16x8, bl);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, acc1, 0, 0, 0);
          }
        } else {
        const float *brow = s_pt + row_off(j16) + my_kbeg + q;
        float bv[KSR];
#pragma unroll
        for (int ks = 0; ks < KSR; ++ks) bv[ks] = brow[4 * ks];
        __builtin_amdgcn_sched_barrier(0);  // all B reads in flight, then the MFMAs back to back
        static_for<0, KSR / 2>([&](auto h_c) {
          constexpr int ks = 2 * decltype(h_c)::value;
          if constexpr (SAPR_ABLATE & 512) {
            acc0[ks % 4] += afr[ks] * bv[ks];
            acc1[ks % 4] += afr[ks + 1] * bv[ks + 1];
          } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[ks], bv[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[ks + 1], bv[ks + 1], acc1, 0, 0, 0);
          }
        });
        }
        stage_next();  // LDS writes + global loads under the matrix pipe's latency
        const int t = tile0 + j16;
        if constexpr (SAPR_ABLATE & 1024) {
          if (lane == 0) s_lm[(tile0 + wave) * P.lm_stride] = acc0[0] + acc1[1] + acc0[2] + acc1[3];
        } else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int mi = 4 * q + i;
          const float v = 10.0f * __log10f(fmaxf(P.amin, acc0[i] + acc1[i]));
          if (j16 < C::kTile && t < T && mi < my_mcnt) {
            if constexpr (TWO_PASS)
              lm_out[(f_beg + t) * P.n_mels + my_mel0 + mi] = v;
            else
              s_lm[t * P.lm_stride + my_mel0 + mi] = v;
            run_max = fmaxf(run_max, v);
          }
        }
      } else {
        for (int mt = wave; mt < P.n_mtiles; mt += kWaves) {
          const int4 ti = *reinterpret_cast<const int4 *>(s_tiles + 4 * mt);
          const int mel0 = ti.x, mcnt = ti.y, kbeg = ti.z, ks0 = ti.w;
          const int nks = s_tiles[4 * (mt + 1) + 3] - ks0;  // sentinel entry at [n_mtiles]
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          const float *afrag = (MEL_LDS ? s_mel : P.mel_frag) + ks0 * kWave + lane;
          const float *brow = s_pt + row_off(j16) + kbeg + q;
          // K-steps are padded to a multiple of 4 on the host; two accumulators break the
          // 40-cycle dependent-MFMA latency
          for (int ks = 0; ks < nks; ks += 4) {
            const float a0 = afrag[(ks + 0) * kWave], a1 = afrag[(ks + 1) * kWave];
            const float a2 = afrag[(ks + 2) * kWave], a3 = afrag[(ks + 3) * kWave];
            const float b0 = brow[4 * ks], b1 = brow[4 * ks + 4], b2 = brow[4 * ks + 8], b3 = brow[4 * ks + 12];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc1, 0, 0, 0);
          }
          // log on the accumulator: rows = mel0 + 4*q + i, column = frame j16
          const int t = tile0 + j16;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int mi = 4 * q + i;
            const float v = 10.0f * __log10f(fmaxf(P.amin, acc0[i] + acc1[i]));
            if (j16 < C::kTile && t < T && mi < mcnt) {
              if constexpr (TWO_PASS)
                lm_out[(f_beg + t) * P.n_mels + mel0 + mi] = v;
              else
                s_lm[t * P.lm_stride + mel0 + mi] = v;
              run_max = fmaxf(run_max, v);
            }
          }
        }
      }
      }
      SAPR_STAMP(7)  // mel MFMA + log
      if constexpr (!(SAPR_ABLATE & 16)) __syncthreads();  // power tile is free again
      SAPR_STAMP(8)  // barrier 3
    }

    if constexpr (SAPR_ABLATE & 32) {
      if (tid == 0) out[f_beg * P.d_out] = run_max + s_lm[tid];
      __syncthreads();
      continue;
    }
    // ===================== utterance-global maximum (top_db reference) ======================
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) run_max = fmaxf(run_max, __shfl_xor(run_max, o, kWave));
    if (lane == 0) s_red[wave] = run_max;
    __syncthreads();
    float gmax = s_red[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) gmax = fmaxf(gmax, s_red[w]);
    if constexpr (TWO_PASS) {  // clip / DCT / deltas happen in mfcc_finish_kernel
      if (tid == 0) gmax_out[u] = enc_ordered(gmax);
      __syncthreads();
      continue;
    }
    const float floor_db = gmax - P.top_db;
    SAPR_STAMP(9)  // utterance max

    // ================================== DCT on the MFMA =====================================
    // A fragments straight from L1/L2 (2.5 KB, shared by every workgroup), four K-steps of loads in
    // flight at a time; the power tile / scratch region is free (barrier above) and receives the cepstra
    const int n_ks = align_up(P.n_mels, 16) / 4;  // multiple of 4
    for (int nt = wave; nt * 16 < T; nt += kWaves) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int t = nt * 16 + j16;
      const int tc = t < T ? t : T - 1;
      const float *lrow = s_lm + tc * P.lm_stride + q;
      for (int ks = 0; ks < n_ks; ks += 4) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = P.dct_frag[(ks + i) * kWave + lane];
          const int mel = 4 * (ks + i) + q;
          b[i] = mel < P.n_mels ? fmaxf(lrow[4 * (ks + i)], floor_db) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
      }
      if (t < T) *reinterpret_cast<f32x4 *>(&s_out[t * 16 + 4 * q]) = acc;
    }
    __syncthreads();
    SAPR_STAMP(10)  // DCT

    // ============================== deltas + coalesced store ================================
    {
      float *__restrict__ o = out + f_beg * P.d_out;
      const int total = T * P.d_out;
      for (int e = tid; e < total; e += kThreads) {
        const int t = e / P.d_out;
        const int c = e - t * P.d_out;
        float v;
        if (c < P.n_mfcc) {
          v = s_out[t * 16 + c];
        } else {
          const int order = c / P.n_mfcc;  // 1 or 2
          const int cc = c - order * P.n_mfcc;
          const float *tab = s_dtab + (order - 1) * 81;
          int row, t0;
          if (t < 4) {
            row = 1 + t;
            t0 = 0;
          } else if (t >= T - 4) {
            row = 5 + (t - (T - 4));
            t0 = T - 9;
          } else {
            row = 0;
            t0 = t - 4;
          }
          v = 0.f;
#pragma unroll
          for (int k = 0; k < 9; ++k) v += tab[row * 9 + k] * s_out[(t0 + k) * 16 + cc];
        }
        o[e] = v;
      }
    }
    __syncthreads();  // s_out / s_lm / stage are reused by the next utterance
    SAPR_STAMP(11)  // deltas + store
  }
  if constexpr (STAMP) {
    if (lane == 0)
      for (int i = 0; i < 12; ++i) stamps[(static_cast<int64_t>(blockIdx.x) * kWaves + wave) * 12 + i] = st_acc[i];
  }
}
#undef SAPR_STAMP

// ------------------------------------------------------------------------------------------
// two-pass mode, second kernel: log-mel [total_frames][n_mels] + per-utterance maximum ->
// top_db clip, DCT-II, delta / delta-delta, frame-major store.  Utterances of any length are
// walked in chunks of kFinChunk frames (+-8 frames of halo for the 9-tap delta filters).
// ------------------------------------------------------------------------------------------
constexpr int kFinChunk = 96, kFinHalo = 8, kFinRows = kFinChunk + 2 * kFinHalo;  // 112 rows = 7 MFMA tiles

__global__ __launch_bounds__(kThreads) void mfcc_finish_kernel(const float *__restrict__ lm,
                                                               const unsigned *__restrict__ gmax,
                                                               const int64_t *__restrict__ frame_offsets,
                                                               int64_t n_utts, MfccDev P, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *s_dtab = reinterpret_cast<float *>(smem);        // [2][81]
  float *s_c = s_dtab + align_up(2 * 81, 4);              // [kFinRows][16] cepstra
  float *s_l = s_c + kFinRows * 16;                       // [kFinRows][lm_stride] clipped log-mel
  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int q = lane >> 4, j16 = lane & 15;
  for (int i = tid; i < 2 * 81; i += kThreads) s_dtab[i] = P.delta_tab[i];
  __syncthreads();
  const int n_ks = align_up(P.n_mels, 16) / 4;  // multiple of 4
  for (int64_t u = blockIdx.x; u < n_utts; u += gridDim.x) {
    const int64_t f_beg = frame_offsets[u];
    const int T = static_cast<int>(frame_offsets[u + 1] - f_beg);
    const float floor_db = dec_ordered(gmax[u]) - P.top_db;
    const int chunk = T <= kFinRows ? T : kFinChunk;  // a short utterance is one chunk, no halo
    for (int c0 = 0; c0 < T; c0 += chunk) {
      const int c1 = c0 + chunk < T ? c0 + chunk : T;
      const int r0 = c0 - kFinHalo > 0 ? c0 - kFinHalo : 0;       // first staged frame
      const int r1 = c1 + kFinHalo < T ? c1 + kFinHalo : T;       // one past the last staged frame
      const int rows = r1 - r0;
      // clipped log-mel rows -> LDS (coalesced read of rows * n_mels contiguous floats)
      const float *src = lm + (f_beg + r0) * P.n_mels;
      for (int i = tid; i < rows * P.n_mels; i += kThreads) {
        const int r = i / P.n_mels, m = i - r * P.n_mels;
        s_l[r * P.lm_stride + m] = fmaxf(src[i], floor_db);
      }
      __syncthreads();
      // DCT-II on the MFMA, 16 rows per wavefront and round, A fragments from L2 (as in mfcc_kernel)
      for (int nt = wave; nt * 16 < rows; nt += kWaves) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int r = nt * 16 + j16;
        const int rc = r < rows ? r : rows - 1;
        const float *lrow = s_l + rc * P.lm_stride + q;
        for (int ks = 0; ks < n_ks; ks += 4) {
          float a[4], bb[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            a[i] = P.dct_frag[(ks + i) * kWave + lane];
            const int mel = 4 * (ks + i) + q;
            bb[i] = mel < P.n_mels ? lrow[4 * (ks + i)] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb[i], acc, 0, 0, 0);
        }
        if (r < rows) *reinterpret_cast<f32x4 *>(&s_c[r * 16 + 4 * q]) = acc;
      }
      __syncthreads();
      float *__restrict__ o = out + (f_beg + c0) * P.d_out;
      const int total = (c1 - c0) * P.d_out;
      for (int e = tid; e < total; e += kThreads) {
        const int tl = e / P.d_out, c = e - tl * P.d_out;
        const int t = c0 + tl;
        float v;
        if (c < P.n_mfcc) {
          v = s_c[(t - r0) * 16 + c];
        } else {
          const int order = c / P.n_mfcc, cc = c - order * P.n_mfcc;
          const float *tab = s_dtab + (order - 1) * 81;
          int row, t0;
          if (t < 4) {
            row = 1 + t;
            t0 = 0;
          } else if (t >= T - 4) {
            row = 5 + (t - (T - 4));
            t0 = T - 9;
          } else {
            row = 0;
            t0 = t - 4;
          }
          v = 0.f;
#pragma unroll
          for (int k = 0; k < 9; ++k) v += tab[row * 9 + k] * s_c[(t0 + k - r0) * 16 + cc];
        }
        o[e] = v;
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------
// host: table construction (float64 maths, float32 tables — librosa's dtype flow)
// ------------------------------------------------------------------------------------------
double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// librosa.filters.mel(htk=False, norm="slaney", dtype=float32) -> [n_mels][n_bins]
std::vector<float> mel_filterbank(double sr, int n_fft, int n_mels, double fmin, double fmax) {
  const int nb = 1 + n_fft / 2;
  std::vector<double> mel_f(n_mels + 2);
  const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
  for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
  std::vector<float> w(static_cast<size_t>(n_mels) * nb, 0.f);
  for (int i = 0; i < n_mels; ++i) {
    const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
    const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
    for (int b = 0; b < nb; ++b) {
      const double f = b * sr / n_fft;
      const double lower = -(mel_f[i] - f) / fd0, upper = (mel_f[i + 2] - f) / fd1;
      const double v = std::fmax(0.0, std::fmin(lower, upper));
      const float v32 = static_cast<float>(v);  // weights array is float32 ...
      w[static_cast<size_t>(i) * nb + b] = static_cast<float>(static_cast<double>(v32) * enorm);  // ... *= enorm
    }
  }
  return w;
}

// Savitzky-Golay derivative coefficients: fit a degree-`order` polynomial to 9 points at
// x = 0..8 and evaluate its `order`-th derivative at x = pos  (scipy savgol_coeffs(use="dot")).
void savgol_row(int order, double pos, double *out9) {
  // normal equations (order <= 2): A[i][p] = (i - pos)^p ; coefficient of x^order * order!
  const int n = 9, m = order + 1;
  double AtA[3][3] = {{0}}, inv[3][3];
  for (int i = 0; i < n; ++i) {
    double pw[3] = {1.0, i - pos, (i - pos) * (i - pos)};
    for (int a = 0; a < m; ++a)
      for (int b = 0; b < m; ++b) AtA[a][b] += pw[a] * pw[b];
  }
  // invert m x m (m <= 3) by Gauss-Jordan
  double aug[3][6];
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) {
      aug[a][b] = AtA[a][b];
      aug[a][m + b] = a == b ? 1.0 : 0.0;
    }
  for (int c = 0; c < m; ++c) {
    int piv = c;
    for (int r2 = c + 1; r2 < m; ++r2)
      if (std::fabs(aug[r2][c]) > std::fabs(aug[piv][c])) piv = r2;
    for (int b = 0; b < 2 * m; ++b) std::swap(aug[c][b], aug[piv][b]);
    const double d = aug[c][c];
    for (int b = 0; b < 2 * m; ++b) aug[c][b] /= d;
    for (int r2 = 0; r2 < m; ++r2)
      if (r2 != c) {
        const double f = aug[r2][c];
        for (int b = 0; b < 2 * m; ++b) aug[r2][b] -= f * aug[c][b];
      }
  }
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < m; ++b) inv[a][b] = aug[a][m + b];
  const double fact = order == 2 ? 2.0 : 1.0;
  for (int i = 0; i < n; ++i) {
    double pw[3] = {1.0, i - pos, (i - pos) * (i - pos)};
    double c = 0.0;
    for (int b = 0; b < m; ++b) c += inv[order][b] * pw[b];
    out9[i] = fact * c;
  }
}

constexpr int kKsr = 24;  // register-resident filterbank fragments per wavefront (bench-style plans)

template <int R, bool PRE, bool MLDS, int KSR, bool TWO, bool BMEL = false>
hipError_t launch_one(const MfccPlan &pl, const float *pcm, const int64_t *so, const int64_t *fo,
                      int64_t n_utts, float *out, int grid, hipStream_t st, float *lm, unsigned *gmax) {
  if (pl.lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mfcc_kernel<R, PRE, MLDS, KSR, false, TWO, BMEL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(pl.lds_bytes));
    if (e != hipSuccess) return e;
  }
  SAPR_LAUNCH((mfcc_kernel<R, PRE, MLDS, KSR, false, TWO, BMEL>), dim3(grid), dim3(kThreads), pl.lds_bytes, st, pcm, so,
              fo, n_utts, pl.dev, out, static_cast<unsigned long long *>(nullptr), lm, gmax);
  return hipGetLastError();
}

template <int R, bool TWO>
hipError_t launch(const MfccPlan &pl, const float *pcm, const int64_t *so, const int64_t *fo,
                  int64_t n_utts, float *out, int grid, hipStream_t st, float *lm, unsigned *gmax) {
  const bool pre = pl.dev.preemph != 0.f, ml = pl.dev.mel_in_lds != 0;
  if constexpr (R == 16) {
    if (pl.dev.ksr == kKsr) {
      if (pl.dev.mel_bf16) {
        if (pre) return launch_one<R, true, false, kKsr, TWO, true>(pl, pcm, so, fo, n_utts, out, grid, st, lm, gmax);
        return launch_one<R, false, false, kKsr, TWO, true>(pl, pcm, so, fo, n_utts, out, grid, st, lm, gmax);
      }
      if (pre) return launch_one<R, true, false, kKsr, TWO>(pl, pcm, so, fo, n_utts, out, grid, st, lm, gmax);
      return launch_one<R, false, false, kKsr, TWO>(pl, pcm, so, fo, n_utts, out, grid, st, lm, gmax);
    }
  }
  if (pre && ml) return launch_one<R, true, true, 0, TWO>(pl, pcm, so, fo, n_utts, out, grid, st, lm, gmax);
  if (pre) return launch_one<R, true, false, 0, TWO>(pl, pcm, so, fo, n_utts, out, grid, st, lm, gmax);
  if (ml) return launch_one<R, false, true, 0, TWO>(pl, pcm, so, fo, n_utts, out, grid, st, lm, gmax);
  return launch_one<R, false, false, 0, TWO>(pl, pcm, so, fo, n_utts, out, grid, st, lm, gmax);
}

// ---- wave-private core (mfcc_wave.h) ----
template <bool PRE, int RLO, int RHI, int S4>
hipError_t wave_prepare(size_t lds, int *blocks_per_cu) {
  const void *fn = reinterpret_cast<const void *>(&mfcc_wave_kernel<PRE, RLO, RHI, S4>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, kThreads, lds);
}
template <bool PRE, int RLO, int RHI, int S4>
hipError_t wave_launch_one(const MfccPlan &pl, const float *pcm, const int64_t *so, const int64_t *fo, int64_t n_utts,
                           int grid, int64_t span, hipStream_t st, float *lm, unsigned *gmax, int64_t total_cap) {
  SAPR_LAUNCH((mfcc_wave_kernel<PRE, RLO, RHI, S4>), dim3(grid), dim3(kThreads), pl.wave_lds, st, pcm, so, fo, n_utts,
              pl.dev, lm, gmax, span, total_cap);
  return hipGetLastError();
}
// dispatch over the instantiated (pre-emphasis, window rows, step quads) combinations; `prepare` != nullptr runs the
// occupancy query instead of a launch
template <bool PRE, int RLO, int RHI>
hipError_t wave_dispatch_s4(const MfccPlan &pl, int *prepare, const float *pcm, const int64_t *so, const int64_t *fo,
                            int64_t n_utts, int grid, int64_t span, hipStream_t st, float *lm, unsigned *gmax,
                            int64_t total_cap) {
  switch (pl.dev.wave_s4) {
#define SAPR_WAVE_CASE(S4)                                                                                        \
  case S4:                                                                                                        \
    return prepare ? wave_prepare<PRE, RLO, RHI, S4>(pl.wave_lds, prepare)                                        \
                   : wave_launch_one<PRE, RLO, RHI, S4>(pl, pcm, so, fo, n_utts, grid, span, st, lm, gmax, total_cap);
    SAPR_WAVE_CASE(6)
    SAPR_WAVE_CASE(7)
    SAPR_WAVE_CASE(8)
#undef SAPR_WAVE_CASE
    default:
      return hipErrorInvalidValue;
  }
}
hipError_t wave_dispatch(const MfccPlan &pl, int *prepare, const float *pcm, const int64_t *so, const int64_t *fo,
                         int64_t n_utts, int grid, int64_t span, hipStream_t st, float *lm, unsigned *gmax,
                         int64_t total_cap) {
  const bool pre = pl.dev.preemph != 0.f, tight = pl.wave_rlo == 1 && pl.wave_rhi == 15;
  if (pre && tight) return wave_dispatch_s4<true, 1, 15>(pl, prepare, pcm, so, fo, n_utts, grid, span, st, lm, gmax, total_cap);
  if (pre) return wave_dispatch_s4<true, 0, 16>(pl, prepare, pcm, so, fo, n_utts, grid, span, st, lm, gmax, total_cap);
  if (tight) return wave_dispatch_s4<false, 1, 15>(pl, prepare, pcm, so, fo, n_utts, grid, span, st, lm, gmax, total_cap);
  return wave_dispatch_s4<false, 0, 16>(pl, prepare, pcm, so, fo, n_utts, grid, span, st, lm, gmax, total_cap);
}

size_t finish_lds_bytes(const MfccDev &d) {
  return static_cast<size_t>(align_up(2 * 81, 4) + kFinRows * 16 + kFinRows * d.lm_stride) * 4;
}

}  // namespace
}  // namespace sapr

using namespace sapr;

extern "C" int sapr_mfcc_plan_create(double sr, int32_t n_fft, int32_t win_length, int32_t hop,
                                     int32_t n_mels, int32_t n_mfcc, double fmin, double fmax,
                                     double top_db, double preemph, int32_t deltas,
                                     int32_t max_frames, void **plan_out) {
  SAPR_REQUIRE(plan_out != nullptr, "plan_out is NULL");
  SAPR_REQUIRE(n_fft == 512 || n_fft == 2048, "n_fft must be 512 or 2048 (got %d)", n_fft);
  SAPR_REQUIRE(win_length > 0 && win_length <= n_fft && hop > 0, "bad window/hop");
  SAPR_REQUIRE(n_mels > 0 && n_mels <= 128 && n_mfcc > 0 && n_mfcc <= 16 && n_mfcc <= n_mels,
               "need 0 < n_mfcc <= 16, n_mfcc <= n_mels <= 128");
  SAPR_REQUIRE(max_frames >= 0, "max_frames must be >= 0 (0 = two-pass, any utterance length)");
  SAPR_REQUIRE(!deltas || max_frames == 0 || max_frames >= 9, "deltas need at least 9 frames");
  if (fmax <= 0) fmax = sr / 2;
  const int R = n_fft == 512 ? 16 : 32;
  const int Nc = R * R, nb = Nc + 1;
  MfccPlan *pl = new MfccPlan();
  pl->R = R;
  MfccDev &d = pl->dev;
  d.n_fft = n_fft;
  d.win_length = win_length;
  d.hop = hop;
  d.n_mels = n_mels;
  d.n_mfcc = n_mfcc;
  d.deltas = deltas ? 1 : 0;
  d.d_out = n_mfcc * (deltas ? 3 : 1);
  d.n_bins = nb;
  d.lm_stride = n_mels | 1;  // odd row stride of the LDS log-mel matrix [t][mel]
  d.t_pad = max_frames;
  d.two_pass = max_frames == 0 ? 1 : 0;
  d.preemph = static_cast<float>(preemph);
  d.top_db = static_cast<float>(top_db);
  d.amin = 1e-10f;

  // window: periodic Hamming (scipy get_window(fftbins=True)) centred in n_fft, times 0.5
  std::vector<float> win(n_fft, 0.f);
  const int lpad = (n_fft - win_length) / 2;
  for (int i = 0; i < win_length; ++i)
    win[lpad + i] = static_cast<float>(0.5 * (0.54 - 0.46 * std::cos(2.0 * kPi * i / win_length)));
  d.r_lo = lpad / (2 * R);
  d.r_hi = (lpad + win_length - 1) / (2 * R) + 1;
  if (d.r_hi > R) d.r_hi = R;

  std::vector<float> twab(2 * R * R), twu(2 * Nc);
  for (int k1 = 0; k1 < R; ++k1)
    for (int l = 0; l < R; ++l) {
      const double a = -2.0 * kPi * k1 * l / Nc;
      twab[2 * (k1 * R + l)] = static_cast<float>(std::cos(a));
      twab[2 * (k1 * R + l) + 1] = static_cast<float>(std::sin(a));
    }
  for (int k = 0; k < Nc; ++k) {
    const double a = -kPi * k / Nc;
    twu[2 * k] = static_cast<float>(std::cos(a));
    twu[2 * k + 1] = static_cast<float>(std::sin(a));
  }

  // banded MFMA A fragments of the mel filterbank.  Mels are cut into consecutive tiles of
  // <= 16 mels whose bands of non-zero bins need about the same number of K-steps (the Slaney
  // filters widen with frequency), one tile per wavefront and round, so the four wavefronts of
  // a workgroup finish the mel phase together and no cross-wavefront reduction is needed.
  std::vector<float> mel = mel_filterbank(sr, n_fft, n_mels, fmin, fmax);
  std::vector<int> mlo(n_mels), mhi(n_mels);
  for (int m = 0; m < n_mels; ++m) {
    int lo = nb, hi = -1;
    for (int b = 0; b < nb; ++b)
      if (mel[static_cast<size_t>(m) * nb + b] != 0.f) {
        lo = b < lo ? b : lo;
        hi = b > hi ? b : hi;
      }
    if (hi < 0) lo = hi = 0;
    mlo[m] = lo;
    mhi[m] = hi;
  }
  auto tile_ks = [&](int m0, int m1) {  // K-steps (multiple of 4) of mels [m0, m1)
    int lo = nb, hi = 0;
    for (int m = m0; m < m1; ++m) {
      lo = mlo[m] < lo ? mlo[m] : lo;
      hi = mhi[m] > hi ? mhi[m] : hi;
    }
    lo = lo / 4 * 4;
    return align_up((hi - lo) / 4 + 1, 4);
  };
  const int want_tiles = align_up((n_mels + 15) / 16, kWaves);
  auto cut = [&](int limit, std::vector<int> *starts) {
    int m0 = 0, n = 0;
    if (starts) starts->clear();
    while (m0 < n_mels) {
      int m1 = m0 + 1;
      if (tile_ks(m0, m1) > limit) return 1 << 30;
      while (m1 < n_mels && m1 - m0 < 16 && tile_ks(m0, m1 + 1) <= limit) ++m1;
      if (starts) starts->push_back(m0);
      m0 = m1;
      ++n;
    }
    return n;
  };
  int lim_lo = 4, lim_hi = align_up(nb / 4 + 1, 4);
  while (lim_lo < lim_hi) {
    const int mid = (lim_lo + lim_hi) / 8 * 4;
    if (cut(mid, nullptr) <= want_tiles)
      lim_hi = mid;
    else
      lim_lo = mid + 4;
  }
  std::vector<int> starts;
  cut(lim_lo, &starts);
  d.n_mtiles = static_cast<int>(starts.size());
  if (d.n_mtiles > 15) {
    delete pl;
    return fail(SAPR_ERR_UNSUPPORTED, "mel filterbank needs %d tiles (> 15)", d.n_mtiles);
  }
  std::vector<int> tiles(4 * (d.n_mtiles + 1), 0);
  std::vector<float> frag;
  int ks_total = 0;
  for (int mt = 0; mt < d.n_mtiles; ++mt) {
    const int m0 = starts[mt], m1 = mt + 1 < d.n_mtiles ? starts[mt + 1] : n_mels;
    int lo = nb;
    for (int m = m0; m < m1; ++m) lo = mlo[m] < lo ? mlo[m] : lo;
    lo = lo / 4 * 4;
    const int nks = tile_ks(m0, m1);
    tiles[4 * mt + 0] = m0;
    tiles[4 * mt + 1] = m1 - m0;
    tiles[4 * mt + 2] = lo;
    tiles[4 * mt + 3] = ks_total;
    for (int ks = 0; ks < nks; ++ks)
      for (int ln = 0; ln < 64; ++ln) {
        const int mi = ln & 15, b = lo + 4 * ks + (ln >> 4);
        frag.push_back((mi < m1 - m0 && b < nb) ? mel[static_cast<size_t>(m0 + mi) * nb + b] : 0.f);
      }
    ks_total += nks;
  }
  tiles[4 * d.n_mtiles + 3] = ks_total;  // sentinel: end of the last tile
  d.total_ks = ks_total;
  // the same bands as split-bf16 A fragments of v_mfma_f32_16x16x32_bf16 (register-fragment kernels only):
  // [tile][chunk c][hi, lo][reg i][lane]; lane (mel ln & 15, k-group ln >> 4) holds, in reg i, the bins
  // lo + 32c + 8i + 2(ln >> 4) + {0, 1} — the k order of the kernel's conflict-free B reads
  auto bf16_rn = [](float v) {
    unsigned b;
    std::memcpy(&b, &v, 4);
    b = (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;  // finite inputs only
    return b;
  };
  auto bf16_val = [](unsigned h) {
    const unsigned b = h << 16;
    float v;
    std::memcpy(&v, &b, 4);
    return v;
  };
  std::vector<unsigned> frag16;
  if (d.n_mtiles <= kWaves) {
    frag16.assign(static_cast<size_t>(d.n_mtiles) * 24 * 64, 0u);
    for (int mt = 0; mt < d.n_mtiles; ++mt) {
      const int m0 = tiles[4 * mt + 0], cnt = tiles[4 * mt + 1], lo = tiles[4 * mt + 2];
      for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 4; ++i)
          for (int ln = 0; ln < 64; ++ln) {
            unsigned hi_w = 0, lo_w = 0;
            for (int h = 0; h < 2; ++h) {
              const int mi = ln & 15, b = lo + 32 * c + 8 * i + 2 * (ln >> 4) + h;
              const float v = (mi < cnt && b < nb) ? mel[static_cast<size_t>(m0 + mi) * nb + b] : 0.f;
              const unsigned vh = bf16_rn(v), vl = bf16_rn(v - bf16_val(vh));
              hi_w |= vh << (16 * h);
              lo_w |= vl << (16 * h);
            }
            frag16[(static_cast<size_t>(mt) * 24 + 8 * c + i) * 64 + ln] = hi_w;
            frag16[(static_cast<size_t>(mt) * 24 + 8 * c + 4 + i) * 64 + ln] = lo_w;
          }
    }
  }
  // DCT-II ortho rows as A fragments: A[c][mel]
  const int nks_d = align_up(n_mels, 16) / 4;
  std::vector<float> dfrag(static_cast<size_t>(std::max(nks_d, 16)) * 64, 0.f);  // wave_finish walks 16 K-steps
  for (int ks = 0; ks < nks_d; ++ks)
    for (int ln = 0; ln < 64; ++ln) {
      const int c = ln & 15, m = 4 * ks + (ln >> 4);
      if (c < n_mfcc && m < n_mels) {
        const double sc = c == 0 ? std::sqrt(1.0 / (4.0 * n_mels)) : std::sqrt(1.0 / (2.0 * n_mels));
        dfrag[ks * 64 + ln] = static_cast<float>(2.0 * std::cos(kPi * c * (2 * m + 1) / (2.0 * n_mels)) * sc);
      }
    }
  // wave-private core: the banded filterbank as 16 MFMA blocks (n_fft 512 only; SAPR_MFCC_CORE=tile keeps the
  // workgroup-tile kernel)
  WavePack wp;
  {
    const char *core = std::getenv("SAPR_MFCC_CORE");
    const bool want = !(core && std::strcmp(core, "tile") == 0);
    if (want && R == 16 && hop % 2 == 0) wp = wave_pack(mel, n_mels, nb);
  }
  // delta tables
  std::vector<float> dtab(2 * 81, 0.f);
  for (int order = 1; order <= 2; ++order) {
    double row[9];
    savgol_row(order, 4.0, row);
    for (int k = 0; k < 9; ++k) dtab[(order - 1) * 81 + k] = static_cast<float>(row[k]);
    for (int p = 0; p < 4; ++p) {
      savgol_row(order, p, row);
      for (int k = 0; k < 9; ++k) dtab[(order - 1) * 81 + (1 + p) * 9 + k] = static_cast<float>(row[k]);
      savgol_row(order, 5 + p, row);
      for (int k = 0; k < 9; ++k) dtab[(order - 1) * 81 + (5 + p) * 9 + k] = static_cast<float>(row[k]);
    }
  }

  // one device buffer, 256-byte aligned sections
  auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t b_win = pad(win.size() * 4), b_ab = pad(twab.size() * 4), b_u = pad(twu.size() * 4),
               b_fr = pad(frag.size() * 4), b_ti = pad(tiles.size() * 4),
               b_df = pad(dfrag.size() * 4), b_dt = pad(dtab.size() * 4), b_f16 = pad(frag16.size() * 4 + 4),
               b_wa = pad(wp.a.size() * 4 + 4), b_wb = pad(wp.blk.size() * 4 + 4);
  const size_t total = b_win + b_ab + b_u + b_fr + b_ti + b_df + b_dt + b_f16 + b_wa + b_wb;
  std::vector<unsigned char> host(total, 0);
  size_t o = 0;
  auto put = [&](const void *src, size_t bytes, size_t padded) {
    std::copy(static_cast<const unsigned char *>(src), static_cast<const unsigned char *>(src) + bytes,
              host.begin() + o);
    const size_t at = o;
    o += padded;
    return at;
  };
  const size_t o_win = put(win.data(), win.size() * 4, b_win);
  const size_t o_ab = put(twab.data(), twab.size() * 4, b_ab);
  const size_t o_u = put(twu.data(), twu.size() * 4, b_u);
  const size_t o_fr = put(frag.data(), frag.size() * 4, b_fr);
  const size_t o_ti = put(tiles.data(), tiles.size() * 4, b_ti);
  const size_t o_df = put(dfrag.data(), dfrag.size() * 4, b_df);
  const size_t o_dt = put(dtab.data(), dtab.size() * 4, b_dt);
  const size_t o_f16 = put(frag16.data(), frag16.size() * 4, b_f16);
  const size_t o_wa = put(wp.a.data(), wp.a.size() * 4, b_wa);
  const size_t o_wb = put(wp.blk.data(), wp.blk.size() * 4, b_wb);
  unsigned char *devbuf = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&devbuf), total);
  if (e != hipSuccess) {
    delete pl;
    return hip_fail(e, "hipMalloc(mfcc tables)");
  }
  e = hipMemcpy(devbuf, host.data(), total, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(devbuf);
    delete pl;
    return hip_fail(e, "hipMemcpy(mfcc tables)");
  }
  pl->buffer = devbuf;
  d.window = reinterpret_cast<const float *>(devbuf + o_win);
  d.tw_ab = reinterpret_cast<const float2 *>(devbuf + o_ab);
  d.tw_u = reinterpret_cast<const float2 *>(devbuf + o_u);
  d.mel_frag = reinterpret_cast<const float *>(devbuf + o_fr);
  d.mel_tiles = reinterpret_cast<const int *>(devbuf + o_ti);
  d.dct_frag = reinterpret_cast<const float *>(devbuf + o_df);
  d.delta_tab = reinterpret_cast<const float *>(devbuf + o_dt);
  d.mel_frag_bf16 = reinterpret_cast<const unsigned *>(devbuf + o_f16);
  d.wave_s4 = wp.s4;
  d.wave_a = reinterpret_cast<const float *>(devbuf + o_wa);
  d.wave_blk = reinterpret_cast<const int *>(devbuf + o_wb);

  // staged PCM span of one tile of frames: from the first sample under the window of the tile's
  // first frame to the last sample under the window of its last frame
  {
    const int tile = R == 16 ? 16 : 8;
    d.span0 = 2 * R * d.r_lo - n_fft / 2;
    d.span_len = (tile - 1) * hop + 2 * R * (d.r_hi - d.r_lo);
    d.stage_floats = align_up(d.span_len, 4);
    if (d.span_len > kStagePasses * kThreads * 4 || hop % 2 != 0) {
      (void)hipFree(devbuf);
      delete pl;
      return fail(SAPR_ERR_UNSUPPORTED, "hop %d / window %d: staged span of %d samples unsupported (max %d, even hop)",
                  hop, win_length, d.span_len, kStagePasses * kThreads * 4);
    }
  }
  // filterbank fragments: in REGISTERS when the four wavefronts can own one mel tile each
  // (<= 4 tiles of <= kKsr K-steps: the 40-mel benchmark preset); else in LDS when that still
  // leaves room for two workgroups per CU (or at least fits); else streamed from L1/L2
  int max_nks = 0;
  for (int mt = 0; mt < d.n_mtiles; ++mt) {
    const int nks = tiles[4 * (mt + 1) + 3] - tiles[4 * mt + 3];
    max_nks = nks > max_nks ? nks : max_nks;
  }
  d.ksr = (R == 16 && d.n_mtiles <= kWaves && max_nks <= kKsr) ? kKsr : 0;
  // SAPR_MFCC_MEL=bf16 (read at plan creation) selects the split-bf16 filterbank product on the
  // register-fragment kernels.  Measured on MI355X (round 2): 9 bf16 MFMAs instead of 24 float32 ones per
  // wavefront and tile change the kernel time by < 0.1 % — the matrix pipe's time is hidden behind the other
  // wavefronts' VALU / LDS work — so the default stays the float32 product (no narrowing of the arithmetic).
  {
    const char *sel = std::getenv("SAPR_MFCC_MEL");
    d.mel_bf16 = (d.ksr && sel && std::strcmp(sel, "bf16") == 0) ? 1 : 0;
  }
  auto lds_total = [&](int ml) {
    const int tp = d.two_pass ? 0 : d.t_pad;
    return R == 16 ? lds_layout<16>(tp, d.lm_stride, d.total_ks, ml, d.n_mels, d.stage_floats).total
                   : lds_layout<32>(tp, d.lm_stride, d.total_ks, ml, d.n_mels, d.stage_floats).total;
  };
  auto pick_mel = [&]() {
    if (d.ksr) return 0;
    return lds_total(1) <= 80 * 1024 || (lds_total(0) > 80 * 1024 && lds_total(1) <= 160 * 1024) ? 1 : 0;
  };
  d.mel_in_lds = pick_mel();
  if (!d.two_pass && lds_total(d.mel_in_lds) > 160 * 1024) {
    // the utterance's log-mel matrix does not fit in LDS next to everything else: two-pass mode
    d.two_pass = 1;
    d.mel_in_lds = pick_mel();
  }
  if (!d.two_pass && lds_total(d.mel_in_lds) > 80 * 1024) {
    // (round 4) a fused layout above half of the CU's LDS runs ONE workgroup per CU — one wavefront per SIMD, 43 % of
    // its cycles in s_waitcnt at the reference preset.  If the layout without the utterance's log-mel matrix fits
    // twice, the log-mel round trip through HBM (512 B per frame at 128 mels) is the cheaper price: 10 000 x 1 s at
    // the reference preset 5.4 -> 3.7 ms.  SAPR_MFCC_FUSED=1 keeps the fused layout.
    const char *keep = std::getenv("SAPR_MFCC_FUSED");
    MfccDev t = d;
    t.two_pass = 1;
    auto total_of = [&](const MfccDev &x, int ml) {
      return R == 16 ? lds_layout<16>(0, x.lm_stride, x.total_ks, ml, x.n_mels, x.stage_floats).total
                     : lds_layout<32>(0, x.lm_stride, x.total_ks, ml, x.n_mels, x.stage_floats).total;
    };
    const int ml2 = t.ksr ? 0 : ((total_of(t, 1) <= 80 * 1024 || (total_of(t, 0) > 80 * 1024 && total_of(t, 1) <= 160 * 1024)) ? 1 : 0);
    if (!(keep && keep[0] == '1') && total_of(t, ml2) <= 80 * 1024) {
      d.two_pass = 1;
      d.mel_in_lds = ml2;
    }
  }
  pl->lds_bytes = static_cast<size_t>(lds_total(d.mel_in_lds));
  if (pl->lds_bytes > 160 * 1024) {
    (void)hipFree(devbuf);
    const size_t need = pl->lds_bytes;
    delete pl;
    return fail(SAPR_ERR_UNSUPPORTED, "configuration needs %zu bytes of LDS (> 160 KiB)", need);
  }
  if (d.wave_s4) {
    // wave-private core: always through the log-mel workspace (any utterance length)
    d.two_pass = 1;
    pl->wave_rlo = (d.r_lo >= 1 && d.r_hi <= 15) ? 1 : 0;
    pl->wave_rhi = (d.r_lo >= 1 && d.r_hi <= 15) ? 15 : 16;
    pl->wave_lds = static_cast<size_t>(wave_lds(d.wave_s4, wave_region_floats(d.n_mels, d.deltas)).total);
    pl->lds_bytes = pl->wave_lds;
    hipError_t oe = wave_dispatch(*pl, &pl->wave_blocks_per_cu, nullptr, nullptr, nullptr, 0, 0, 1, nullptr, nullptr,
                                  nullptr, 0);
    if (oe != hipSuccess || pl->wave_blocks_per_cu < 1) {
      (void)hipFree(devbuf);
      delete pl;
      return hip_fail(oe, "occupancy query (mfcc_wave_kernel)");
    }
  }
  *plan_out = pl;
  return 0;
}

extern "C" int sapr_mfcc_plan_destroy(void *plan) {
  if (!plan) return 0;
  MfccPlan *pl = static_cast<MfccPlan *>(plan);
  if (pl->buffer) (void)hipFree(pl->buffer);
  delete pl;
  return 0;
}

extern "C" int sapr_mfcc_plan_info(const void *plan, int32_t *d_out, int32_t *max_frames,
                                   int64_t *lds_bytes, int32_t *mel_ksteps) {
  SAPR_REQUIRE(plan != nullptr, "plan is NULL");
  const MfccPlan *pl = static_cast<const MfccPlan *>(plan);
  if (d_out) *d_out = pl->dev.d_out;
  if (max_frames) *max_frames = pl->dev.two_pass ? 0 : pl->dev.t_pad;  // 0: two-pass, unlimited
  if (lds_bytes) *lds_bytes = static_cast<int64_t>(pl->lds_bytes);
  if (mel_ksteps) *mel_ksteps = pl->dev.total_ks;
  return 0;
}

// Diagnostic: same launch as sapr_mfcc_batch for the (n_fft 512, no pre-emphasis, LDS filterbank)
// configuration with per-phase s_memtime sums; stamps[grid][4 wavefronts][12 phases].  The stamped
// build serialises memory waits at phase edges: read its SHARES, never its run time.
extern "C" int sapr_mfcc_batch_stamped(const void *plan, const float *pcm, const int64_t *sample_offsets,
                                       const int64_t *frame_offsets, int64_t n_utts, float *out,
                                       int32_t grid_blocks, uint64_t *stamps, void *stream) {
  SAPR_REQUIRE(plan && pcm && sample_offsets && frame_offsets && out && stamps, "NULL pointer argument");
  const MfccPlan *pl = static_cast<const MfccPlan *>(plan);
  SAPR_REQUIRE(pl->R == 16 && pl->dev.preemph == 0.f && pl->dev.ksr == kKsr, "stamped build: bench preset only");
  SAPR_REQUIRE(grid_blocks > 0 && grid_blocks <= n_utts, "bad grid");
  if (pl->dev.mel_bf16) {
    if (pl->lds_bytes > 64 * 1024)
      SAPR_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&mfcc_kernel<16, false, false, kKsr, true, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(pl->lds_bytes)));
    SAPR_LAUNCH((mfcc_kernel<16, false, false, kKsr, true, false, true>), dim3(grid_blocks), dim3(kThreads),
                pl->lds_bytes, as_stream(stream), pcm, sample_offsets, frame_offsets, n_utts, pl->dev, out,
                reinterpret_cast<unsigned long long *>(stamps));
  } else {
    if (pl->lds_bytes > 64 * 1024)
      SAPR_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&mfcc_kernel<16, false, false, kKsr, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(pl->lds_bytes)));
    SAPR_LAUNCH((mfcc_kernel<16, false, false, kKsr, true>), dim3(grid_blocks), dim3(kThreads), pl->lds_bytes,
                as_stream(stream), pcm, sample_offsets, frame_offsets, n_utts, pl->dev, out,
                reinterpret_cast<unsigned long long *>(stamps));
  }
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int sapr_mfcc_workspace_bytes(const void *plan, int64_t total_frames, int64_t n_utts, size_t *bytes) {
  SAPR_REQUIRE(plan && bytes && total_frames >= 0 && n_utts >= 0, "bad arguments");
  const MfccPlan *pl = static_cast<const MfccPlan *>(plan);
  *bytes = pl->dev.two_pass
               ? (static_cast<size_t>(total_frames) * pl->dev.n_mels + static_cast<size_t>(n_utts)) * sizeof(float) + 256
               : 0;
  return 0;
}

extern "C" int sapr_mfcc_batch(const void *plan, const float *pcm, const int64_t *sample_offsets,
                               const int64_t *frame_offsets, int64_t n_utts, int64_t total_frames, float *out,
                               int32_t grid_blocks, void *workspace, size_t workspace_size, void *stream) {
  SAPR_REQUIRE(plan != nullptr, "plan is NULL");
  SAPR_REQUIRE(n_utts >= 0 && total_frames >= 0, "bad sizes");
  if (n_utts == 0) return 0;
  SAPR_REQUIRE(pcm && sample_offsets && frame_offsets && out, "NULL pointer argument");
  const MfccPlan *pl = static_cast<const MfccPlan *>(plan);
  int dev = 0, cus = 256;
  SAPR_HIP_TRY(hipGetDevice(&dev));
  SAPR_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int grid = grid_blocks;
  if (grid <= 0) {
    const int per_cu = static_cast<int>((160 * 1024) / pl->lds_bytes);
    const int cap = 3;
    grid = cus * (per_cu < 1 ? 1 : (per_cu > cap ? cap : per_cu));
  }
  if (grid > n_utts) grid = static_cast<int>(n_utts);
  hipStream_t st = as_stream(stream);
  if (!pl->dev.two_pass) {
    if (pl->R == 16)
      SAPR_HIP_TRY((launch<16, false>(*pl, pcm, sample_offsets, frame_offsets, n_utts, out, grid, st, nullptr, nullptr)));
    else
      SAPR_HIP_TRY((launch<32, false>(*pl, pcm, sample_offsets, frame_offsets, n_utts, out, grid, st, nullptr, nullptr)));
    return 0;
  }
  const size_t need = (static_cast<size_t>(total_frames) * pl->dev.n_mels + static_cast<size_t>(n_utts)) * sizeof(float) + 256;
  if (!workspace || workspace_size < need)
    return fail(SAPR_ERR_WORKSPACE, "two-pass MFCC plan needs a %zu-byte workspace (sapr_mfcc_workspace_bytes)", need);
  float *lm = static_cast<float *>(workspace);
  unsigned *gmax = reinterpret_cast<unsigned *>(lm + static_cast<size_t>(total_frames) * pl->dev.n_mels);
  if (pl->dev.wave_s4) {
    // persistent wavefronts; each takes an equal run of `span` consecutive frames of the batch (whole 4-frame sets,
    // across utterance boundaries), so the grid is balanced to +- one set for any batch size and length mix
    int wgrid = grid_blocks > 0 ? grid_blocks : cus * pl->wave_blocks_per_cu;
    const int64_t n_waves = static_cast<int64_t>(wgrid) * kWaves;
    int64_t span = (total_frames + n_waves - 1) / n_waves;
    span = span < 4 ? 4 : (span + 3) / 4 * 4;
    const int64_t need_blocks = ((total_frames + span - 1) / span + kWaves - 1) / kWaves;
    if (need_blocks < wgrid) wgrid = static_cast<int>(need_blocks < 1 ? 1 : need_blocks);
    SAPR_HIP_TRY(hipMemsetAsync(gmax, 0, static_cast<size_t>(n_utts) * sizeof(unsigned), st));
    SAPR_HIP_TRY(wave_dispatch(*pl, nullptr, pcm, sample_offsets, frame_offsets, n_utts, wgrid, span, st, lm, gmax,
                               total_frames));
    // second half: a wavefront per utterance again (16 resident wavefronts per CU keep ~48 log-mel tiles in flight)
    int fgrid = cus * SAPR_FINISH_OCC;
    const int64_t fwaves = static_cast<int64_t>(fgrid) * kWaves;
    int fsplit = 1;
    if (n_utts < fwaves) {
      fsplit = static_cast<int>(std::min<int64_t>(8, fwaves / n_utts));
      const int64_t need_blocks = (n_utts * fsplit + kWaves - 1) / kWaves;
      if (need_blocks < fgrid) fgrid = static_cast<int>(need_blocks);
    }
    SAPR_LAUNCH(mfcc_wave_finish_kernel, dim3(fgrid), dim3(kThreads), wave_finish_lds(pl->dev.n_mels, pl->dev.deltas), st,
                lm, gmax, frame_offsets, n_utts, pl->dev, out, fsplit, total_frames);
    SAPR_HIP_TRY(hipGetLastError());
    return 0;
  } else if (pl->R == 16)
    SAPR_HIP_TRY((launch<16, true>(*pl, pcm, sample_offsets, frame_offsets, n_utts, out, grid, st, lm, gmax)));
  else
    SAPR_HIP_TRY((launch<32, true>(*pl, pcm, sample_offsets, frame_offsets, n_utts, out, grid, st, lm, gmax)));
  int fgrid = cus * 6;
  if (fgrid > n_utts) fgrid = static_cast<int>(n_utts);
  SAPR_LAUNCH(mfcc_finish_kernel, dim3(fgrid), dim3(kThreads), finish_lds_bytes(pl->dev), st, lm, gmax, frame_offsets,
              n_utts, pl->dev, out);
  SAPR_HIP_TRY(hipGetLastError());
  return 0;
}
