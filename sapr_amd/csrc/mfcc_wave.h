// Wave-private MFCC core for n_fft = 512 (included by mfcc.hip inside namespace sapr::<anonymous>).
//
// Replaces the spectral half of librosa.feature.mfcc as the reference calls it (mfcc_extract.py:15-23):
// centre padding -> Hamming window -> rFFT -> |X|^2 -> Slaney mel filterbank -> 10 log10; the utterance-global
// top_db clip, the DCT-II and the deltas run in mfcc_finish_kernel over the log-mel workspace.
//
// One WAVEFRONT owns four consecutive frames of one utterance from their PCM samples to their log-mel rows and
// shares nothing with the other wavefronts of its workgroup but read-only tables: after the table load there is no
// workgroup barrier, so co-resident wavefronts drift apart and cover each other's LDS round trips.
//
//   samples  raw buffer loads straight into the FFT's register layout (lane l of a frame's 16-lane group holds the
//            complex points z[l + 16 r] = y[2n] + i y[2n+1]); a descriptor of the utterance's own samples makes
//            every offset outside the signal read as 0 = librosa's centre padding.  They are issued one set
//            ahead, into the registers the FFT has just vacated, under the filterbank phase.
//   FFT      256-point complex = 16 lanes x 16 register points: in-lane radix-2 DIF, twiddle, 16 x 16 transpose
//            through the wavefront's own LDS scratch (real and imaginary plane in turn; rows of 18 floats make
//            the ds_write_b32 columns and the ds_read_b64 rows conflict-free), in-lane DIF again.
//   untangle lanes are laid out so that the conjugate partner sits in the mirrored lane of the DPP row: ONE
//            row_mirror move per value (the two self-paired residues 0 and 8 occupy lanes 0 and 15).
//   mel      the four power rows go to the wavefront's scratch and come back as B operands of
//            v_mfma_f32_4x4x1_16b_f32: 16 independent 4 mel x 4 frame blocks.  The triangular filters are
//            banded, so block b walks only the bins of ITS four mels (wide high-frequency bands are cut into up
//            to four parts on neighbouring blocks, summed by two DPP row shifts): 4*S4 steps instead of 257, no
//            16-frame tile, no cross-wavefront exchange.
//   log      on the accumulators, one 16-byte store per lane into log_mel[frame][mel]; running maximum per
//            utterance for the top_db clip.
//   split    equal runs of consecutive frames per wavefront, across utterance boundaries (see the kernel).
#pragma once

constexpr int kWRowPad = 18;              // floats per transpose row
constexpr int kWGroup = 16 * kWRowPad;    // transpose scratch of one frame group
constexpr int kWRegion = 4 * kWGroup;     // floats of LDS scratch per wavefront (1152)
#include "mfcc_wave_pack.h"  // kWPRow, kWMinS4, kWMaxS4, wave_pack()

struct WaveLds {
  int win, twab, twu, mela, dtab, scr, total;
};
__host__ __device__ inline int wave_region_floats(int, int) { return kWRegion; }
__host__ __device__ inline WaveLds wave_lds(int s4, int region_floats) {
  WaveLds L;
  int o = 0;
  L.win = o;
  o += 512 * 4;
  L.twab = o;
  o += 256 * 8;
  L.twu = o;
  o += 128 * 8;
  L.mela = o;
  o += s4 * kWave * 16;
  L.dtab = o;
  o += 2 * 81 * 4 + 8;
  L.scr = o;
  o += kWaves * region_floats * 4;
  L.total = o;
  return L;
}

// float -> unsigned whose order is the float order (atomicMax across the wavefronts that share an utterance)
__host__ __device__ inline unsigned enc_ordered(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float dec_ordered(unsigned e) {
  const unsigned u = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
  return __builtin_bit_cast(float, u);
}

struct __attribute__((packed, aligned(4))) f2u {
  float x, y;
};

template <bool PREEMPH, int NR>
struct WaveSet {
  f2u y[NR];
  float m0;  // pre-emphasis: the sample before the first row's pair (later rows take theirs from the neighbouring lane)
};

// Explicit 8-byte LDS reads.  Left to itself the compiler fuses two neighbouring 8-byte reads into one ds_read2_b64,
// which the LDS serves at ~120 B/clk/CU against ~200 for two ds_read_b64 (scripts/ubench/lds_rate.hip) — and the LDS
// pipe is this kernel's busiest unit.  The reads are therefore issued by hand, in batches, and waited for by hand:
// the compiler does not count them in lgkmcnt, which is harmless (its own waits can only become longer, LDS returns in
// order) as long as every value is consumed behind lds_wait() + lds_dep().
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void *)p));
}
template <int OFF>
__device__ __forceinline__ v2f ds_rd64(unsigned addr) {  // read-only tables
  v2f v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ v2f ds_rd64_mem(unsigned addr) {  // data the wavefront has just stored
  v2f v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void lds_dep(v2f (&v)[N]) {  // ties the values to the preceding lds_wait()
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

#ifndef SAPR_WAVE_CNDDPP
#define SAPR_WAVE_CNDDPP 1
#endif
#ifndef SAPR_WAVE_FRAG64
#define SAPR_WAVE_FRAG64 0
#endif
// (a, b) of the row-mirrored lane, except in the lanes of `keep`, which get their own: one v_cndmask_b32 with a DPP
// source each.  VOP2 DPP takes the select from vcc; the s_mov + s_nop are also the two wait states a DPP read needs
// behind a vector write of its source (the compiler's hazard recogniser does not look into inline assembly).
__device__ __forceinline__ void mirror_unless2(float a, float b, unsigned long long keep, float &ma, float &mb) {
  asm("s_mov_b64 vcc, %4\n\ts_nop 0\n\t"
      "v_cndmask_b32_dpp %0, %2, %2, vcc row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %1, %3, %3, vcc row_mirror row_mask:0xf bank_mask:0xf"
      : "=&v"(ma), "=v"(mb)
      : "v"(a), "v"(b), "s"(keep)
      : "vcc");
}

// The same as a v_mov_b32 with a DPP source followed by a VOP3 v_cndmask_b32 whose select is an SGPR pair
// (SAPR_WAVE_CNDDPP=2).  scripts/ubench/mix_rate: a v_cndmask_b32 that takes its select from vcc issues at 0.08 per cycle
// and SIMD on gfx950 (12 cycles; the DPP form has no other), the VOP3 form with an SGPR pair at 0.47, v_mov_b32_dpp at 0.32.
__device__ __forceinline__ void mirror_unless2_e64(float a, float b, unsigned long long keep, float &ma, float &mb) {
  asm("s_nop 1\n\t"
      "v_mov_b32_dpp %0, %2 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %1, %3 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_e64 %0, %0, %2, %4\n\t"
      "v_cndmask_b32_e64 %1, %1, %3, %4"
      : "=&v"(ma), "=&v"(mb)
      : "v"(a), "v"(b), "s"(keep));
}

// Second half of the chain for ONE utterance, by ONE wavefront, 16 frames at a time: top_db clip of the log-mel rows
// (mfcc_extract.py:15-23 -> librosa power_to_db), orthonormal DCT-II on v_mfma_f32_16x16x4_f32 (A fragments from L1/L2,
// as in mfcc_finish_kernel), Savitzky-Golay delta / delta-delta (scipy mode="interp" edge polynomials), frame-major
// store.  s_tile: 16 x (n_mels + 4) floats; s_ceps: one 16 x 16 cepstra tile, or a ring of three with deltas (tile e
// is emitted once tile e + 1 is there).  Wave-private LDS, wavefront-level ordering only.
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void wave_finish(const float *__restrict__ lm, int T, float floor_db, float *__restrict__ out,
                                            const MfccDev &P, float *s_tile, float *s_ceps, const float *s_dtab,
                                            int lane, int tile_lo, int tile_hi) {
  const int q = lane >> 4, j16 = lane & 15;
  const int NQ = P.n_mels >> 2, LS = P.n_mels + 4, per_tile = 16 * NQ;  // NQ <= 16: at most 4 float4 per lane and tile
  const float inv_nm = 1.0f / static_cast<float>(P.n_mfcc);
  const int n_tiles = (T + 15) >> 4;
  const bool deltas = P.deltas != 0;
  float dc1[9], dc2[9];  // the centred Savitzky-Golay filters (row 0 of either table)
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    dc1[k] = deltas ? s_dtab[k] : 0.f;
    dc2[k] = deltas ? s_dtab[81 + k] : 0.f;
  }
  // this wavefront emits tiles [tile_lo, tile_hi); with deltas it also needs the cepstra of one tile either side
  const int first = deltas && tile_lo > 0 ? tile_lo - 1 : tile_lo;
  const int last = deltas && tile_hi < n_tiles ? tile_hi + 1 : tile_hi;  // one past the last tile transformed
  // DCT rows as MFMA A fragments, 16 K-steps (zero beyond n_mels)
  float afr[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) afr[ks] = P.dct_frag[ks * kWave + lane];
  // (row, column quad) of this lane's float4 pieces of a tile
  int prow_[4], pcg_[4];
  {
    const float inv_nq = 1.0f / static_cast<float>(NQ);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int idx = lane + kWave * c;
      prow_[c] = static_cast<int>((static_cast<float>(idx) + 0.5f) * inv_nq);
      pcg_[c] = idx - prow_[c] * NQ;
    }
  }
  auto fetch = [&](int nt, float4 (&v)[4]) {  // rows past the utterance repeat its last row: never stored
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (lane + kWave * c < per_tile) {
        const int t = 16 * nt + prow_[c] < T ? 16 * nt + prow_[c] : T - 1;
        v[c] = reinterpret_cast<const float4 *>(lm + static_cast<int64_t>(t) * P.n_mels)[pcg_[c]];
      }
  };
  // The rows were written ~26 sets ago and have left the L2 (the PCM stream runs through it): each fetch is an
  // Infinity Cache / HBM round trip, so three tiles are kept in flight.
  auto tile = [&](int nt, float4 (&cur)[4]) {
    if (nt < last) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (lane + kWave * c < per_tile) {
          float4 v = cur[c];
          v.x = fmaxf(v.x, floor_db);
          v.y = fmaxf(v.y, floor_db);
          v.z = fmaxf(v.z, floor_db);
          v.w = fmaxf(v.w, floor_db);
          *reinterpret_cast<float4 *>(s_tile + prow_[c] * LS + 4 * pcg_[c]) = v;
        }
      if (nt + 3 < last) fetch(nt + 3, cur);
      wave_fence();
      // K-steps past n_mels read the next rows' values times a zero fragment
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float *lrow = s_tile + j16 * LS + q;
      float bv[16];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) bv[ks] = lrow[4 * ks];
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[ks], bv[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[ks + 1], bv[ks + 1], acc1, 0, 0, 0);
      }
      *reinterpret_cast<f32x4 *>(s_ceps + (deltas ? (nt % 3) * 256 : 0) + j16 * 16 + 4 * q) = acc0 + acc1;
      wave_fence();
    }
    const int e = deltas ? nt - 1 : nt;
    if (e >= tile_lo && e < tile_hi) {
      const int rows = T - 16 * e < 16 ? T - 16 * e : 16;
      float *__restrict__ o = out + static_cast<int64_t>(16 * e) * P.d_out;
      if (!deltas) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {  // 16 x n_mfcc <= 256 contiguous floats: d_out == n_mfcc
          const int idx = lane + kWave * c4;
          if (idx < rows * P.n_mfcc) {
            const int tl = static_cast<int>((static_cast<float>(idx) + 0.5f) * inv_nm), c = idx - tl * P.n_mfcc;
            o[idx] = s_ceps[tl * 16 + c];
          }
        }
      } else {
        auto ceps = [&](int t, int c) { return s_ceps[((t >> 4) % 3) * 256 + (t & 15) * 16 + c]; };
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const int idx = lane + kWave * c4;
          if (idx < rows * P.n_mfcc) {
            const int tl = static_cast<int>((static_cast<float>(idx) + 0.5f) * inv_nm), c = idx - tl * P.n_mfcc;
            const int t = 16 * e + tl;
            float d1 = 0.f, d2 = 0.f;
            if (t >= 4 && t < T - 4) {  // interior frames: the centred filters, kept in registers (no table reads)
#pragma unroll
              for (int k = 0; k < 9; ++k) {
                const float x = ceps(t - 4 + k, c);
                d1 += dc1[k] * x;
                d2 += dc2[k] * x;
              }
            } else {  // the four frames at either end: scipy's mode="interp" edge polynomials
              const int row = t < 4 ? 1 + t : 5 + (t - (T - 4)), t0 = t < 4 ? 0 : T - 9;
#pragma unroll
              for (int k = 0; k < 9; ++k) {
                const float x = ceps(t0 + k, c);
                d1 += s_dtab[row * 9 + k] * x;
                d2 += s_dtab[81 + row * 9 + k] * x;
              }
            }
            float *ot = o + tl * P.d_out + c;
            ot[0] = ceps(t, c);
            ot[P.n_mfcc] = d1;
            ot[2 * P.n_mfcc] = d2;
          }
        }
      }
      wave_fence();
    }
  };
  float4 t0[4], t1[4], t2[4];
  if (first < last) fetch(first, t0);
  if (first + 1 < last) fetch(first + 1, t1);
  if (first + 2 < last) fetch(first + 2, t2);
  const int end = last + (deltas ? 1 : 0);  // the last emission trails the last transform by one tile
  for (int base = first; base < end; base += 3) {
    tile(base, t0);
    if (base + 1 < end) tile(base + 1, t1);
    if (base + 2 < end) tile(base + 2, t2);
  }
}

#ifndef SAPR_FINISH_OCC
#define SAPR_FINISH_OCC 3  // wavefronts per SIMD of the finish pass (workgroups per CU of its grid)
#endif
// The finish pass as its own launch: a wavefront per utterance (long utterances: `split` wavefronts take contiguous
// runs of its tiles), wave-private LDS, no workgroup barrier after the table load.  gmax_enc holds the utterance
// maxima the spectral kernel found.
__global__ __launch_bounds__(kThreads, SAPR_FINISH_OCC) void mfcc_wave_finish_kernel(const float *__restrict__ lm,
                                                                       const unsigned *__restrict__ gmax_enc,
                                                                       const int64_t *__restrict__ frame_offsets,
                                                                       int64_t n_utts, MfccDev P, float *__restrict__ out,
                                                                       int split, int64_t total_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *s_dtab = reinterpret_cast<float *>(smem);
  const int per_wave = 16 * (P.n_mels + 4) + (P.deltas ? 3 : 1) * 256;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
  for (int i = tid; i < 2 * 81; i += kThreads) s_dtab[i] = P.delta_tab[i];
  // DCT K-steps past n_mels read whatever follows a row (times a zero fragment): it has to be finite
  for (int i = tid; i < kWaves * per_wave; i += kThreads) s_dtab[168 + i] = 0.f;
  __syncthreads();
  float *s_tile = s_dtab + 168 + wave * per_wave;
  const int n_waves = gridDim.x * kWaves;
  const int wid = blockIdx.x * kWaves + wave;
  if (frame_offsets[n_utts] > total_cap) {
    // the offsets on the device describe MORE frames than the caller sized `out` and the workspace for (sapr_hip.h:
    // frame_offsets[n_utts] == total_frames is a hard precondition): nothing was computed (mfcc_wave_kernel
    // returned) — the whole output becomes NaN so that the mismatch cannot pass for features
    const int64_t n = total_cap * P.d_out;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + tid; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads)
      out[i] = __builtin_nanf("");
    return;
  }
  const int n_teams = n_waves / split, part = wid % split;
  if (wid >= n_teams * split) return;
  for (int64_t u = wid / split; u < n_utts; u += n_teams) {
    const int64_t f_beg = frame_offsets[u];
    const int T = static_cast<int>(frame_offsets[u + 1] - f_beg);
    const int n_tiles = (T + 15) >> 4, per = (n_tiles + split - 1) / split;
    const int lo = part * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
    if (lo >= hi) continue;
    wave_finish(lm + f_beg * P.n_mels, T, dec_ordered(gmax_enc[u]) - P.top_db, out + f_beg * P.d_out, P, s_tile,
                s_tile + 16 * (P.n_mels + 4), s_dtab, lane, lo, hi);
  }
}
inline size_t wave_finish_lds(int n_mels, int deltas) {
  return static_cast<size_t>(168 + kWaves * (16 * (n_mels + 4) + (deltas ? 3 : 1) * 256)) * 4;
}

#ifndef SAPR_WAVE_OCC
#define SAPR_WAVE_OCC 4  // wavefronts per SIMD the register allocation aims at
#endif

template <bool PREEMPH, int RLO, int RHI, int S4>
__global__ __launch_bounds__(kThreads, SAPR_WAVE_OCC) void mfcc_wave_kernel(
    const float *__restrict__ pcm, const int64_t *__restrict__ sample_offsets,
    const int64_t *__restrict__ frame_offsets, int64_t n_utts, MfccDev P, float *__restrict__ lm_out,
    unsigned *__restrict__ gmax_enc, int64_t span, int64_t total_cap) {
  constexpr int R = 16, kNc = 256, kBits = 4, NR = RHI - RLO;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int region_floats = wave_region_floats(P.n_mels, P.deltas);
  const WaveLds L = wave_lds(S4, region_floats);
  float *s_win = reinterpret_cast<float *>(smem + L.win);
  float2 *s_twab = reinterpret_cast<float2 *>(smem + L.twab);
  float2 *s_twu = reinterpret_cast<float2 *>(smem + L.twu);
  float4 *s_a = reinterpret_cast<float4 *>(smem + L.mela);
  float *s_dtab = reinterpret_cast<float *>(smem + L.dtab);
  float *s_scr = reinterpret_cast<float *>(smem + L.scr);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int lane = tid % kWave;

  // ---- read-only tables -> LDS, scratch zeroed (filterbank K padding reads it times a zero weight) ----
  for (int i = tid; i < 512; i += kThreads) s_win[i] = P.window[i];
  for (int i = tid; i < 256; i += kThreads) s_twab[i] = P.tw_ab[i];
  for (int i = tid; i < 128; i += kThreads) s_twu[i] = P.tw_u[i];
  for (int i = tid; i < S4 * kWave; i += kThreads) s_a[i] = reinterpret_cast<const float4 *>(P.wave_a)[i];
  for (int i = tid; i < 2 * 81; i += kThreads) s_dtab[i] = P.delta_tab[i];
  for (int i = tid; i < kWaves * region_floats; i += kThreads) s_scr[i] = 0.f;
  __syncthreads();  // the only workgroup barrier

  const int grp = lane >> 4;  // frame of the set (FFT layout)
  const int l = lane & 15;    // lane inside the frame's DPP row
  // pass B: lane l plays residue sigma (k = sigma + 16 k2); the conjugate partner 16 - sigma sits in lane 15 - l,
  // the self-paired residues 0 and 8 in lanes 0 and 15
  const int sigma = l < 8 ? l : (l == 15 ? 8 : l + 1);
  const bool is0 = l == 0;
  [[maybe_unused]] const bool special = l == 0 || l == 15;  // (only the build without SAPR_WAVE_CNDDPP reads it)
  const unsigned long long special_mask = 0x8001800180018001ull;  // lanes 0 and 15 of every DPP row
  float *region = s_scr + wave * region_floats;
  float *scr = region + grp * kWGroup;
  float *prow = region + grp * kWPRow;
  // Transpose scratch of a frame group: 16 rows of 18 floats.  The ds_read_b64 rows of the two groups one LDS pass
  // serves (32 lanes) are conflict-free when the groups lie 32 banks apart — which puts their ds_write_b32 columns on
  // the same banks.  Odd groups therefore keep row k1 at position (k1 + 8) % 16: +8 rows = +144 floats = +16 banks.
  const int rot = (grp & 1) * 8;
  float *wlo = scr + rot * kWRowPad + l;                  // rows k1 < 8
  float *whi = scr + (rot ? -8 : 0) * kWRowPad + l;       // rows k1 >= 8 (plus k1 rows)
  const unsigned a_rrow = lds_addr(scr + ((sigma + rot) & 15) * kWRowPad);
  const unsigned a_twu = lds_addr(s_twu + sigma);
  const unsigned a_twab = lds_addr(s_twab + l);
  const unsigned a_win = lds_addr(reinterpret_cast<const float2 *>(s_win) + l);

  // filterbank: block b = lane / 4 accumulates four mels x the set's four frames (column j = lane % 4)
  const int j4 = lane & 3;
  const int4 bi = reinterpret_cast<const int4 *>(P.wave_blk)[lane >> 2];
  const float4 *bq = reinterpret_cast<const float4 *>(region + j4 * kWPRow + bi.x);
  const float4 *aq = s_a + lane;
  const int mel0 = bi.y;  // first mel of the block's group if it is the group's head part, else -1
  const float f1 = bi.z ? 1.f : 0.f, f2 = bi.w ? 1.f : 0.f;

  // Work split: the batch's frames, numbered through all utterances, are cut into equal runs of `span` frames, one
  // run per wavefront; a wavefront owns every 4-frame set whose FIRST frame lies in its run.  Runs ignore utterance
  // boundaries (a wavefront finishes the tail of one utterance, takes whole ones, starts the head of another), so
  // every wavefront does the same number of sets +- 1 whatever the batch size and the utterance lengths; an
  // utterance shared by several wavefronts gets its maximum by atomicMax (gmax_enc is zeroed before the launch).
  const int64_t total = frame_offsets[n_utts];
  const int64_t wid = static_cast<int64_t>(blockIdx.x) * kWaves + wave;
  // the caller sized the log-mel workspace (and the maxima placed behind it) and the grid for `total_cap` frames: when
  // the offsets on the device describe more, nothing is written (mfcc_wave_finish_kernel then marks the output)
  if (total > total_cap) return;
  const int64_t run_lo = wid * span;
  if (run_lo >= total) return;
  const int64_t run_hi = run_lo + span < total ? run_lo + span : total;
  int64_t u_first = 0;
  {  // the last utterance that starts at or before frame run_lo (frame_offsets[0] = 0 <= run_lo < total = frame_offsets[n_utts])
    int64_t hi = n_utts;
    while (hi - u_first > 1) {
      const int64_t mid = (u_first + hi) >> 1;
      if (frame_offsets[mid] <= run_lo)
        u_first = mid;
      else
        hi = mid;
    }
  }

  for (int64_t u = u_first; u < n_utts; ++u) {
    const int64_t f_beg = frame_offsets[u];
    if (f_beg >= run_hi) break;
    const int T = static_cast<int>(frame_offsets[u + 1] - f_beg);
    const int n_sets = (T + 3) >> 2;
    const int s_lo = f_beg >= run_lo ? 0 : static_cast<int>((run_lo - f_beg + 3) >> 2);
    const int s_hi_raw = static_cast<int>((run_hi - f_beg + 3) >> 2);
    const int s_hi = s_hi_raw < n_sets ? s_hi_raw : n_sets;
    if (s_lo >= s_hi) continue;
    const int64_t s_beg = sample_offsets[u];
    const int n_samp = static_cast<int>(sample_offsets[u + 1] - s_beg);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(pcm + s_beg), 0, n_samp * 4, 0x00020000 /* raw dword buffer */);
    float run_max = -3.0e38f;

    // samples of set s -> registers.  Offsets before sample 0 are negative = huge unsigned = out of range = 0; the
    // hardware adds the instruction's immediate offset (128 r bytes, folded by the compiler) to the vector offset
    // modulo 2^32 before the range check (scripts/ubench/bufoff_probe.hip), so a negative base with an in-range
    // sum still reads the sample.
    WaveSet<PREEMPH, NR> nxt;
    auto issue = [&](int s) {
      const int vo = ((4 * s + grp) * P.hop - kNc + 2 * l) * 4;
      static_for<RLO, RHI>([&](auto r_c) {
        constexpr int r = decltype(r_c)::value;
        nxt.y[r - RLO] = __builtin_bit_cast(f2u, __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo + 8 * R * r, 0, 0));
      });
      if constexpr (PREEMPH)
        nxt.m0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo + 8 * R * RLO - 4, 0, 0));
    };
    issue(s_lo);

    for (int s = s_lo; s < s_hi; ++s) {
      // ============================ window, FFT pass A ============================
      float re[R], im[R];
      {
        v2f w[NR];
        static_for<RLO, RHI>([&](auto r_c) {
          constexpr int r = decltype(r_c)::value;
          w[r - RLO] = ds_rd64<8 * R * r>(a_win);
        });
        const int n0 = (4 * s + grp) * P.hop - kNc + 2 * l;
        // does the set's last sample lie inside the signal?  (pre-emphasis must not leak the last real sample
        // into the first padded one)
        const bool inside = (4 * s + 3) * P.hop + kNc <= n_samp;
        lds_wait();
        lds_dep(w);
        static_for<0, R>([&](auto r_c) {
          constexpr int r = decltype(r_c)::value;
          if constexpr (r < RLO || r >= RHI) {
            re[r] = 0.f;
            im[r] = 0.f;
          } else {
            f2u y = nxt.y[r - RLO];
            if constexpr (PREEMPH) {
              // y[2n - 1] is the second sample of lane l - 1's pair of this row (row_shr:1; lane 0 has no source and
              // keeps `old`) or, for lane 0, of lane 15's pair of the previous row (row_ror:1)
              float m = nxt.m0;
              if constexpr (r > RLO) {
                const int wrap = __builtin_amdgcn_mov_dpp(__float_as_int(nxt.y[r - 1 - RLO].y), 0x121, 0xf, 0xf, true);
                m = __int_as_float(__builtin_amdgcn_update_dpp(wrap, __float_as_int(y.y), 0x111, 0xf, 0xf, false));
              }
              const float ya = y.x - P.preemph * m;
              const float yb = y.y - P.preemph * y.x;
              y.x = ya;
              y.y = yb;
              if (!inside) {
                const int idx = n0 + 2 * R * r;
                y.x = idx < n_samp ? y.x : 0.f;
                y.y = idx + 1 < n_samp ? y.y : 0.f;
              }
            }
            re[r] = y.x * w[r - RLO].x;
            im[r] = y.y * w[r - RLO].y;
          }
        });
      }
      {
        v2f tw[R - 1];  // in flight under the in-lane FFT
        static_for<1, R>([&](auto k1_c) {
          constexpr int k1 = decltype(k1_c)::value;
          tw[k1 - 1] = ds_rd64<8 * R * k1>(a_twab);
        });
        fft_inlane<R>(re, im);
        lds_wait();
        lds_dep(tw);
        static_for<1, R>([&](auto k1_c) {
          constexpr int k1 = decltype(k1_c)::value;
          constexpr int p = bitrev(k1, kBits);
          const float tr = re[p] * tw[k1 - 1].x - im[p] * tw[k1 - 1].y;
          const float ti = re[p] * tw[k1 - 1].y + im[p] * tw[k1 - 1].x;
          re[p] = tr;
          im[p] = ti;
        });
      }
      // ================= 16 x 16 transpose through the wavefront's scratch =================
      // DS instructions of one wavefront execute in order: the stores need no wait before the reads
      {
        v2f t[R / 2];
        static_for<0, R>([&](auto k1_c) {
          constexpr int k1 = decltype(k1_c)::value;
          (k1 < 8 ? wlo : whi)[k1 * kWRowPad] = re[bitrev(k1, kBits)];
        });
        static_for<0, R / 2>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value;
          t[c] = ds_rd64_mem<8 * c>(a_rrow);
        });
        lds_wait();
        lds_dep(t);
        static_for<0, R / 2>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value;
          re[2 * c] = t[c].x;
          re[2 * c + 1] = t[c].y;
        });
        static_for<0, R>([&](auto k1_c) {
          constexpr int k1 = decltype(k1_c)::value;
          (k1 < 8 ? wlo : whi)[k1 * kWRowPad] = im[bitrev(k1, kBits)];
        });
        static_for<0, R / 2>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value;
          t[c] = ds_rd64_mem<8 * c>(a_rrow);
        });
        lds_wait();
        lds_dep(t);
        static_for<0, R / 2>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value;
          im[2 * c] = t[c].x;
          im[2 * c + 1] = t[c].y;
        });
      }
      // ============================ FFT pass B ============================
      v2f tu[R / 2];  // untangle twiddles, in flight under the in-lane FFT
      static_for<0, R / 2>([&](auto k2_c) {
        constexpr int k2 = decltype(k2_c)::value;
        tu[k2] = ds_rd64<8 * R * k2>(a_twu);
      });
      fft_inlane<R>(re, im);  // Z[sigma + 16 k2] at bitrev(k2)
      lds_wait();
      lds_dep(tu);

      // ============== untangle to the real spectrum, power -> the wavefront's four rows ==============
      // X[k] = E + W_k O and X[Nc - k] = conj(E - W_k O), E = Z[k] + conj Z[Nc - k], O = (Z[k] - conj Z[Nc - k]) / i
      // (the window carries the 1/2).  Z[Nc - k] is register 15 - k2 of the mirrored lane; lane 0 (residue 0) pairs
      // its own registers k2 and 16 - k2, lane 15 (residue 8) its own k2 and 15 - k2.
      static_for<0, R / 2>([&](auto k2_c) {
        constexpr int k2 = decltype(k2_c)::value;
        constexpr int pz = bitrev(k2, kBits);
        constexpr int po = bitrev(R - 1 - k2, kBits);
        constexpr int ps = bitrev((R - k2) % R, kBits);
#if SAPR_WAVE_CNDDPP == 2
        float tr_, ti_;
        mirror_unless2_e64(re[po], im[po], special_mask, tr_, ti_);
        const float prr = is0 ? re[ps] : tr_, pii = is0 ? im[ps] : ti_;
#elif SAPR_WAVE_CNDDPP
        // fetch and select in one instruction: v_cndmask_b32 with a DPP source takes the mirrored lane's value except
        // where vcc (lanes 0 and 15 of every row) keeps the lane's own; lane 0 then swaps in its register 16 - k2
        float tr_, ti_;
        mirror_unless2(re[po], im[po], special_mask, tr_, ti_);
        const float prr = is0 ? re[ps] : tr_, pii = is0 ? im[ps] : ti_;
#else
        const float selfr = is0 ? re[ps] : re[po], selfi = is0 ? im[ps] : im[po];
        const float mr = dpp_mov<0x140>(re[po]), mi = dpp_mov<0x140>(im[po]);  // row_mirror
        const float prr = special ? selfr : mr, pii = special ? selfi : mi;
#endif
        const float zr = re[pz], zi = im[pz];
        const float er = zr + prr, ei = zi - pii;
        const float o_r = zi + pii, o_i = prr - zr;
        const float wx = tu[k2].x, wy = tu[k2].y;
        // E + W O by two fused multiply-adds per component, E - W O = 2 E - (E + W O) by one: six instructions where
        // forming W O first took eight
        const float ar = __builtin_fmaf(wx, o_r, __builtin_fmaf(-wy, o_i, er));
        const float ai = __builtin_fmaf(wx, o_i, __builtin_fmaf(wy, o_r, ei));
        const float br = __builtin_fmaf(2.0f, er, -ar), bi2 = __builtin_fmaf(2.0f, ei, -ai);
        prow[sigma + R * k2] = ar * ar + ai * ai;
        prow[kNc - sigma - R * k2] = br * br + bi2 * bi2;  // k == 0: the Nyquist bin
      });
      if (is0) {  // the self-paired middle bin Nc/2
        constexpr int pm = bitrev(R / 2, kBits);
        prow[kNc / 2] = 4.f * (re[pm] * re[pm] + im[pm] * im[pm]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // next set's samples: the FFT registers are free, the loads land under the filterbank phase
      if (s + 1 < s_hi) issue(s + 1);

      // ========================= mel filterbank on 16 4x4 MFMA blocks =========================
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      {
        float4 a[S4], bv[S4];
#if SAPR_WAVE_FRAG64
        // 16-byte operands as two hand-issued ds_read_b64 each: a ds_read_b128 moves fewer bytes per clock
        // (scripts/ubench/mix_rate: 0.041 per cycle and SIMD against 0.145 for ds_read_b64)
        {
          const unsigned a_aq = lds_addr(aq), a_bq = lds_addr(bq);
          v2f t[4 * S4];
          static_for<0, S4>([&](auto q_c) {
            constexpr int q = decltype(q_c)::value;
            t[4 * q] = ds_rd64<q * kWave * 16>(a_aq);
            t[4 * q + 1] = ds_rd64<q * kWave * 16 + 8>(a_aq);
            t[4 * q + 2] = ds_rd64_mem<q * 16>(a_bq);
            t[4 * q + 3] = ds_rd64_mem<q * 16 + 8>(a_bq);
          });
          lds_wait();
          lds_dep(t);
#pragma unroll
          for (int q = 0; q < S4; ++q) {
            a[q] = float4{t[4 * q].x, t[4 * q].y, t[4 * q + 1].x, t[4 * q + 1].y};
            bv[q] = float4{t[4 * q + 2].x, t[4 * q + 2].y, t[4 * q + 3].x, t[4 * q + 3].y};
          }
        }
#else
#pragma unroll
        for (int q = 0; q < S4; ++q) {
          a[q] = aq[q * kWave];
          bv[q] = bq[q];
        }
#endif
#pragma unroll
        for (int q = 0; q < S4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[q].x, bv[q].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[q].y, bv[q].y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[q].z, bv[q].z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[q].w, bv[q].w, acc1, 0, 0, 0);
        }
      }
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = acc0[i] + acc1[i];
      // parts of a group sit on neighbouring blocks of one DPP row, head first: fold them into the head
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] += f1 * dpp_mov<0x104>(e[i]);  // row_shl:4 = the next block
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] += f2 * dpp_mov<0x108>(e[i]);  // row_shl:8
      const int t = 4 * s + j4;
      if (mel0 >= 0 && t < T) {
        // 10 log10(x) = (10 log10 2) log2(x); the argument is >= amin = 1e-10, a normal number, so the bare
        // v_log_f32 needs none of the denormal scaling __log10f wraps around it
        constexpr float kDb = 3.01029995663981195f;
        float4 v;
        v.x = kDb * __builtin_amdgcn_logf(fmaxf(P.amin, e[0]));
        v.y = kDb * __builtin_amdgcn_logf(fmaxf(P.amin, e[1]));
        v.z = kDb * __builtin_amdgcn_logf(fmaxf(P.amin, e[2]));
        v.w = kDb * __builtin_amdgcn_logf(fmaxf(P.amin, e[3]));
        // (address from an opaque copy of mel0: a per-lane base kept across the set is the first thing the register
        // allocator spills, and a reload here would wait for the sample prefetch issued above)
        int mel0_here = mel0;
        asm volatile("" : "+v"(mel0_here));
        *reinterpret_cast<float4 *>(lm_out + ((f_beg + t) * P.n_mels + mel0_here)) = v;
        run_max = fmaxf(run_max, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
      }
    }
    // ===================== utterance maximum = top_db reference of the finish pass =====================
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) run_max = fmaxf(run_max, __shfl_xor(run_max, o, kWave));
    if (lane == 0) {  // mfcc_wave_finish_kernel follows
      if (s_lo == 0 && s_hi == n_sets)
        gmax_enc[u] = enc_ordered(run_max);
      else
        atomicMax(gmax_enc + u, enc_ordered(run_max));
    }
  }
}
