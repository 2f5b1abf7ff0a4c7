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
#pragma once

constexpr int kWRowPad = 18;              // floats per transpose row
constexpr int kWGroup = 16 * kWRowPad;    // transpose scratch of one frame group
constexpr int kWRegion = 4 * kWGroup;     // floats of LDS scratch per wavefront (1152)
#include "mfcc_wave_pack.h"  // kWPRow, kWMinS4, kWMaxS4, wave_pack()

struct WaveLds {
  int win, twab, twu, mela, scr, total;
};
__host__ __device__ inline WaveLds wave_lds(int s4) {
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
  L.scr = o;
  o += kWaves * kWRegion * 4;
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
  float m[PREEMPH ? NR : 1];  // sample before each pair (pre-emphasis)
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

#ifndef SAPR_WAVE_OCC
#define SAPR_WAVE_OCC 4  // wavefronts per SIMD the register allocation aims at
#endif

template <bool PREEMPH, int RLO, int RHI, int S4>
__global__ __launch_bounds__(kThreads, SAPR_WAVE_OCC) void mfcc_wave_kernel(
    const float *__restrict__ pcm, const int64_t *__restrict__ sample_offsets,
    const int64_t *__restrict__ frame_offsets, int64_t n_utts, MfccDev P, float *__restrict__ lm_out,
    unsigned *__restrict__ gmax_enc, int split) {
  constexpr int R = 16, kNc = 256, kBits = 4, NR = RHI - RLO;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const WaveLds L = wave_lds(S4);
  float *s_win = reinterpret_cast<float *>(smem + L.win);
  float2 *s_twab = reinterpret_cast<float2 *>(smem + L.twab);
  float2 *s_twu = reinterpret_cast<float2 *>(smem + L.twu);
  float4 *s_a = reinterpret_cast<float4 *>(smem + L.mela);
  float *s_scr = reinterpret_cast<float *>(smem + L.scr);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int lane = tid % kWave;
  const int grp = lane >> 4;  // frame of the set (FFT layout)
  const int l = lane & 15;    // lane inside the frame's DPP row

  // ---- read-only tables -> LDS, scratch zeroed (filterbank K padding reads it times a zero weight) ----
  for (int i = tid; i < 512; i += kThreads) s_win[i] = P.window[i];
  for (int i = tid; i < 256; i += kThreads) s_twab[i] = P.tw_ab[i];
  for (int i = tid; i < 128; i += kThreads) s_twu[i] = P.tw_u[i];
  for (int i = tid; i < S4 * kWave; i += kThreads) s_a[i] = reinterpret_cast<const float4 *>(P.wave_a)[i];
  for (int i = tid; i < kWaves * kWRegion; i += kThreads) s_scr[i] = 0.f;
  __syncthreads();  // the only workgroup barrier

  // pass B: lane l plays residue sigma (k = sigma + 16 k2); the conjugate partner 16 - sigma sits in lane 15 - l,
  // the self-paired residues 0 and 8 in lanes 0 and 15
  const int sigma = l < 8 ? l : (l == 15 ? 8 : l + 1);
  const bool is0 = l == 0, special = l == 0 || l == 15;
  float *region = s_scr + wave * kWRegion;
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

  const int n_waves = gridDim.x * kWaves;
  const int wid = blockIdx.x * kWaves + wave;
  const int n_teams = n_waves / split;  // `split` wavefronts share an utterance
  const int part = wid % split;
  if (wid >= n_teams * split) return;

  for (int64_t u = wid / split; u < n_utts; u += n_teams) {
    const int64_t s_beg = sample_offsets[u];
    const int n_samp = static_cast<int>(sample_offsets[u + 1] - s_beg);
    const int64_t f_beg = frame_offsets[u];
    const int T = static_cast<int>(frame_offsets[u + 1] - f_beg);
    const int n_sets = (T + 3) >> 2;
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
        if constexpr (PREEMPH)
          nxt.m[r - RLO] =
              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo + 8 * R * r - 4, 0, 0));
      });
    };
    if (part < n_sets) issue(part);

    for (int s = part; s < n_sets; s += split) {
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
              const float ya = y.x - P.preemph * nxt.m[r - RLO];
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
        const float selfr = is0 ? re[ps] : re[po], selfi = is0 ? im[ps] : im[po];
        const float mr = dpp_mov<0x140>(re[po]), mi = dpp_mov<0x140>(im[po]);  // row_mirror
        const float prr = special ? selfr : mr, pii = special ? selfi : mi;
        const float zr = re[pz], zi = im[pz];
        const float er = zr + prr, ei = zi - pii;
        const float o_r = zi + pii, o_i = prr - zr;
        const float wx = tu[k2].x, wy = tu[k2].y;
        const float wr = wx * o_r - wy * o_i, wi = wx * o_i + wy * o_r;
        const float ar = er + wr, ai = ei + wi, br = er - wr, bi2 = ei - wi;
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
      if (s + split < n_sets) issue(s + split);

      // ========================= mel filterbank on 16 4x4 MFMA blocks =========================
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      {
        float4 a[S4], bv[S4];
#pragma unroll
        for (int q = 0; q < S4; ++q) {
          a[q] = aq[q * kWave];
          bv[q] = bq[q];
        }
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
        float4 v;
        v.x = 10.0f * __log10f(fmaxf(P.amin, e[0]));
        v.y = 10.0f * __log10f(fmaxf(P.amin, e[1]));
        v.z = 10.0f * __log10f(fmaxf(P.amin, e[2]));
        v.w = 10.0f * __log10f(fmaxf(P.amin, e[3]));
        *reinterpret_cast<float4 *>(lm_out + (f_beg + t) * P.n_mels + mel0) = v;
        run_max = fmaxf(run_max, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
      }
    }
    // ===================== utterance maximum (top_db reference of mfcc_finish_kernel) =====================
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) run_max = fmaxf(run_max, __shfl_xor(run_max, o, kWave));
    if (lane == 0 && part < n_sets) {
      if (split == 1)
        gmax_enc[u] = enc_ordered(run_max);
      else
        atomicMax(gmax_enc + u, enc_ordered(run_max));
    }
  }
}

