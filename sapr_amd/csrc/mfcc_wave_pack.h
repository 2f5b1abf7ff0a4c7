// Host side of the wave-private MFCC core (mfcc_wave.h): packs the banded mel filterbank into the 16 blocks of
// v_mfma_f32_4x4x1_16b_f32.  Plain C++ (no HIP), so that tests/test_wave_pack_cpu.py can compile and check it with g++.
#pragma once

#include <algorithm>
#include <cstddef>
#include <functional>
#include <vector>

constexpr int kWPRow = 272;               // floats between the four power rows (68 quads == 4 mod 16: the b128
                                          // B-operand reads of four frames touch different bank quads)
constexpr int kWMinS4 = 6, kWMaxS4 = 8;  // instantiated quads of filterbank steps per block

// ------------------------------------------------------------------------------------------
// host: pack the banded filterbank into 16 MFMA blocks
// ------------------------------------------------------------------------------------------
// Groups of four mels; group g needs the bins [lo4_g, hi_g] (lo4 a multiple of 4).  With S4 quads of steps per
// block a group takes ceil(quads / S4) parts; the parts of a group must sit on neighbouring blocks of one
// four-block row (DPP row of 16 lanes).  Returns the smallest feasible S4 <= kWMaxS4 or 0.
struct WavePack {
  int s4 = 0;
  int conflict_free_passes = 0;  // of the four ds_read_b128 passes of a B-operand read
  std::vector<int> blk;    // [16][4]: first bin of the block's step 0, head mel or -1, next block continues, next but one
  std::vector<float> a;    // [s4][64][4]
};

inline bool wave_pack_rows(const std::vector<int> &parts, size_t g, int (&room)[4], std::vector<int> &row_of) {
  if (g == parts.size()) return true;
  for (int r = 0; r < 4; ++r) {
    if (room[r] < parts[g]) continue;
    bool dup = false;  // rows with equal room are interchangeable
    for (int q = 0; q < r; ++q) dup = dup || room[q] == room[r];
    if (dup) continue;
    room[r] -= parts[g];
    row_of[g] = r;
    if (wave_pack_rows(parts, g + 1, room, row_of)) return true;
    room[r] += parts[g];
  }
  return false;
}

inline WavePack wave_pack(const std::vector<float> &mel, int n_mels, int nb) {
  WavePack wp;
  if (n_mels % 4 != 0 || n_mels > 64 || nb != 257) return wp;
  const int G = n_mels / 4;
  std::vector<int> lo4(G), quads(G);
  for (int g = 0; g < G; ++g) {
    int lo = nb, hi = -1;
    for (int m = 4 * g; m < 4 * g + 4; ++m)
      for (int b = 0; b < nb; ++b)
        if (mel[static_cast<size_t>(m) * nb + b] != 0.f) {
          lo = b < lo ? b : lo;
          hi = b > hi ? b : hi;
        }
    if (hi < 0) lo = hi = 0;
    lo4[g] = lo / 4 * 4;
    quads[g] = (hi - lo4[g]) / 4 + 1;
  }
  for (int s4 = kWMinS4; s4 <= kWMaxS4; ++s4) {
    std::vector<int> parts(G), row_of(G, -1);
    int total = 0;
    bool ok = true;
    for (int g = 0; g < G; ++g) {
      parts[g] = (quads[g] + s4 - 1) / s4;
      total += parts[g];
      ok = ok && parts[g] <= 4;
    }
    if (!ok || total > 16) continue;
    // big items first
    std::vector<int> order(G);
    for (int g = 0; g < G; ++g) order[g] = g;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return parts[x] > parts[y]; });
    std::vector<int> sorted(G);
    for (int g = 0; g < G; ++g) sorted[g] = parts[order[g]];
    int room[4] = {4, 4, 4, 4};
    std::vector<int> row_sorted(G, -1);
    if (!wave_pack_rows(sorted, 0, room, row_sorted)) continue;
    for (int g = 0; g < G; ++g) row_of[order[g]] = row_sorted[g];
    // Which block hosts which part is free up to: a group's parts on neighbouring blocks of one row, head first.
    // Try every order of the rows and of the groups inside a row and keep the placement whose B-operand reads are
    // conflict-free: one ds_read_b128 serves 16 lanes = four blocks x four frames, the frames' rows lie 4 bank quads
    // apart, so the four blocks of a pass must start in different quads mod 4.  A block's first bin k0 may be any
    // multiple of 4 with k0 <= p_lo, k0 + 4 s4 >= p_hi and the reads inside the power row.
    static const int kPass[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
    std::vector<std::vector<int>> rows(4);
    for (int g = 0; g < G; ++g) rows[row_of[g]].push_back(g);
    int best_ok = -1, best_plo[16], best_phi[16], best_grp[16], best_k0[16], best_flags[16][3];
    std::vector<int> rperm = {0, 1, 2, 3};
    do {
      std::vector<std::vector<int>> rr(4);
      for (int r = 0; r < 4; ++r) {
        rr[r] = rows[rperm[r]];
        std::sort(rr[r].begin(), rr[r].end());
      }
      std::function<void(int)> orders = [&](int r) {
        if (best_ok == 4) return;
        if (r == 4) {
          int p_lo[16], p_hi[16], grp_of[16], flags[16][3];
          for (int b = 0; b < 16; ++b) p_lo[b] = p_hi[b] = 0, grp_of[b] = -1, flags[b][0] = -1, flags[b][1] = flags[b][2] = 0;
          for (int r2 = 0; r2 < 4; ++r2) {
            int slot = 0;
            for (int g : rr[r2]) {
              const int np = parts[g], per = (quads[g] + np - 1) / np;  // quads per part
              for (int p2 = 0; p2 < np; ++p2, ++slot) {
                const int b = 4 * r2 + slot;
                p_lo[b] = lo4[g] + 4 * (p2 * per);
                p_hi[b] = lo4[g] + 4 * std::min(quads[g], (p2 + 1) * per);
                grp_of[b] = g;
                flags[b][0] = p2 == 0 ? 4 * g : -1;
                flags[b][1] = p2 + 1 < np ? 1 : 0;
                flags[b][2] = p2 + 2 < np ? 1 : 0;
              }
            }
          }
          int k0[16], ok = 0;
          for (int b = 0; b < 16; ++b) k0[b] = std::min(p_lo[b], kWPRow - 4 * s4);
          for (const auto &pass : kPass) {
            int cur[4];
            bool found = false;
            std::function<void(int, unsigned)> rec = [&](int i, unsigned used) {
              if (found) return;
              if (i == 4) {
                found = true;
                for (int x = 0; x < 4; ++x) k0[pass[x]] = cur[x];
                return;
              }
              const int b = pass[i];
              const int hi = std::min(p_lo[b], kWPRow - 4 * s4), lo = std::max(0, p_hi[b] - 4 * s4);
              for (int k = hi; k >= lo && k > hi - 16; k -= 4) {
                const unsigned bit = 1u << ((k / 4) & 3);
                if (used & bit) continue;
                cur[i] = k;
                rec(i + 1, used | bit);
              }
            };
            rec(0, 0u);
            ok += found ? 1 : 0;
          }
          if (ok > best_ok) {
            best_ok = ok;
            for (int b = 0; b < 16; ++b) {
              best_plo[b] = p_lo[b], best_phi[b] = p_hi[b], best_grp[b] = grp_of[b], best_k0[b] = k0[b];
              for (int x = 0; x < 3; ++x) best_flags[b][x] = flags[b][x];
            }
          }
          return;
        }
        do {
          orders(r + 1);
        } while (best_ok < 4 && std::next_permutation(rr[r].begin(), rr[r].end()));
      };
      orders(0);
    } while (best_ok < 4 && std::next_permutation(rperm.begin(), rperm.end()));
    wp.s4 = s4;
    wp.conflict_free_passes = best_ok;
    wp.blk.assign(16 * 4, 0);
    wp.a.assign(static_cast<size_t>(s4) * 64 * 4, 0.f);
    for (int b = 0; b < 16; ++b) {
      wp.blk[4 * b + 0] = best_k0[b];
      wp.blk[4 * b + 1] = best_flags[b][0];
      wp.blk[4 * b + 2] = best_flags[b][1];
      wp.blk[4 * b + 3] = best_flags[b][2];
      if (best_grp[b] < 0) continue;
      for (int step = 0; step < 4 * s4; ++step) {
        const int bin = best_k0[b] + step;
        if (bin < best_plo[b] || bin >= best_phi[b] || bin >= nb) continue;
        for (int i = 0; i < 4; ++i)
          wp.a[(static_cast<size_t>(step / 4) * 64 + 4 * b + i) * 4 + step % 4] =
              mel[static_cast<size_t>(4 * best_grp[b] + i) * nb + bin];
      }
    }
    return wp;
  }
  return wp;
}
