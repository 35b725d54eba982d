// Tile body of the windowed convolution (conv_window.hip wraps it in a kernel).
//
// Dilated k=5 convolution of the ByteNet block with the activation window kept in LDS (gfx950, bf16).
//
// gemm_bf16.hip treats the convolution as a GEMM over (tap, 64-channel slice) steps and copies the 256 activation
// rows of every step from L2 again although the five taps of a slice read almost the same rows: rows
// [l0 - 2d, l0 + 256 + 2d) shifted by 0, d, .. 4d.  With 576 rows to copy per step the LDS-DMA path (one 1-KiB
// copy instruction per ~40 cycles per CU) is busy longer than the matrix pipe (2880 vs 2560 cycles per step), so
// the waves queue on it.  Here the K loop runs slice-major: the WINDOW of 256 + 4d rows of a 64-channel slice is
// copied once (spread over the five tap steps of the previous slice, one copy per wave per step) and each tap step
// only brings in its 320 (256) weight rows.  Copies per step and wave drop from 9 to 6 (5 with N-tile 256), the
// copy path is then shorter than the MFMA path, and the A fragments of tap t are simply read t*d rows further down
// the window.  Same tile shape, swizzle and pinned issue order as gemm_bf16.hip.
//
// MFMA shape.  The loop is paced by the matrix pipe (~86 % busy in cycles) and the chip answers by lowering its
// clock: measured with in-kernel stamps (ed_debug_gemm kernel 9), the 32x32x16 form of this loop ran at
// 1.64-1.74 GHz, the same FLOPs issued as 16x16x32 MFMAs at 1.96-2.00 GHz at equal cycles (-11 % loop time).
// Hence v_mfma_f32_16x16x32_bf16, with the weights as the row operand so that a lane ends up with four consecutive
// output channels of one position (8/16-byte stores in the epilogue).
//
// LDS: 2 windows x 320 rows x 128 B + 2 weight stages x BN rows x 128 B = 160 KiB at BN = 320: the window is
// limited to 320 rows, i.e. dilation <= 16 (five of the eight dilations of a ByteNet cycle) in general.
//
// Whole-sequence form (L <= 256: one tile per sequence; dilation 32, 64, 128): everything outside the tile is zero
// padding, so the window is the sequence itself (+ 32 rows either side at dilation 32).  Taps that no row of the
// tile can reach leave the K loop, and a wave skips a tap that moves all of its 64 rows into the padding (it still
// issues its share of the copies); its SIMD partner then has the matrix pipe to itself.  See window_plan().
#pragma once
#include "kernels.hpp"
#include "ln_epilogue.hpp"

#include <type_traits>

namespace ed {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace conv_tile {


constexpr int SEQ_ROWS = 256, BK = 64, ROW_B = BK * 2;
constexpr int WIN_ROWS = 320, WIN_BYTES = WIN_ROWS * ROW_B;
constexpr int NTAPS = 5;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

constexpr int lds_bytes(int nt) { return 2 * WIN_BYTES + 2 * nt * 64 * ROW_B; }


// STAMP: diagnostic build that also records the K loop's duration (GemmProblem::stamps).
// WM: waves along M, as in gemm_bf16.hip: 4 = 256-row tile, 2 = 128-row tile (4 waves) for launches that would
// leave half of the CUs without a 256-row tile.
// LNF: the LayerNorm + GELU behind the convolution (g.ln) is applied in the epilogue (ln_epilogue.hpp) and only the
// normalised rows are stored.
// `lds`: lds_bytes(NT) bytes, 1024-byte aligned; bid_x: the workgroup's index in the launch grid.
template <int NT, bool OUT_BF16, bool STAMP = false, int WM = 4, bool LNF = false>
__device__ __forceinline__ void tile(const GemmProblem& g, int tiles_n, int tiles_total, const WindowPlan& wp, char* lds,
                                     int bid_x) {
  constexpr int BM = WM * 64;      // tile rows
  constexpr int NW = 2 * WM;       // waves
  constexpr int BN = NT * 64;
  constexpr int W_BYTES = BN * ROW_B;
  constexpr int NGB = BN / (8 * NW);  // 8-row copy groups of the weight tile per wave and step
  constexpr int NJ = 2 * NT;       // 16-column output tiles per wave
  constexpr int NGRP = 2 * NJ;     // (32-channel k-slice, column tile) groups of 4 MFMAs per step
  constexpr int RING = 4;          // weight fragments in flight
  static_assert(NGRP % RING == 0, "the fragment ring must line up across steps");
  static_assert(NGB + 3 < NGRP, "copy slots of a step");
  char* const lds_w = lds + 2 * WIN_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  unsigned long long r_entry = 0;
  if constexpr (STAMP) r_entry = __builtin_amdgcn_s_memrealtime();

  // XCD-aware tile order (as gemm_bf16.hip)
  int tile;
  {
    const int id = bid_x, q = tiles_total >> 3, r = tiles_total & 7, x = id & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int tiles_per_seq = (g.L + BM - 1) / BM;
  const int b = tile_m / tiles_per_seq;
  const int l0 = (tile_m - b * tiles_per_seq) * BM;
  const int n0 = tile_n * BN;
  const int ksteps = g.K / BK, d = g.dilation;
  const int win_groups = wp.win_groups, ntaps = wp.tap_hi - wp.tap_lo + 1;

  // ---- copy addressing: lane -> (row in 8-row group, physical 16-B chunk); logical chunk = physical ^ swizzle,
  // swizzle of window / tile row p = (p >> 1) & 7 = 4 * (group parity) + (lane >> 4)
  const long row_bytes = (long)g.K * 2;
  const unsigned lane_row = (unsigned)((lane >> 3) * row_bytes);
  const unsigned lane_off0 = lane_row + (((lane & 7) ^ (lane >> 4)) << 4);
  const unsigned lane_off1 = lane_row + (((lane & 7) ^ (4 + (lane >> 4))) << 4);
  const unsigned grp_bytes = (unsigned)(8 * row_bytes);
  // window row 0 = position lw - wofs of sequence b
  const int lw = wp.clip ? 0 : l0, lt = l0 - lw;  // window origin; tile's first position relative to it
  const char* a_next = (const char*)g.A + ((long)b * g.a_seq_pitch + g.a_row0 + lw - wp.wofs) * row_bytes;
  const char* w_dma = (const char*)g.W + ((long)wp.tap_lo * g.N + n0) * row_bytes;  // first tap, slice 0, first channel
  const unsigned w_lane_extra = (unsigned)((lane >> 5) * 4 * row_bytes);  // permuted rows: group row k reads channel + 4*(k >> 2)
  const long w_tap_step = (long)g.N * row_bytes;
  const long w_wrap_step = ROW_B - (long)(ntaps - 1) * g.N * row_bytes;

  auto copy_window_group = [&](int ga_raw, char* win) {  // groups past the window re-copy its last one
    const int ga = ga_raw < win_groups ? ga_raw : win_groups - 1;
    glds16(a_next + (unsigned)ga * grp_bytes + ((ga & 1) ? lane_off1 : lane_off0), win + ga * 1024);
  };
  auto copy_weight_group = [&](int jj, char* wst) {
    const int grp = wid * NGB + jj;  // LDS rows 8*grp .. +7 <- channels c0 + {0..3, 8..11}
    const int c0 = 32 * (grp >> 2) + 16 * (grp & 1) + 4 * ((grp >> 1) & 1);
    glds16(w_dma + (long)c0 * row_bytes + (((grp & 1) ? lane_off1 : lane_off0) + w_lane_extra), wst + grp * 1024);
  };

  // ---- MFMA 16x16x32 with the WEIGHTS as the row operand: D[n][pos], lane (fr, fq) holds columns pos = fr of
  // rows n = 4*fq .. 4*fq+3 of a 16 x 16 tile.
  // Accumulators start at the bias: the epilogue is a pure store.
  const int fr = lane & 15, fq = lane >> 4;
  const bool full_rows = l0 + BM <= g.L;
  // The weight tile sits in LDS with its rows permuted (by the copy's source addresses): LDS row 32J + 16t + i holds
  // channel 32J + 8*(i >> 2) + 4t + (i & 3), so that column tiles 2J and 2J+1 together give a lane EIGHT consecutive
  // channels, 32J + 8*fq + 0..7: one 16-byte store per pair for bf16 outputs (the store tail is issue-bound).
  const int nb = n0 + wc * (BN / 2) + 8 * fq;  // this lane's first channel
  f32x4 acc[NJ][4];
#pragma unroll
  for (int jn = 0; jn < NJ; ++jn) {
    const f32x4 bj = *(const f32x4*)(as_global(g.bias) + nb + 32 * (jn >> 1) + 4 * (jn & 1));
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) acc[jn][ip] = bj;
  }

  // ---- fragment addressing: 16 rows x 32 channels per fragment, lane = (row fr, 8-channel chunk fq)
  const int b_rd = (wc * (BN / 2) + fr) * ROW_B;
  const int b_x = (fq ^ ((fr >> 1) & 7)) << 4;
  const int a_row = lt + wr * 64 + fr + wp.wofs - wp.center * d;  // window row of this lane's first activation fragment at tap 0
  bf16x8 fact[2][4], fw[RING];

  // prologue: window of slice 0 and weights of step 0
  {
#pragma unroll
    for (int k = 0; k < WIN_ROWS / (8 * NW); ++k) copy_window_group(k * NW + wid, lds);
#pragma unroll
    for (int jj = 0; jj < NGB; ++jj) copy_weight_group(jj, lds_w);
    w_dma += ntaps > 1 ? w_tap_step : w_wrap_step;
    a_next += ksteps > 1 ? ROW_B : 0;
    __syncthreads();
  }

  // One step = tap TAP of one 64-channel slice: 2 k-slices of 32 x NJ column tiles x 4 position tiles = 8*NJ MFMAs.
  // The issue order is pinned; behind each MFMA comes at most one auxiliary instruction: the read of the weight
  // fragment three groups ahead, a read of the second k-slice's activation fragments, or one of the step's copies.
  auto step = [&](int tap, int j, const char* win, char* win_next, const char* wst, char* wst_next) {
    // whole-sequence form: rows [64 wr + (tap-2) d, +64) of this wave; all of them outside [0, 256) = all zeros
    const int first = lt + wr * 64 + (tap - 2) * d;
    const bool active = !wp.clip || (first + 64 > 0 && first < SEQ_ROWS);
    if (active) {
      const int shift = tap * d;  // window rows between tap 0 and this tap
      const char* a_base = win + (a_row + shift) * ROW_B;
      const int a_x = (fq ^ (((a_row + shift) >> 1) & 7)) << 4;  // chunk of k-slice s: (s << 6) ^ a_x
      const char* b_base = wst + b_rd;
      auto read_act = [&](int s2, int ip) { fact[s2][ip] = *(const bf16x8*)(a_base + ip * 16 * ROW_B + ((s2 << 6) ^ a_x)); };
      auto read_w = [&](int grp) {
        const int s2 = grp / NJ, jn = grp - s2 * NJ;
        fw[grp % RING] = *(const bf16x8*)(b_base + jn * 16 * ROW_B + ((s2 << 6) ^ b_x));
      };
#pragma unroll
      for (int ip = 0; ip < 4; ++ip) read_act(0, ip);
#pragma unroll
      for (int grp = 0; grp < RING - 1; ++grp) read_w(grp);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int grp = 0; grp < NGRP; ++grp) {
        const int s2 = grp / NJ, jn = grp - s2 * NJ;
#pragma unroll
        for (int ip = 0; ip < 4; ++ip) {
          acc[jn][ip] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[grp % RING], fact[s2][ip], acc[jn][ip], 0, 0, 0);
          if (ip == 0) {
            if (grp + RING - 1 < NGRP) read_w(grp + RING - 1);
          } else if (ip == 1) {
            if (grp >= 1 && grp <= 4) read_act(1, grp - 1);
          } else if (ip == 2) {
            if (grp >= 1 && grp <= NGB) copy_weight_group(grp - 1, wst_next);
            else if (grp == NGB + 1) copy_window_group((j * wp.na) * NW + wid, win_next);
            else if (grp == NGB + 2 && wp.na > 1) copy_window_group((j * wp.na + 1) * NW + wid, win_next);
            else if (grp == NGB + 3 && wp.na > 2) copy_window_group((j * wp.na + 2) * NW + wid, win_next);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // nothing to multiply in this step: only this wave's share of the copies
#pragma unroll
      for (int jj = 0; jj < NGB; ++jj) copy_weight_group(jj, wst_next);
      copy_window_group((j * wp.na) * NW + wid, win_next);
      if (wp.na > 1) copy_window_group((j * wp.na + 1) * NW + wid, win_next);
      if (wp.na > 2) copy_window_group((j * wp.na + 2) * NW + wid, win_next);
    }
  };

  const int T = ksteps * ntaps;
  int t = 0;  // global step index
  unsigned long long c0 = 0, r0 = 0;
  if constexpr (STAMP) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  for (int kc = 0; kc < ksteps; ++kc) {
    const char* win = lds + (kc & 1) * WIN_BYTES;
    char* win_next = lds + ((kc + 1) & 1) * WIN_BYTES;
    for (int j = 0; j < ntaps; ++j) {
      step(wp.tap_lo + j, j, win, win_next, lds_w + (t & 1) * W_BYTES, lds_w + ((t + 1) & 1) * W_BYTES);
      ++t;
      // weights of step t+1: next tap, or the first tap of the next slice; past the end the stream stalls
      w_dma += t + 1 < T ? (j == ntaps - 2 || ntaps == 1 ? w_wrap_step : w_tap_step) : 0L;
      __syncthreads();  // copies of this step landed (the fence drains them); stages free
    }
    a_next += kc + 2 < ksteps ? ROW_B : 0;
  }
  // diagnostic build only (eight values per tile): loop duration in shader cycles and in 100-MHz ticks, then the
  // absolute 100-MHz times of entry, loop start, loop end / statistics published, and exit (stores drained)
  unsigned long long r1 = 0;
  if constexpr (STAMP) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      g.stamps[8 * tile] = 0;
      g.stamps[8 * tile + 1] = (long long)(c1 - c0);
      g.stamps[8 * tile + 2] = 0;
      g.stamps[8 * tile + 3] = (long long)(r1 - r0);
      g.stamps[8 * tile + 4] = (long long)r_entry;
      g.stamps[8 * tile + 5] = (long long)r0;
      g.stamps[8 * tile + 6] = (long long)r1;
    }
  }

  // ---- epilogue: four consecutive channels of one position per lane and accumulator.  The row base is laundered
  // through an empty asm so that the addresses are computed here and not kept live across the K loop.
  int l0e = l0;
  asm volatile("" : "+s"(l0e));
  if constexpr (LNF) {
    const unsigned ln_target = ln_publish<NT, WM>(g, acc, lds, tile_m, tile_n, tiles_n, tid);
    if constexpr (STAMP) {
      if (tid == 0) g.stamps[8 * tile + 6] = (long long)__builtin_amdgcn_s_memrealtime();  // statistics published
    }
    ln_finish<NT, WM, false>(g, acc, lds, ln_target, tile_m, tiles_n, (long)b * g.L + l0e,
                             g.L - l0e < BM ? g.L - l0e : BM, n0, tid, STAMP ? g.stamps + 8 * tile : nullptr);
    if constexpr (STAMP) {
      __builtin_amdgcn_s_waitcnt(0);
      if (tid == 0) g.stamps[8 * tile + 7] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    return;
  }
#pragma unroll
  for (int ip = 0; ip < 4; ++ip) {
    const int l = l0e + wr * 64 + ip * 16 + fr;
    if (full_rows || l < g.L) {
      const long o = ((long)b * g.L + l) * g.N + nb;
#pragma unroll
      for (int jp = 0; jp < NT; ++jp) {  // column tiles 2jp, 2jp+1 hold channels nb + 32jp + 0..3, + 4..7
        const f32x4 v = acc[2 * jp][ip], w = acc[2 * jp + 1][ip];
        if constexpr (OUT_BF16) {
          uint4 pk;
          pk.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
          pk.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
          pk.z = (unsigned)f32_to_bf16_bits(w[0]) | ((unsigned)f32_to_bf16_bits(w[1]) << 16);
          pk.w = (unsigned)f32_to_bf16_bits(w[2]) | ((unsigned)f32_to_bf16_bits(w[3]) << 16);
          *(uint4*)(as_global((bf16_t*)g.out) + o + 32 * jp) = pk;
        } else {
          *(f32x4*)(as_global((float*)g.out) + o + 32 * jp) = v;
          *(f32x4*)(as_global((float*)g.out) + o + 32 * jp + 4) = w;
        }
      }
    }
  }
  if constexpr (STAMP) {
    __builtin_amdgcn_s_waitcnt(0);
    if (tid == 0) g.stamps[8 * tile + 7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace conv_tile
}  // namespace ed
