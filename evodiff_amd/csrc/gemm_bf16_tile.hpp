// Tile body of the tuned bf16 implicit-GEMM kernel (gemm_bf16.hip wraps it in a kernel).
//
// Tuned bf16 implicit-GEMM kernel for the ByteNet block on gfx950 (MI355X).
//
//   out[b*L + l][n] = bias[n] (+ resid) + sum_{tap, k} A[b, l + (tap - ntaps/2)*dilation][k] * W[tap][n][k]
//
// Tiling.  One 512-thread workgroup (8 waves = 4 (M) x 2 (N)) owns a 256 x BN output tile, BN = 320 or 256:
// each wave a 64 x BN/2 sub-tile as 4 x 2*NT MFMA 16x16x32 bf16 tiles (NT = 5 or 4; 160 / 128 fp32
// accumulators per lane).  640M has N = 1280 = 4 x 320 and config 2 has 64 row tiles, i.e. exactly one
// tile per CU of the 256-CU chip with no tail; N = 512 / 1024 (38M) use BN = 256.
//
// K loop.  K runs over (tap, 64-channel slice) "steps".  The operand tiles of a step (256 + BN rows of 128
// bytes = whole cache lines) are copied global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 tile
// rows per wave-instruction) into one of two LDS stages, the copy of step t+1 in flight while step t is
// multiplied; one barrier per step.  The issue order inside a step is pinned (sched_barrier after every
// MFMA): behind each of the 80 (64) MFMAs of a step comes at most one auxiliary instruction - a fragment
// read for a later MFMA group or one of the step's 9 (8) DMA copies - so neither the LDS read latency nor
// the ~50-100-cycle issue cost of a copy ever opens an MFMA-free stretch in a wave.  (Left to
// itself hipcc batches: all copies at the top of the step, then reads, a full wait, then MFMAs; with the two
// waves of a SIMD in lockstep behind the same barrier that left the matrix pipe idle 45 % of the time.)
//
// LDS image.  A DMA cannot scatter (the image is lane-linear), so the bank swizzle - 16-byte chunk c of tile
// row r is stored at chunk c ^ ((r >> 1) & 7) - is applied to the per-lane SOURCE address and again on the
// ds_read_b128 fragment reads: the 16 rows of every ds_read_b128 lane group then fall on 16 different
// 16-byte slots of the 256-byte bank row (SQ_LDS_BANK_CONFLICT = 0 measured).
//
// Dilation halo.  A rows of tap t are the tile's rows shifted by (t - ntaps/2)*dilation.  The A buffer of
// the convolution is laid out per sequence with zeroed halo rows in front and behind (a_row0, a_seq_pitch),
// so a shifted tile is a plain contiguous row range and 'same' padding costs no bounds logic.  Taps whose
// whole 256-row tile lies outside [0, L) are dropped from the K loop (at L = 256, dilation 128: 2 of 5 taps).
#pragma once
#include "kernels.hpp"
#include "ln_epilogue.hpp"

namespace ed {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace gemm_tile {

constexpr int BM = 256, BK = 64, ROW_B = BK * 2;  // 128 bytes of K per tile row per step

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

constexpr int lds_bytes(int nt, int wm) { return 2 * (wm * 64 + nt * 64) * ROW_B; }

// ABLATE (diagnostic builds): 1 = no MFMAs, 2 = no DMA copies (results invalid); 3 = in-kernel clock stamps.
// RT > 0 (1x1 + residual only): the K loop has exactly RT steps and is fully unrolled, and the residual tile is
// streamed in DURING the loop (see below) instead of before it.
// WM: waves along M (2 along N always).  4 = the 256-row tile; 2 = a 128-row tile with 4 waves (one per SIMD) for
// problems that would fill less than half the chip with 256-row tiles: every wave keeps its 64 x BN/2 sub-tile and
// copies twice as many weight groups.
// SPLIT: the K loop of a tile is cut into g.ksplit ranges (bid_y); range z accumulates from zero and stores raw
// fp32 sums to g.partial[z] (bias, residual and rounding happen where the ranges are added up: launch_reduce_partials
// or the LayerNorm behind the GEMM).  For problems with too few tiles to fill the chip.
// LNF: the LayerNorm + GELU behind this GEMM (g.ln) is applied in the epilogue (ln_epilogue.hpp); the raw result is
// stored only in the residual form (PFF2: it is the residual stream).
// `lds`: 2 * (WM*64 + NT*64) * 128 bytes, 1024-byte aligned; bid_x / bid_y: the workgroup's index in the launch grid.
template <int NT, bool CONV, bool OUT_BF16, bool RESID, int ABLATE = 0, int RT = 0, int WM = 4, bool SPLIT = false,
          bool LNF = false>
__device__ __forceinline__ void tile(const GemmProblem& g, int tiles_n, int tiles_total, char* lds, int bid_x, int bid_y) {
  static_assert(!SPLIT || (!OUT_BF16 && !RESID && RT == 0 && ABLATE == 0), "split form: raw fp32 partial sums only");
  static_assert(!LNF || (!SPLIT && (ABLATE == 0 || ABLATE == 3)), "fused LayerNorm: whole sums only");
  constexpr int BM = WM * 64;       // (shadows the file-scope 256)
  constexpr int BN = NT * 64;
  constexpr int A_BYTES = BM * ROW_B, B_BYTES = BN * ROW_B, STAGE = A_BYTES + B_BYTES;
  constexpr int NGB = BN / (16 * WM);  // 8-row DMA groups of the W tile per wave (A: 4 per wave)
  constexpr int N_DMA = 4 + NGB;    // copies per wave per step
  constexpr int NJ = 2 * NT;        // 16-column output tiles per wave
  constexpr int NGRP = 2 * NJ;      // (32-channel k-slice, column tile) groups of 4 MFMAs per step
  constexpr int RING = 4;           // weight fragments in flight
  static_assert(NGRP % RING == 0 && N_DMA < NGRP, "step schedule");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  unsigned long long sc0 = 0, sc1 = 0, sr0 = 0, sr_loop0 = 0;
  if constexpr (ABLATE == 3) {
    sc0 = __builtin_amdgcn_s_memtime();
    sr0 = __builtin_amdgcn_s_memrealtime();
  }

  // XCD-aware tile order: workgroups id, id+8, ... share an XCD (and its L2); give each XCD a compact
  // block of the tile grid so that the tiles resident on it share A row panels and W column panels.
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

  const int ksteps = g.K / BK, half_taps = g.center();
  int tap_lo = 0, tap_hi = g.ntaps - 1;
  while (tap_lo < g.ntaps && l0 + (tap_lo - half_taps) * g.dilation + BM <= 0) ++tap_lo;
  while (tap_hi >= 0 && l0 + (tap_hi - half_taps) * g.dilation >= g.L) --tap_hi;
  const int T_all = (tap_hi - tap_lo + 1) * ksteps;  // number of steps of the tile
  int t_first = 0, T = T_all;                       // this workgroup's range of them
  if constexpr (SPLIT) {
    const int chunk = (T_all + g.ksplit - 1) / g.ksplit;
    t_first = bid_y * chunk;
    T = (t_first + chunk < T_all ? t_first + chunk : T_all) - t_first;
    if (T < 0) T = 0;
    if (t_first > T_all) t_first = T_all;
  }
  const int tap_first = t_first / ksteps, kk_first = t_first - tap_first * ksteps;

  // ---- DMA addressing.  Lane -> (row in 8-row group, physical chunk); logical chunk = physical ^ swizzle.
  // Group bases are multiples of 8 rows, so (row >> 1) & 7 = 4 * (group parity) + (lane >> 4).
  const long row_bytes = (long)g.K * 2;
  const unsigned lane_row = (unsigned)((lane >> 3) * row_bytes);
  const unsigned lane_off0 = lane_row + (((lane & 7) ^ (lane >> 4)) << 4);
  const unsigned lane_off1 = lane_row + (((lane & 7) ^ (4 + (lane >> 4))) << 4);
  const unsigned grp_bytes = (unsigned)(8 * row_bytes);
  // wave-uniform bases of the copy stream (this wave's first A / W group), advanced step by step
  const char* a_dma = (const char*)g.A +
                      ((long)b * g.a_seq_pitch + g.a_row0 + l0 + (long)(tap_lo + tap_first - half_taps) * g.dilation + wid * 32) *
                          row_bytes + (long)kk_first * ROW_B;
  const char* w_dma = (const char*)g.W + ((long)(tap_lo + tap_first) * g.N + n0) * row_bytes + (long)kk_first * ROW_B;
  const unsigned w_lane_extra = (unsigned)((lane >> 5) * 4 * row_bytes);  // permuted rows: group row k reads channel + 4*(k >> 2)
  const long a_tap_step = (long)(g.dilation - 1) * row_bytes + ROW_B, w_tap_step = (long)(g.N - 1) * row_bytes + ROW_B;
  int dma_kk = kk_first, dma_t = 0;

  // piece j (0 .. N_DMA-1) of the copy of step dma_t into stage dma_t & 1
  auto dma_piece = [&](int j) {
    if constexpr (ABLATE == 2) return;
    char* st = lds + (dma_t & 1) * STAGE;
    if (j < 4) {  // A rows wid*32 + 8j .. +8   (group wid*4 + j, parity j & 1)
      glds16(a_dma + (unsigned)j * grp_bytes + ((j & 1) ? lane_off1 : lane_off0), st + (wid * 4 + j) * 1024);
    } else {      // W rows n0 + 8*(wid*NGB + jj) .. +8
      const int jj = j - 4, grp = wid * NGB + jj;  // LDS rows 8*grp .. +7 <- channels c0 + {0..3, 8..11} (see below)
      const int c0 = 32 * (grp >> 2) + 16 * (grp & 1) + 4 * ((grp >> 1) & 1);
      glds16(w_dma + (long)c0 * row_bytes + (((grp & 1) ? lane_off1 : lane_off0) + w_lane_extra), st + A_BYTES + grp * 1024);
    }
  };
  // Advance the copy stream to the next step (branch-free).  Past the last step it stalls on the last tile:
  // those copies stay in bounds and land in a stage nobody reads any more.
  auto dma_advance = [&]() {
    ++dma_t;
    const bool more = dma_t < T;
    const bool wrap = dma_kk + 1 == ksteps;  // next tap: rows shift by dilation, channels back to 0
    dma_kk = wrap ? 0 : dma_kk + 1;
    a_dma += more ? (wrap ? a_tap_step : (long)ROW_B) : 0L;
    w_dma += more ? (wrap ? w_tap_step : (long)ROW_B) : 0L;
  };

  // ---- MFMA 16x16x32 with the WEIGHTS as the row operand: D[n][pos]; lane (fr, fq) holds, per accumulator, the
  // four consecutive output channels n = 4*fq .. 4*fq+3 of position fr of a 16 x 16 tile (shape choice: see
  // conv_window.hip; the chip holds a higher clock on this shape than on 32x32x16).
  // Accumulators start at bias (+ residual): the epilogue is then a pure store, and the residual tile is
  // fetched here, where its latency hides behind the first DMA wait, instead of serialising the store tail
  // (measured: the add-in-the-epilogue form made the x += PFF2(...) GEMM 2x slower than the same GEMM without it).
  const int fr = lane & 15, fq = lane >> 4;
  const bool full_rows = l0 + BM <= g.L;  // no row masking needed (always true when 256 | L)
  const int rsel = RESID ? g.resid_sel.get() : 0;  // residual row of output row R: R * resid_mul + rsel
  // The weight tile sits in LDS with its rows permuted (by the copy's source addresses): LDS row 32J + 16t + i holds
  // channel 32J + 8*(i >> 2) + 4t + (i & 3), so that column tiles 2J and 2J+1 together give a lane EIGHT consecutive
  // channels, 32J + 8*fq + 0..7: one 16-byte store per pair for bf16 outputs (the store tail is issue-bound).
  const int nb = n0 + wc * (BN / 2) + 8 * fq;      // this lane's first channel
  f32x4 acc[NJ][4];
#pragma unroll
  for (int jn = 0; jn < NJ; ++jn) {
    f32x4 bj = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!SPLIT) bj = *(const f32x4*)(as_global(g.bias) + nb + 32 * (jn >> 1) + 4 * (jn & 1));
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) acc[jn][ip] = bj;
  }
  if constexpr (RESID && RT == 0) {
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
      const int l = l0 + wr * 64 + ip * 16 + fr;
      if (full_rows || l < g.L) {
        const float* pr = as_global(g.resid) + (((long)b * g.L + l) * g.resid_mul + rsel) * g.N + nb;
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) acc[jn][ip] += *(const f32x4*)(pr + 32 * (jn >> 1) + 4 * (jn & 1));
      }
    }
  }

  // ---- fragment addressing: 16 rows x 32 channels per fragment, lane = (row fr, 8-channel chunk fq); the
  // swizzle (row >> 1) & 7 only depends on fr because every fragment starts at a multiple of 16 rows
  const int f_x = (fq ^ ((fr >> 1) & 7)) << 4;       // chunk of k-slice s2: (s2 << 6) ^ f_x
  const int a_rd = (wr * 64 + fr) * ROW_B;
  const int b_rd = A_BYTES + (wc * (BN / 2) + fr) * ROW_B;
  bf16x8 fact[2][4], fw[RING];  // activation fragments [k-slice][position tile]; ring of weight fragments

  if (T > 0) {
    // prologue: copy of step 0
#pragma unroll
    for (int j = 0; j < N_DMA; ++j) dma_piece(j);
    dma_advance();
    __syncthreads();  // (its fence drains the DMA)

    // One step: 2 k-slices of 32 x NJ column tiles x 4 position tiles = 8*NJ MFMAs in (k-slice, column tile) groups
    // of 4.  The issue order is pinned; behind each MFMA comes at most one auxiliary instruction: the read of the
    // weight fragment three groups ahead, a read of the second k-slice's activation fragments, or one of the
    // step's copies into the other stage.
    if constexpr (ABLATE == 3) {
      sc1 = __builtin_amdgcn_s_memtime();
      sr_loop0 = __builtin_amdgcn_s_memrealtime();
    }
    auto step = [&](int t) {
      const char* st = lds + (t & 1) * STAGE;
      auto read_act = [&](int s2, int ip) { fact[s2][ip] = *(const bf16x8*)(st + a_rd + ip * 16 * ROW_B + ((s2 << 6) ^ f_x)); };
      auto read_w = [&](int grp) {
        const int s2 = grp / NJ, jn = grp - s2 * NJ;
        fw[grp % RING] = *(const bf16x8*)(st + b_rd + jn * 16 * ROW_B + ((s2 << 6) ^ f_x));
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
          if constexpr (ABLATE == 1) {
            asm volatile("" ::"v"(fw[grp % RING]), "v"(fact[s2][ip]));  // keep the fragment reads alive
          } else {
            acc[jn][ip] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[grp % RING], fact[s2][ip], acc[jn][ip], 0, 0, 0);
          }
          if (ip == 0) {
            if (grp + RING - 1 < NGRP) read_w(grp + RING - 1);
          } else if (ip == 1) {
            if (grp >= 1 && grp <= 4) read_act(1, grp - 1);
          } else if (ip == 2) {
            if (grp >= 1 && grp <= N_DMA) dma_piece(grp - 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      dma_advance();
      __syncthreads();  // copy of step t+1 landed (fence drains the DMA); stage t free to be overwritten
    };

    if constexpr (RESID && RT > 0) {
      // Residual tile hidden inside the K loop.  The x += PFF2(...) GEMM would otherwise read its 256 x BN fp32
      // tile either before the loop or in the epilogue, in both cases with every CU doing it at the same time
      // and no MFMA running (measured: +40 us on a 60 us GEMM).  Here step t issues the loads of a few
      // accumulators' worth of residual (one 16-byte load per accumulator) right after the barrier, and step t+1
      // adds them into the accumulators before its first MFMA: the loads fly under a whole step of MFMAs and are
      // retired by the step's own end-of-step wait.
      constexpr int NACC = 4 * NJ;                          // accumulators (= 16-byte residual pieces) per lane
      constexpr int APG = (NACC + RT - 1) / RT;             // pieces per group (one group per step)
      constexpr int NG = (NACC + APG - 1) / APG;            // group g is loaded during step g-1, added in step g
      static_assert(NG <= RT, "K loop too short to hide the residual tile");
      // wave-uniform base + one 32-bit per-lane offset: the loads then need no per-piece address registers
      // (the launcher only picks this form for resid_mul == 1: residual row == output row)
      const float* rbase = as_global(g.resid) + ((long)b * g.L + l0 + wr * 64) * g.N + n0 + wc * (BN / 2);
      const unsigned rlane = (unsigned)(fr * g.N + 8 * fq);
      f32x4 rv[APG];
      auto load_group = [&](int grp) {
        const float* rb0 = rbase;
        asm volatile("" : "+s"(rb0));  // addresses formed here, step by step, not hoisted out of the unrolled loop
        // (typed as a global pointer; behind the asm the compiler no longer knows the address space)
        const __attribute__((address_space(1))) float* const rb = (const __attribute__((address_space(1))) float*)rb0;
#pragma unroll
        for (int v = 0; v < APG; ++v) {
          const int e = grp * APG + v, jn = e >> 2, ip = e & 3;  // accumulator (jn, ip)
          if (e < NACC)
            rv[v] = *(const __attribute__((address_space(1))) f32x4*)(rb + (long)(ip * 16) * g.N + 32 * (jn >> 1) + 4 * (jn & 1) + rlane);
        }
      };
      load_group(0);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        if (t < NG) {
#pragma unroll
          for (int v = 0; v < APG; ++v) {
            const int e = t * APG + v;
            if (e < NACC) acc[e >> 2][e & 3] += rv[v];
          }
          if (t + 1 < NG) load_group(t + 1);
        }
        step(t);
      }
    } else {
      for (int t = 0; t < T; ++t) step(t);
    }
  }

  unsigned long long sc2 = 0, sr_loop1 = 0;
  if constexpr (ABLATE == 3) {
    sc2 = __builtin_amdgcn_s_memtime();
    sr_loop1 = __builtin_amdgcn_s_memrealtime();
  }
  // ---- epilogue: four consecutive channels of one position per lane and accumulator.  The row base is laundered
  // through an empty asm so that the addresses are computed HERE: hoisted above the K loop as loop invariants (or
  // shared with the residual loads) they cost live registers.
  int l0e = l0;
  asm volatile("" : "+s"(l0e));
  unsigned ln_target = 0;
  if constexpr (LNF) ln_target = ln_publish<NT, WM>(g, acc, lds, tile_m, tile_n, tiles_n, tid);
  // (fused LayerNorm with a residual: ln_finish stores the raw sums itself, behind its wait; g.ln.dbg bit 16 keeps
  // the older order - raw stores here, in front of the wait - for comparisons)
  if (!LNF || (RESID && (g.ln.dbg & 16))) {
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
          float* const outf = as_global(SPLIT ? g.partial + (long)bid_y * g.B * g.L * g.N : (float*)g.out);
          *(f32x4*)(outf + o + 32 * jp) = v;
          *(f32x4*)(outf + o + 32 * jp + 4) = w;
        }
      }
    }
  }
  }
  if constexpr (LNF)
    ln_finish<NT, WM, RESID>(g, acc, lds, ln_target, tile_m, tiles_n, (long)b * g.L + l0e,
                             g.L - l0e < BM ? g.L - l0e : BM, n0, tid);
  if constexpr (ABLATE == 3) {  // diagnostic build (eight values per tile): {prologue, K loop, epilogue incl. store
    // drain} cycles, total ticks, then the absolute 100-MHz times of entry, loop start, loop end and exit
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long sc3 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      g.stamps[8 * tile] = (long long)(sc1 - sc0);
      g.stamps[8 * tile + 1] = (long long)(sc2 - sc1);
      g.stamps[8 * tile + 2] = (long long)(sc3 - sc2);
      g.stamps[8 * tile + 3] = (long long)(sr1 - sr0);
      g.stamps[8 * tile + 4] = (long long)sr0;
      g.stamps[8 * tile + 5] = (long long)sr_loop0;
      g.stamps[8 * tile + 6] = (long long)sr_loop1;
      g.stamps[8 * tile + 7] = (long long)sr1;
    }
  }
}

}  // namespace gemm_tile
}  // namespace ed
