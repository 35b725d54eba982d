// Tuned bf16 implicit-GEMM kernel for the ByteNet block on gfx950 (MI355X).
//
//   out[b*L + l][n] = bias[n] (+ resid) + sum_{tap, k} A[b, l + (tap - ntaps/2)*dilation][k] * W[tap][n][k]
//
// Tiling.  One 512-thread workgroup (8 waves = 4 (M) x 2 (N)) owns a 256 x BN output tile, BN = 320 or 256:
// each wave a 64 x BN/2 sub-tile as 2 x NT MFMA 32x32x16 bf16 tiles (NT = 5 or 4; 160 / 128 fp32
// accumulators per lane).  640M has N = 1280 = 4 x 320 and config 2 has 64 row tiles, i.e. exactly one
// tile per CU of the 256-CU chip with no tail; N = 512 / 1024 (38M) use BN = 256.
//
// K loop.  K runs over (tap, 64-channel slice) "steps".  The operand tiles of a step (256 + BN rows of 128
// bytes = whole cache lines) are copied global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 tile
// rows per wave-instruction) into one of two LDS stages, the copy of step t+1 in flight while step t is
// multiplied; one barrier per step.  The issue order inside a step is pinned (sched_barrier after every
// MFMA): behind each of the 40 (32) MFMAs of a step comes exactly one auxiliary instruction - one of the
// 7 (6) fragment reads of the NEXT k-slice or one of the step's 9 (8) DMA copies - so neither the LDS read
// latency nor the ~50-100-cycle issue cost of a copy ever opens an MFMA-free stretch in a wave.  (Left to
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
#include "kernels.hpp"

namespace ed {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int BM = 256, BK = 64, ROW_B = BK * 2;  // 128 bytes of K per tile row per step

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ABLATE (diagnostic builds, results invalid): 1 = no MFMAs, 2 = no DMA copies.
// RT > 0 (1x1 + residual only): the K loop has exactly RT steps and is fully unrolled, and the residual tile is
// streamed in DURING the loop (see below) instead of before it.
template <int NT, bool CONV, bool OUT_BF16, bool RESID, int ABLATE = 0, int RT = 0>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(GemmProblem g, int tiles_n, int tiles_total) {
  constexpr int BN = NT * 64;
  constexpr int A_BYTES = BM * ROW_B, B_BYTES = BN * ROW_B, STAGE = A_BYTES + B_BYTES;
  constexpr int NGB = BN / 64;      // 8-row DMA groups of the W tile per wave (A: 4 per wave)
  constexpr int N_DMA = 4 + NGB;    // copies per wave per step
  constexpr int N_RD = NT + 2;      // fragment reads per k-slice
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;

  // XCD-aware tile order: workgroups id, id+8, ... share an XCD (and its L2); give each XCD a compact
  // block of the tile grid so that the tiles resident on it share A row panels and W column panels.
  int tile;
  {
    const int id = blockIdx.x, q = tiles_total >> 3, r = tiles_total & 7, x = id & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
  }
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  const int tiles_per_seq = (g.L + BM - 1) / BM;
  const int b = tile_m / tiles_per_seq;
  const int l0 = (tile_m - b * tiles_per_seq) * BM;
  const int n0 = tile_n * BN;

  const int ksteps = g.K / BK, half_taps = g.ntaps / 2;
  int tap_lo = 0, tap_hi = g.ntaps - 1;
  while (tap_lo < g.ntaps && l0 + (tap_lo - half_taps) * g.dilation + BM <= 0) ++tap_lo;
  while (tap_hi >= 0 && l0 + (tap_hi - half_taps) * g.dilation >= g.L) --tap_hi;
  const int T = (tap_hi - tap_lo + 1) * ksteps;  // number of steps

  // ---- DMA addressing.  Lane -> (row in 8-row group, physical chunk); logical chunk = physical ^ swizzle.
  // Group bases are multiples of 8 rows, so (row >> 1) & 7 = 4 * (group parity) + (lane >> 4).
  const long row_bytes = (long)g.K * 2;
  const unsigned lane_row = (unsigned)((lane >> 3) * row_bytes);
  const unsigned lane_off0 = lane_row + (((lane & 7) ^ (lane >> 4)) << 4);
  const unsigned lane_off1 = lane_row + (((lane & 7) ^ (4 + (lane >> 4))) << 4);
  const unsigned grp_bytes = (unsigned)(8 * row_bytes);
  // wave-uniform bases of the copy stream (this wave's first A / W group), advanced step by step
  const char* a_dma = (const char*)g.A +
                      ((long)b * g.a_seq_pitch + g.a_row0 + l0 + (long)(tap_lo - half_taps) * g.dilation + wid * 32) *
                          row_bytes;
  const char* w_dma = (const char*)g.W + ((long)tap_lo * g.N + n0 + wid * (NGB * 8)) * row_bytes;
  const long a_tap_step = (long)(g.dilation - 1) * row_bytes + ROW_B, w_tap_step = (long)(g.N - 1) * row_bytes + ROW_B;
  int dma_kk = 0, dma_t = 0;

  // piece j (0 .. N_DMA-1) of the copy of step dma_t into stage dma_t & 1
  auto dma_piece = [&](int j) {
    if constexpr (ABLATE == 2) return;
    char* st = lds + (dma_t & 1) * STAGE;
    if (j < 4) {  // A rows wid*32 + 8j .. +8   (group wid*4 + j, parity j & 1)
      glds16(a_dma + (unsigned)j * grp_bytes + ((j & 1) ? lane_off1 : lane_off0), st + (wid * 4 + j) * 1024);
    } else {      // W rows n0 + 8*(wid*NGB + jj) .. +8
      const int jj = j - 4, grp = wid * NGB + jj;
      glds16(w_dma + (unsigned)jj * grp_bytes + ((grp & 1) ? lane_off1 : lane_off0), st + A_BYTES + grp * 1024);
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

  // Accumulators start at bias (+ residual): the epilogue is then a pure store, and the residual tile is
  // fetched here, where its latency hides behind the first DMA wait, instead of serialising the store tail
  // (measured: the add-in-the-epilogue form made the x += PFF2(...) GEMM 2x slower than the same GEMM without it).
  const int fr = lane & 31, fh = lane >> 5;
  const bool full_rows = l0 + BM <= g.L;  // no row masking needed (always true when 256 | L)
  const int rsel = RESID ? g.resid_sel.get() : 0;  // residual row of output row R: R * resid_mul + rsel
  f32x16 acc[2][NT];
  {
    const int nb = n0 + wc * (BN / 2) + fr;
    float bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = g.bias[nb + j * 32];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int l = l0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        const bool on = full_rows || l < g.L;
        const float* pr = RESID ? g.resid + (((long)b * g.L + l) * g.resid_mul + rsel) * g.N + nb : nullptr;  // one address per row
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float v = bias[j];
          if constexpr (RESID && RT == 0)
            if (on) v += pr[j * 32];
          acc[i][j][r] = v;
        }
      }
  }

  // ---- fragment addressing (32x32x16: lane = (row fr, k-half fh) holds 8 consecutive k)
  const int fsw = (lane >> 1) & 7;
  const int a_rd = (wr * 64 + fr) * ROW_B;
  const int b_rd = A_BYTES + (wc * (BN / 2) + fr) * ROW_B;

  bf16x8 fa[2][2], fb[2][NT];  // [k-slice parity][fragment]
  // read r (0 .. N_RD-1) of k-slice ks: A rows 0-31 first (the first MFMAs need it), then W, A rows 32-63 last
  auto read_piece = [&](const char* st, int ks, int r) {
    const char* base = st + (((2 * ks + fh) ^ fsw) << 4);
    if (r == 0) fa[ks & 1][0] = *(const bf16x8*)(base + a_rd);
    else if (r <= NT) fb[ks & 1][r - 1] = *(const bf16x8*)(base + b_rd + (r - 1) * 32 * ROW_B);
    else fa[ks & 1][1] = *(const bf16x8*)(base + a_rd + 32 * ROW_B);
  };
  auto mfma_piece = [&](int ks, int q) {  // MFMA q (0 .. 2*NT-1) of k-slice ks
    const int i = q / NT, j = q - i * NT;
    if constexpr (ABLATE == 1) {
      asm volatile("" ::"v"(fa[ks & 1][i]), "v"(fb[ks & 1][j]));  // keep the fragment reads alive
    } else {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
    }
  };

  if (T > 0) {
    // prologue: copy of step 0
#pragma unroll
    for (int j = 0; j < N_DMA; ++j) dma_piece(j);
    dma_advance();
    __syncthreads();  // (its fence drains the DMA)

    // One step: 4 k-slices x 2*NT MFMAs; behind each MFMA one auxiliary instruction: first the reads of the
    // next slice, then (slots left over) the copies of step t+1 into the other stage.
    auto step = [&](int t) {
      const char* st = lds + (t & 1) * STAGE;
#pragma unroll
      for (int r = 0; r < N_RD; ++r) read_piece(st, 0, r);
      __builtin_amdgcn_sched_barrier(0);
      int dma_j = 0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int q = 0; q < 2 * NT; ++q) {
          mfma_piece(ks, q);
          if (ks < 3 && q < N_RD) {
            read_piece(st, ks + 1, q);
          } else if (dma_j < N_DMA) {
            dma_piece(dma_j);
            ++dma_j;
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
      // and no MFMA running (measured: +40 us on a 60 us GEMM).  Here step t issues the loads of CPS accumulator
      // tiles' worth of residual (16 dwords per lane each) right after the barrier, and step t+1 adds them into
      // the accumulators before its first MFMA: the loads fly under a whole step of MFMAs and are retired by
      // the step's own end-of-step wait.
      constexpr int NVAL = 32 * NT;                       // residual values per lane
      constexpr int VPG = ((NVAL + RT - 1) / RT + 7) & ~7;  // values per group (one group per step), 8 or 16
      constexpr int NG = (NVAL + VPG - 1) / VPG;          // groups: group g is loaded during step g-1, added in step g
      static_assert(NG <= RT, "K loop too short to hide the residual tile");
      // wave-uniform base + one 32-bit per-lane offset: the loads then need no per-element address registers
      // (the launcher only picks this form for resid_mul == 1: residual row == output row)
      const float* rbase = g.resid + ((long)b * g.L + l0 + wr * 64) * g.N + n0 + wc * (BN / 2);
      const unsigned rlane = (unsigned)(4 * fh * g.N + fr);
      float rv[VPG];
      auto load_group = [&](int grp) {
#pragma unroll
        for (int v = 0; v < VPG; ++v) {
          const int e = grp * VPG + v, c = e >> 4, r = e & 15;  // accumulator tile c = (i, j), register r
          if (e < NVAL) rv[v] = (rbase + (long)((c / NT) * 32 + (r & 3) + 8 * (r >> 2)) * g.N + (c % NT) * 32)[rlane];
        }
      };
      load_group(0);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        if (t < NG) {
#pragma unroll
          for (int v = 0; v < VPG; ++v) {
            const int e = t * VPG + v, c = e >> 4, r = e & 15;
            if (e < NVAL) acc[c / NT][c % NT][r] += rv[v];
          }
          if (t + 1 < NG) load_group(t + 1);
        }
        step(t);
      }
    } else {
      for (int t = 0; t < T; ++t) step(t);
    }
  }

  // ---- epilogue: C[row][col], col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // The row base is laundered through an empty asm so that the 32 row addresses are computed HERE: hoisted
  // above the K loop as loop invariants (or shared with the residual loads) they cost 40-64 live registers.
  int l0e = l0;
  asm volatile("" : "+s"(l0e));
  {
    const int nb = n0 + wc * (BN / 2) + fr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int l = l0e + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (full_rows || l < g.L) {
          const long o = ((long)b * g.L + l) * g.N + nb;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if constexpr (OUT_BF16) ((bf16_t*)g.out)[o + j * 32] = f32_to_bf16_bits(acc[i][j][r]);
            else ((float*)g.out)[o + j * 32] = acc[i][j][r];
          }
        }
      }
  }
}

template <int NT, bool CONV>
hipError_t launch_nt(const GemmProblem& p, hipStream_t stream) {
  const int tiles_n = p.N / (NT * 64);
  const int tiles_total = p.B * ((p.L + BM - 1) / BM) * tiles_n;
  const dim3 grid(tiles_total), block(512);
#define ED_GO(OB, RS, AB) \
  hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, OB, RS, AB>), grid, block, 0, stream, p, tiles_n, tiles_total)
  if (p.ablate == 1) ED_GO(false, false, 1);
  else if (p.ablate == 2) ED_GO(false, false, 2);
  else if (p.resid) {
    // hidden-residual form: 1x1, every tile full, K loop of a known length
    const bool whole = !CONV && ((long)p.B * p.L) % BM == 0 && (p.B == 1 || p.L % BM == 0) && p.resid_mul == 1;
    if (whole && p.K == 20 * BK)
      hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 0, 20>), grid, block, 0, stream, p, tiles_n, tiles_total);
    else if (whole && NT == 4 && p.K == 8 * BK)
      hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 0, 8>), grid, block, 0, stream, p, tiles_n, tiles_total);
    else ED_GO(false, true, 0);
  }
  else if (p.out_bf16) ED_GO(true, false, 0);
  else ED_GO(false, false, 0);
#undef ED_GO
  return hipGetLastError();
}

}  // namespace

// Shape contract of the tuned kernel.  The caller guarantees the memory contract: A rows
// [a_row0 - halo, a_row0 + ceil(L/256)*256 + halo) of every sequence are readable, with halo =
// (ntaps/2)*dilation, and rows outside [0, L) hold zeros whenever ntaps > 1.
bool gemm_bf16_supported(const GemmProblem& p) {
  if (!p.a_bf16 || p.K % BK != 0 || p.K < BK) return false;
  if (p.N % 320 != 0 && p.N % 256 != 0) return false;
  if ((long)p.B * p.L < BM) return false;               // less than one 256-row tile: generic kernel
  if ((long)p.K * 2 * 64 >= (1L << 31)) return false;   // per-lane DMA offsets are 32-bit
  if (p.ntaps > 1 && p.a_row0 < (long)(p.ntaps / 2) * p.dilation) return false;
  if (p.resid && p.out_bf16) return false;
  return true;
}

hipError_t launch_gemm_bf16(const GemmProblem& p, hipStream_t stream) {
  if (!gemm_bf16_supported(p)) return hipErrorInvalidValue;
  if (p.ntaps > 1) return p.N % 320 == 0 ? launch_nt<5, true>(p, stream) : launch_nt<4, true>(p, stream);
  return p.N % 320 == 0 ? launch_nt<5, false>(p, stream) : launch_nt<4, false>(p, stream);
}

}  // namespace ed
