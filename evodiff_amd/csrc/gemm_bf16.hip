// Tuned bf16 implicit-GEMM kernel for the ByteNet block on gfx950 (MI355X).
//
//   out[b*L + l][n] = bias[n] (+ resid) + sum_{tap, k} A[b, l + (tap - ntaps/2)*dilation][k] * W[tap][n][k]
//
// Tiling.  One 512-thread workgroup (8 waves = 4 (M) x 2 (N)) owns a 256 x BN output tile, BN = 320 or 256:
// each wave a 64 x BN/2 sub-tile as 2 x NT MFMA 32x32x16 bf16 tiles (NT = 5 or 4; 160 / 128 fp32
// accumulators per lane).  640M has N = 1280 = 4 x 320 and config 2 has 64 row tiles, i.e. exactly one
// tile per CU of the 256-CU chip with no tail; N = 512 / 1024 (38M) use BN = 256.
//
// K loop.  K runs over (tap, 64-channel slice).  Both operand tiles of a K-step are copied global -> LDS
// by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 tile rows per wave-instruction) into one of two LDS
// stages, the copy of step t+1 in flight while step t is multiplied; one barrier per K-step.
// The LDS image is lane-linear (a DMA cannot scatter), so the bank swizzle
// (16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7)) is applied to the per-lane SOURCE
// address and again on the ds_read_b128 fragment reads: the 16 rows of every ds_read_b128 lane group then
// fall on 16 different 16-byte slots of the 256-byte bank row (conflict-free).
//
// Dilation halo.  A rows of tap t are the tile's rows shifted by (t - ntaps/2)*dilation.  The A buffer of
// the convolution is laid out per sequence with zeroed halo rows in front and behind (a_row0, a_seq_pitch),
// so a shifted tile is a plain contiguous row range and 'same' padding costs no bounds logic.  Work that is
// provably zero is skipped at two levels: taps whose whole 256-row tile lies outside [0, L) are dropped from
// the K loop, and a wave skips a K-step when all 64 of its rows lie outside for the current tap (at L = 256,
// dilation 128 this removes 3 of 5 taps' worth of MFMA work; finer, per-fragment skipping was measured to
// cost more in register pressure than it saves).
#include "kernels.hpp"

#include <type_traits>

namespace ed {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int BM = 256, BK = 64, ROW_B = BK * 2;  // 128 bytes of K per tile row per K-step

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int NT, bool OUT_BF16, bool RESID>
__global__ __launch_bounds__(512) void gemm_bf16_kernel(GemmProblem g, int tiles_n, int tiles_total) {
  constexpr int BN = NT * 64;
  constexpr int A_BYTES = BM * ROW_B, B_BYTES = BN * ROW_B, STAGE = A_BYTES + B_BYTES;
  constexpr int NGB = BN / 64;  // 8-row DMA groups of the B tile per wave
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
  const int it_begin = tap_lo * ksteps, it_end = (tap_hi + 1) * ksteps;

  // ---- DMA source addressing: lane -> (row in 8-row group, physical chunk); logical chunk = physical ^ swizzle
  const int srow = lane >> 3, pch = lane & 7, sw_hi = lane >> 4;
  const long row_bytes = (long)g.K * 2;
  const char* a_src = (const char*)g.A + (((long)b * g.a_seq_pitch + g.a_row0 + l0 + wid * 32 + srow) * g.K) * 2;
  const char* w_src = (const char*)g.W + ((long)(n0 + wid * (NGB * 8) + srow) * g.K) * 2;
  const int a_ch0 = (pch ^ sw_hi) * 16, a_ch1 = (pch ^ (4 + sw_hi)) * 16;  // 8-row group parity 0 / 1

  auto stage = [&](int s, int it) {
    const int tap = it / ksteps;
    const int kk = it - tap * ksteps;
    char* la = lds + s * STAGE;
    char* lb = la + A_BYTES;
    const char* ap = a_src + (long)(tap - half_taps) * g.dilation * row_bytes + kk * ROW_B;
    const char* wp = w_src + (long)tap * g.N * row_bytes + kk * ROW_B;
#pragma unroll
    for (int j = 0; j < 4; ++j)  // A rows wid*32 + 8j .. +8
      glds16(ap + (long)(8 * j) * row_bytes + ((j & 1) ? a_ch1 : a_ch0), la + (wid * 4 + j) * 1024);
#pragma unroll
    for (int j = 0; j < NGB; ++j) {  // W rows n0 + 8*(wid*NGB + j) .. +8
      const int par = (wid * NGB + j) & 1;
      glds16(wp + (long)(8 * j) * row_bytes + (par ? a_ch1 : a_ch0), lb + (wid * NGB + j) * 1024);
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- fragment read addressing (32x32x16: lane = (row fr, k-half fh) holds 8 consecutive k)
  const int fr = lane & 31, fh = lane >> 5, fsw = (lane >> 1) & 7;
  const int a_rd = (wr * 64 + fr) * ROW_B;
  const int b_rd = (wc * (BN / 2) + fr) * ROW_B;

  // One K-step of MFMA work for this wave.  LIVE0/LIVE1: which of the wave's two 32-row fragments have any
  // row inside [0, L) for the current tap (compile-time here, so the body is one straight-line block the
  // scheduler can software-pipeline: fragments of k-slice ks+1 are read while slice ks is multiplied).
  auto kstep = [&](const char* la, const char* lb, auto live0_t, auto live1_t) {
    constexpr bool LIVE0 = decltype(live0_t)::value, LIVE1 = decltype(live1_t)::value;
    bf16x8 fa0[2], fa1[2], fb[2][NT];
    auto load = [&](int ks, int buf) {
      const int ch = ((2 * ks + fh) ^ fsw) * 16;
#pragma unroll
      for (int j = 0; j < NT; ++j) fb[buf][j] = *(const bf16x8*)(lb + b_rd + j * 32 * ROW_B + ch);
      if constexpr (LIVE0) fa0[buf] = *(const bf16x8*)(la + a_rd + ch);
      if constexpr (LIVE1) fa1[buf] = *(const bf16x8*)(la + a_rd + 32 * ROW_B + ch);
    };
    load(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) load(ks + 1, (ks + 1) & 1);
      if constexpr (LIVE0) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[ks & 1], fb[ks & 1][j], acc[0][j], 0, 0, 0);
      }
      if constexpr (LIVE1) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[ks & 1], fb[ks & 1][j], acc[1][j], 0, 0, 0);
      }
    }
    // Pin the interleave for the whole K-step (one scheduling region): the 7 fragment reads of slice 0,
    // then one read of slice ks+1 behind each of the first 7 MFMAs of slice ks, so reads are long complete
    // when their MFMAs come up (left alone, hipcc batches reads, waits for all, then batches MFMAs).
    __builtin_amdgcn_sched_group_barrier(0x100, NT + 2, 0);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
#pragma unroll
      for (int q = 0; q < NT + 2; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NT - 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  if (it_begin < it_end) stage(0, it_begin);
  __syncthreads();

  for (int it = it_begin; it < it_end; ++it) {
    const int cur = (it - it_begin) & 1;
    if (it + 1 < it_end) stage(cur ^ 1, it + 1);
    const int tap = it / ksteps;
    const int shift = l0 + wr * 64 + (tap - half_taps) * g.dilation;  // first A position of this wave's rows
    const bool live = shift + 64 > 0 && shift < g.L;  // any of the wave's 64 rows inside [0, L) for this tap
    const char* la = lds + cur * STAGE;
    const char* lb = la + A_BYTES;
    if (live) kstep(la, lb, T_{}, T_{});
    __syncthreads();  // next stage landed (the barrier's fence drains the DMA), this stage free to overwrite
  }

  // ---- epilogue: C[row][col], col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wc * (BN / 2) + j * 32 + fr;
      const float bias = g.bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int l = l0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (l < g.L) {
          const long o = ((long)b * g.L + l) * g.N + n;
          float v = acc[i][j][r] + bias;
          if constexpr (RESID) v += g.resid[o];
          if constexpr (OUT_BF16) ((bf16_t*)g.out)[o] = f32_to_bf16_bits(v);
          else ((float*)g.out)[o] = v;
        }
      }
    }
}

template <int NT>
hipError_t launch_nt(const GemmProblem& p, hipStream_t stream) {
  const int tiles_n = p.N / (NT * 64);
  const int tiles_total = p.B * ((p.L + BM - 1) / BM) * tiles_n;
  const dim3 grid(tiles_total), block(512);
  if (p.resid) hipLaunchKernelGGL((gemm_bf16_kernel<NT, false, true>), grid, block, 0, stream, p, tiles_n, tiles_total);
  else if (p.out_bf16) hipLaunchKernelGGL((gemm_bf16_kernel<NT, true, false>), grid, block, 0, stream, p, tiles_n, tiles_total);
  else hipLaunchKernelGGL((gemm_bf16_kernel<NT, false, false>), grid, block, 0, stream, p, tiles_n, tiles_total);
  return hipGetLastError();
}

}  // namespace

// Shape contract of the tuned kernel.  The caller guarantees the memory contract: A rows
// [a_row0 - halo, a_row0 + ceil(L/256)*256 + halo) of every sequence are readable, with halo =
// (ntaps/2)*dilation, and rows outside [0, L) hold zeros whenever ntaps > 1.
bool gemm_bf16_supported(const GemmProblem& p) {
  if (!p.a_bf16 || p.K % BK != 0 || p.K < BK) return false;
  if (p.N % 320 != 0 && p.N % 256 != 0) return false;
  if ((long)p.B * p.L < BM) return false;  // less than one 256-row tile: generic kernel
  if (p.ntaps > 1 && p.a_row0 < (long)(p.ntaps / 2) * p.dilation) return false;
  if (p.resid && p.out_bf16) return false;
  return true;
}

hipError_t launch_gemm_bf16(const GemmProblem& p, hipStream_t stream) {
  if (!gemm_bf16_supported(p)) return hipErrorInvalidValue;
  return p.N % 320 == 0 ? launch_nt<5>(p, stream) : launch_nt<4>(p, stream);
}

}  // namespace ed
