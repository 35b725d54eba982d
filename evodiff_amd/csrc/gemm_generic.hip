// Generic implicit-GEMM tile kernel for the ByteNet block (gfx950).
//
// One kernel serves the 1x1 projections (ntaps = 1) and the dilated k=5 MaskedConv1d
// (sequence_models MaskedConv1d as called from evodiff/model.py:113-117): the convolution is a GEMM
// whose K axis runs over (tap, channel) and whose A rows for tap t are the activation rows shifted by
// (t - ntaps/2) * dilation positions inside the SAME sequence; rows that fall outside [0, L) are
// zero ('same' padding) and taps whose whole 128-row tile is outside are skipped.
//
// This is the correctness-first kernel: 128x128 output tile per 256-thread workgroup (4 waves, each
// 64x64 as 2x2 MFMA 32x32 tiles), register-staged global->LDS copies (four K-steps ahead, two LDS stages, one barrier
// per step) with bounds checks, so
// it takes any B, L, N and any K that is a multiple of 64 bytes.  It is the fp32 validation path
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains) and the fallback for bf16 shapes the tuned kernel
// in gemm_bf16.hip does not cover.
#include "kernels.hpp"

namespace ed {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

// Tile: 64 TM rows x 64 TM columns per 256-thread workgroup, TM = 2 (128 x 128) or 1 (64 x 64: four times as many
// workgroups for problems that would leave most CUs without a 128 x 128 tile, instead of cutting their K loops as far).
constexpr int ROW_BYTES = 128;   // bytes of K per tile row per K-step (32 fp32 / 64 bf16)
constexpr int LDS_STRIDE = 144;  // 128 + 16 pad: the 16 rows of a ds_read_b128 lane group hit 16 distinct slots

template <bool BF16, bool OUT_BF16, bool RESID, int TM>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmProblem g) {
  constexpr int BM = 64 * TM, BN = 64 * TM;
  constexpr int WT = 32 * TM;               // rows / columns of a wave's piece (TM x TM MFMA 32 x 32 tiles)
  constexpr int NV = 2 * TM;                // 16-byte pieces per loader thread, operand and K-step
  constexpr int TPR = 4 / TM;               // loader threads per tile row
  constexpr int ESZ = BF16 ? 2 : 4;
  constexpr int KSTEP = ROW_BYTES / ESZ;
  constexpr int TILE = BM * LDS_STRIDE, STAGE = 2 * TILE;  // A tile + W tile of one K-step
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];  // two stages: step it+1 is written while step it is read

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  // diagnostics (GemmProblem::stamps): 100-MHz ticks at entry, after the first K-step's operands are in LDS, after the
  // K loop and at the end, per workgroup
  long long* const stamp = g.stamps ? g.stamps + 4 * ((long)blockIdx.x + gridDim.x * ((long)blockIdx.y + gridDim.y * (long)blockIdx.z)) : nullptr;
  if (stamp && tid == 0) stamp[0] = (long long)__builtin_amdgcn_s_memrealtime();
  const int tiles_per_seq = (g.L + BM - 1) / BM;
  const int b = blockIdx.y / tiles_per_seq;
  const int l0 = (blockIdx.y % tiles_per_seq) * BM;
  const int n0 = blockIdx.x * BN;

  // loader role: thread -> (tile row, NV consecutive 16-byte pieces of the 128-byte K slice)
  const int lrow = tid / TPR, lbyte = (tid % TPR) * (NV * 16), lhalf = lbyte >> 6;
  const int ksteps = (g.K + KSTEP - 1) / KSTEP;   // the last slice may be half full (K is a multiple of 64 bytes)
  const int half_taps = g.center();
  const bool last_half_empty = (g.K % KSTEP) != 0;

  // contiguous range of taps that touch this tile at all
  int tap_lo = 0, tap_hi = g.ntaps - 1;
  if (!g.a_unbounded) {
    while (tap_lo < g.ntaps && l0 + (tap_lo - half_taps) * g.dilation + BM <= 0) ++tap_lo;
    while (tap_hi >= 0 && l0 + (tap_hi - half_taps) * g.dilation >= g.L) --tap_hi;
  }
  // row groups: this loader thread's A row without the tap shift, and whether it exists at all
  long a_row_grouped = 0;
  if (g.seq_rows > 0) {
    const unsigned R = (unsigned)(l0 + lrow), bs = R / (unsigned)g.seq_rows, ls = R - bs * (unsigned)g.seq_rows;
    a_row_grouped = (long)bs * g.a_seq_pitch + g.a_row0 + g.a_sel.get() + ls;
  }
  int it_begin = tap_lo * ksteps, it_end = (tap_hi + 1) * ksteps;
  if (g.ksplit > 1) {  // this workgroup's share of the K loop
    const int chunk = (it_end - it_begin + g.ksplit - 1) / g.ksplit;
    it_begin += (int)blockIdx.z * chunk;
    it_end = it_begin + chunk < it_end ? it_begin + chunk : it_end;
  }

  const char* Abase = (const char*)g.A;
  const char* Wbase = (const char*)g.W;
  const int nrow = n0 + lrow;

  // PF register sets: the loads of K-step it+PF are issued as soon as the set of step it has been written to LDS, so
  // every load has PF full steps to return (small problems run few workgroups per CU, stream cold weights from HBM
  // and are latency-bound otherwise).
  // gload() is called for it_begin, it_begin + 1, ... in this order, so the (tap, slice) pair and the two row pointers
  // are carried along instead of being recomputed (a division and 64-bit products per call made the prologue of a
  // small problem as long as its K loop).
  constexpr int PF = 4;
  static_assert(PF % 2 == 0, "the LDS stage of a step is the parity of its register set");
  uint4 ra[PF][NV], rb[PF][NV];
  int g_tap = it_begin / ksteps, g_ks = it_begin - g_tap * ksteps;
  const char *a_tap = nullptr, *w_tap = nullptr;  // this thread's A / W row of the current tap, at its 64-byte half
  bool row_on = false;
  auto set_tap = [&](int tap) __attribute__((always_inline)) {
    const int shift = (tap - half_taps) * g.dilation;
    const int l = l0 + lrow + shift;
    row_on = g.seq_rows > 0 ? (l0 + lrow < g.L && (g.a_unbounded || (l >= 0 && l < g.L))) : (l >= 0 && l < g.L);
    const long a_row = g.seq_rows > 0 ? a_row_grouped + shift : (long)b * g.a_seq_pitch + g.a_row0 + l;
    a_tap = Abase + a_row * g.K * ESZ + lbyte;
    w_tap = Wbase + ((long)tap * g.N + nrow) * g.K * ESZ + lbyte;
  };
  set_tap(g_tap);
  auto gload = [&](uint4 (&a4)[NV], uint4 (&b4)[NV]) __attribute__((always_inline)) {
    const uint4 z = make_uint4(0, 0, 0, 0);
    const bool k_on = !(last_half_empty && g_ks == ksteps - 1 && lhalf == 1);
    const int koff = g_ks * ROW_BYTES;
    if (k_on && row_on) {
      const uint4* p = (const uint4*)(a_tap + koff);
#pragma unroll
      for (int i = 0; i < NV; ++i) a4[i] = p[i];
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) a4[i] = z;
    }
    if (k_on && nrow < g.N) {
      const uint4* p = (const uint4*)(w_tap + koff);
#pragma unroll
      for (int i = 0; i < NV; ++i) b4[i] = p[i];
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) b4[i] = z;
    }
    if (++g_ks == ksteps) {
      g_ks = 0;
      set_tap(++g_tap);
    }
  };
  auto to_lds = [&](char* stage, const uint4 (&a4)[NV], const uint4 (&b4)[NV]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      *(uint4*)(stage + lrow * LDS_STRIDE + lbyte + i * 16) = a4[i];
      *(uint4*)(stage + TILE + lrow * LDS_STRIDE + lbyte + i * 16) = b4[i];
    }
  };

  f32x16 acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fh = lane >> 5;
  // Step `it` of register set u: its operands already sit in stage u & 1.  The fragments are read, the next step's
  // registers go to the other stage and are refilled from memory, then the MFMAs run: one barrier per step.
  auto step = [&](int it, int u, uint4 (&an)[NV], uint4 (&bn)[NV]) __attribute__((always_inline)) {
    const char* cur = lds + (u & 1) * STAGE;
    char* nxt = lds + ((u + 1) & 1) * STAGE;
    // Each lane takes 32 contiguous bytes of its row at byte 64*h + 32*fh of the slice.  The k order seen by the
    // MFMA is a permutation of the true k, identical for A and B, so the sum over k is unchanged.
    uint4 fa[TM][2], fb[TM][2];
    auto read_frags = [&](int h) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* pa = cur + (wr * WT + i * 32 + fr) * LDS_STRIDE + h * 64 + fh * 32;
        fa[i][0] = *(const uint4*)pa;
        fa[i][1] = *(const uint4*)(pa + 16);
        const char* pb = cur + TILE + (wc * WT + i * 32 + fr) * LDS_STRIDE + h * 64 + fh * 32;
        fb[i][0] = *(const uint4*)pb;
        fb[i][1] = *(const uint4*)(pb + 16);
      }
    };
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          if constexpr (BF16) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, fa[i][s]), __builtin_bit_cast(bf16x8, fb[j][s]), acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              const float* a4f = (const float*)&fa[i][s];
              const float* b4f = (const float*)&fb[j][s];
#pragma unroll
              for (int q = 0; q < 4; ++q)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4f[q], b4f[q], acc[i][j], 0, 0, 0);
            }
          }
        }
    };
    read_frags(0);
    if (it + 1 < it_end) {
      to_lds(nxt, an, bn);
      if (it + 1 + PF < it_end) gload(an, bn);
    }
    multiply();
    read_frags(1);
    multiply();
    __syncthreads();
  };

#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (it_begin + u < it_end) gload(ra[u], rb[u]);
  if (it_begin < it_end) {
    to_lds(lds, ra[0], rb[0]);
    if (it_begin + PF < it_end) gload(ra[0], rb[0]);
  }
  __syncthreads();
  if (stamp && tid == 0) stamp[1] = (long long)__builtin_amdgcn_s_memrealtime();
  for (int it = it_begin; it < it_end; it += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (it + u < it_end) step(it + u, u, ra[(u + 1) % PF], rb[(u + 1) % PF]);
  }

  if (stamp && tid == 0) stamp[2] = (long long)__builtin_amdgcn_s_memrealtime();
  // Element (i, j, r) of this lane sits at tile row 32 i + (r & 3) + 8 (r >> 2) + 4 fh, tile column 32 j + fr: the lane
  // part of the offset is computed once, the rest are compile-time multiples of N (32-bit: a tile spans 128 rows).
  const int row_w = l0 + wr * WT, col_w = n0 + wc * WT;            // first row / column of this wave's piece
  const int lane_off = 4 * fh * g.N + fr;
  const int row_lim = g.L - row_w - 4 * fh;                         // rows 32 i + c(r) below this exist
  if (g.ksplit > 1) {  // raw partial sums; bias, residual and rounding happen in the reduction
    float* part = g.partial + ((long)blockIdx.z * g.B * g.L + (long)b * g.L + row_w) * g.N + col_w;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if (col_w + j * 32 + fr >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tr = i * 32 + (r & 3) + 8 * (r >> 2);
          if (tr < row_lim) part[lane_off + tr * g.N + j * 32] = acc[i][j][r];
        }
      }
    if (stamp) {
      __builtin_amdgcn_s_waitcnt(0);
      if (tid == 0) stamp[3] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    return;
  }
  const int rsel = RESID ? g.resid_sel.get() : 0;
  if (g.seq_rows == 0) {
    // epilogue: C[row][col] with col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const long orow = (long)b * g.L + row_w;
    char* const obase = (char*)g.out + (orow * g.N + col_w) * (OUT_BF16 ? 2 : 4);
    const float* const rbase = RESID ? g.resid + ((orow * g.resid_mul + rsel) * g.N + col_w) : nullptr;
    const int lane_roff = 4 * fh * g.resid_mul * g.N + fr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        if (col_w + j * 32 + fr >= g.N) continue;
        const float bias = g.bias[col_w + j * 32 + fr];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int tr = i * 32 + (r & 3) + 8 * (r >> 2);
          if (tr >= row_lim) continue;
          const int o = lane_off + tr * g.N + j * 32;
          float v = acc[i][j][r] + bias;
          if constexpr (RESID) v += rbase[lane_roff + tr * g.resid_mul * g.N + j * 32];
          if constexpr (OUT_BF16) ((bf16_t*)obase)[o] = f32_to_bf16_bits(v);
          else ((float*)obase)[o] = v;
        }
      }
    return;
  }
  // row groups: output and residual rows addressed per sequence
  const int osel = g.out_sel.get();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int n = n0 + wc * WT + j * 32 + fr;
      if (n >= g.N) continue;
      const float bias = g.bias[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int l = l0 + wr * WT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
        if (l >= g.L) continue;
        const unsigned bs = (unsigned)l / (unsigned)g.seq_rows, ls = (unsigned)l - bs * (unsigned)g.seq_rows;
        const long o = ((long)bs * g.out_pitch + osel + ls) * g.N + n;
        const long ro = ((long)bs * g.resid_pitch + (long)ls * g.resid_mul + rsel) * g.N + n;
        float v = acc[i][j][r] + bias;
        if constexpr (RESID) v += g.resid[ro];
        if constexpr (OUT_BF16) ((bf16_t*)g.out)[o] = f32_to_bf16_bits(v);
        else ((float*)g.out)[o] = v;
      }
    }
}

// out[r][n] = bias[n] (+ resid[r * resid_mul + sel][n]) + partial[0][r][n] + partial[1][r][n] + ...  (fixed order)
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void reduce_partials_kernel(GemmProblem g) {
  const long rows = (long)g.B * g.L, quads = rows * g.N / 4;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const long e = q * 4, r = e / g.N;
  const int n = (int)(e - r * g.N);
  long orow = r, rrow = r * g.resid_mul + g.resid_sel.get();
  if (g.seq_rows > 0) {
    const long bs = r / g.seq_rows, ls = r - bs * g.seq_rows;
    orow = bs * g.out_pitch + g.out_sel.get() + ls;
    rrow = bs * g.resid_pitch + ls * g.resid_mul + g.resid_sel.get();
  }
  float4 s = *(const float4*)(g.bias + n);
  if (g.resid) {
    const float4 x = *(const float4*)(g.resid + rrow * g.N + n);
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < g.ksplit; ++z) {
    const float4 p = *(const float4*)(g.partial + z * rows * g.N + e);
    t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
  }
  s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  if constexpr (OUT_BF16) {
    uint2 pk;
    pk.x = (unsigned)f32_to_bf16_bits(s.x) | ((unsigned)f32_to_bf16_bits(s.y) << 16);
    pk.y = (unsigned)f32_to_bf16_bits(s.z) | ((unsigned)f32_to_bf16_bits(s.w) << 16);
    *(uint2*)((bf16_t*)g.out + orow * g.N + n) = pk;
  } else {
    *(float4*)((float*)g.out + orow * g.N + n) = s;
  }
}

}  // namespace

hipError_t launch_reduce_partials(const GemmProblem& p, hipStream_t stream) {
  if (p.ksplit < 2 || !p.partial || p.N % 4 != 0) return hipErrorInvalidValue;
  const long quads = (long)p.B * p.L * p.N / 4;
  const dim3 grid((unsigned)((quads + 255) / 256)), block(256);
  if (p.out_bf16) hipLaunchKernelGGL(reduce_partials_kernel<true>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(reduce_partials_kernel<false>, grid, block, 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_gemm_generic(const GemmProblem& p, hipStream_t stream) {
  const int kstep = p.a_bf16 ? 32 : 16;  // 64 bytes: the unit the kernel masks the K tail in
  if (p.K % kstep != 0 || p.B <= 0 || p.L <= 0 || p.N <= 0 || (p.ntaps & 1) == 0) return hipErrorInvalidValue;
  if (p.resid && p.out_bf16) return hipErrorInvalidValue;
  if (p.ksplit < 1 || (p.ksplit > 1 && !p.partial)) return hipErrorInvalidValue;
  if (p.seq_rows > 0 && (p.B != 1 || p.L % p.seq_rows != 0)) return hipErrorInvalidValue;
  if (p.tile != 128 && p.tile != 64) return hipErrorInvalidValue;
  const int T = p.tile;
  dim3 grid((p.N + T - 1) / T, (unsigned)(p.B * ((p.L + T - 1) / T)), (unsigned)p.ksplit);
  dim3 block(256);
#define ED_LAUNCH(BF, OB, RS)                                                                        \
  do {                                                                                               \
    if (T == 128) hipLaunchKernelGGL((gemm_generic_kernel<BF, OB, RS, 2>), grid, block, 0, stream, p); \
    else hipLaunchKernelGGL((gemm_generic_kernel<BF, OB, RS, 1>), grid, block, 0, stream, p);         \
  } while (0)
  if (p.a_bf16) {
    if (p.resid) ED_LAUNCH(true, false, true);
    else if (p.out_bf16) ED_LAUNCH(true, true, false);
    else ED_LAUNCH(true, false, false);
  } else {
    if (p.resid) ED_LAUNCH(false, false, true);
    else if (p.out_bf16) ED_LAUNCH(false, true, false);
    else ED_LAUNCH(false, false, false);
  }
#undef ED_LAUNCH
  return hipGetLastError();
}

}  // namespace ed
