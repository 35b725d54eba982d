// Weight-streaming kernels for autoregressive steps on a few rows (ed_ar_sample / ed_ar_sample_scaffold, bf16 path).
//
// An autoregressive step of a causal ByteNet (evodiff/generate.py:238-244 without re-running the prefix) computes ONE new
// row per sequence and layer: every GEMM of the block is a matrix-vector product per sequence, 23 MB of bf16 weights per
// 640M block against a few thousand activations - the purest weight-streaming case of the path (north_star: small
// batches against the HBM peak).  On the tile kernels such a step is six launches per block (three GEMMs on 64 x 64
// tiles with their K loops cut across workgroups + three LayerNorms that add up the partial sums), each a latency chain
// of ~6 us for ~1 us of traffic.  Here a block is three launches and nothing is cut or exchanged:
//   - consumer-side LayerNorm: every workgroup normalises the B input rows itself (B x C values: a few microseconds of
//     one wave per row, redundantly in every workgroup, in place of a launch and a round trip through memory), applies
//     gamma / beta / activation and keeps the rows as bf16 in LDS;
//   - a workgroup owns a few output columns (five at the 640M widths: 1280 columns over 256 CUs) and its eight waves
//     split the K range of those columns between them.  The products run on the matrix pipe: one
//     v_mfma_f32_16x16x32_bf16 takes 32 channels of up to 16 weight columns (row operand, 16-byte pieces straight from
//     memory, all of a wave's pieces in flight together and issued before anything else) against the same channels of up
//     to 16 sequences (column operand, from the LDS rows), so the cost of a launch does not depend on the number of
//     sequences and the sums over K need no exchange between lanes; the eight waves' sums meet in LDS and are added in a
//     fixed order - nothing leaves the workgroup unfinished (one sequence: a wave per column instead, see the kernel);
//   - the convolution reads its older taps from the per-layer cache of normalised rows (api.hip: Car) and workgroup 0
//     appends the new row to it.
// The result of a launch is the raw fp32 GEMM output (bias and residual included); the LayerNorm behind it belongs to
// the next launch.
#include "kernels.hpp"

#include <algorithm>

namespace ed {
namespace {

constexpr float LN_EPS = 1e-5f;  // nn.LayerNorm default (as elementwise.hip)
constexpr int ARG_WAVES = 8, ARG_THREADS = 64 * ARG_WAVES;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// x * sigmoid(x (a + b x^2 + c x^4)): the bf16 path's fit of the exact-erf GELU (elementwise.hip, 2.6e-5 absolute)
__device__ __forceinline__ float gelu_sigmoid_fit(float x) {
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float A = -1.59501577f * LOG2E, B = -7.40112924e-02f * LOG2E, C = 7.03033587e-04f * LOG2E;
  const float xc = __builtin_amdgcn_fmed3f(x, -9.0f, 9.0f);
  const float x2 = xc * xc;
  const float w = xc * fmaf(x2, fmaf(x2, C, B), A);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}

// sum over the 64 lanes, result in every lane (all lanes active): DPP adds inside each 16-lane row and four readlanes
// across them (as elementwise.hip; the generic shuffle is a round trip through the LDS crossbar per step)
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  const int vi = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (r0 + r1) + (r2 + r3);
}

constexpr int ARG_NC = 16;  // output columns of a workgroup and pass, at most (the MFMA's row operand)

// NCH: 512-channel chunks per row (C <= NCH * 512); NTAPS: 1 or 5; BT: LDS rows per tap (B <= BT <= 16)
//
// Order inside a workgroup - every launch is a chain of memory latencies (rows written by the previous launch, cold
// weights), so everything that depends on nothing is issued first and all of it is in flight together:
//   1. the weight pieces of the workgroup's first columns (registers);
//   2. the older taps' cache rows (registers, flattened over the workgroup's threads);
//   3. the raw input rows with gamma / beta, then the LayerNorm of each row by one wave, into LDS;
//   4. barrier; products, the waves' sums through LDS, bias / residual, store.
template <int NCH, int NTAPS, int BT>
__global__ __launch_bounds__(ARG_THREADS) void ar_gemv_kernel(ArGemv p) {
  __shared__ __attribute__((aligned(16))) bf16_t act[NTAPS * BT * NCH * 512];  // [tap][b][NCH * 512]
  __shared__ float red[ARG_WAVES][BT * ARG_NC];                                // [wave][b][column]
  constexpr int RS = NCH * 512;                                                // LDS row stride (elements)
  constexpr int RPW = (BT + ARG_WAVES - 1) / ARG_WAVES;                       // rows a wave normalises
  constexpr int NMT = (NCH * 64 / ARG_WAVES + 3) / 4;                          // MFMAs per tap and wave, at most
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.C, B = p.B, N = p.N;
  const int pos = p.pos.get();
  const int ncol = p.ncol, ngroups = (N + ncol - 1) / ncol;

  // a tap's C / 8 sixteen-byte pieces are split between the waves: wave w takes pieces [w * S, min((w + 1) * S, C / 8)),
  // MFMA m of a tap the pieces 4 m .. 4 m + 3 of them (K = 32 channels), lane l piece 4 m + l / 16 of weight column /
  // sequence l % 16
  const int per_row = C / 8, S = (per_row + ARG_WAVES - 1) / ARG_WAVES;
  const int q_end = min((wave + 1) * S, per_row), q0 = wave * S + (lane >> 4), l16 = lane & 15;
  auto load_group = [&](int g, auto& w) {
    const int n = g * ncol + l16;
    const bool col_on = l16 < ncol && n < N;
    const bf16_t* wcol = p.W + (long)(col_on ? n : 0) * C;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        const int q = q0 + 4 * m;
        w[t][m] = make_uint4(0, 0, 0, 0);
        if (col_on && q < q_end) w[t][m] = *(const uint4*)(wcol + (long)t * N * C + q * 8);
      }
  };
  // One sequence (BT == 1): a WAVE owns whole columns instead - it streams a column's weight rows with all 64 lanes (16
  // bytes each, 1 KB per instruction), multiplies them with the one LDS row (v_dot2c_f32_bf16 on the packed pairs) and
  // sums across its lanes; no second barrier, no exchange between waves.  Measured faster for one sequence (1.14 against
  // 1.33 ms per step at 640M) and slower from two on, where its products and LDS reads grow with the batch.
  constexpr bool COLWAVE = BT == 1;
  const int gw = blockIdx.x * ARG_WAVES + wave, n_waves = gridDim.x * ARG_WAVES;
  auto load_column = [&](int n, uint4 (&w)[NTAPS][NCH]) {
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const bf16_t* wrow = p.W + ((long)t * N + n) * C;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int k = c * 512 + lane * 8;
        w[t][c] = k < C ? *(const uint4*)(wrow + k) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  // ---- 1. first column group of this workgroup (first column of this wave); bias and residual of the value this
  // thread finishes (an element of the residual stream is read and written by the same thread)
  uint4 w0[COLWAVE ? 1 : NTAPS][COLWAVE ? 1 : NMT];
  uint4 c0[COLWAVE ? NTAPS : 1][COLWAVE ? NCH : 1];
  float cadd0 = 0.f;
  if constexpr (COLWAVE) {
    if (gw < N) {
      load_column(gw, c0);
      cadd0 = p.bias[gw] + (p.resid ? p.resid[gw] : 0.f);
    }
  } else {
    load_group(blockIdx.x, w0);
  }
  auto load_tail = [&](int g, float& add) {
    add = 0.f;
    if (tid < BT * ARG_NC) {
      const int b = tid / ARG_NC, j = tid - b * ARG_NC, n = g * ncol + j;
      if (b < B && j < ncol && n < N) add = p.bias[n] + (p.resid ? p.resid[(long)b * N + n] : 0.f);
    }
  };
  float add0 = 0.f;
  if constexpr (!COLWAVE) load_tail(blockIdx.x, add0);

  // ---- 2. older taps of the convolution: rows pos - (NTAPS-1-t) * dilation of the cache (real zero rows in front of
  // every sequence stand for the positions before its start), 16 bytes per thread and piece, all pieces in flight
  if constexpr (NTAPS > 1) {
    const int pieces = (NTAPS - 1) * B * per_row;
    constexpr int MAXP = ((NTAPS - 1) * BT * NCH * 64 + ARG_THREADS - 1) / ARG_THREADS;  // pieces per thread, at most
    uint4 piece[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int q = i * ARG_THREADS + tid;
      piece[i] = make_uint4(0, 0, 0, 0);
      if (q < pieces) {
        const int r = q / per_row, k = (q - r * per_row) * 8, t = r / B, b = r - t * B;
        piece[i] = *(const uint4*)(p.cache + ((long)b * p.cache_pitch + p.cache_row0 + pos - (long)(NTAPS - 1 - t) * p.dilation) * C + k);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      const int q = i * ARG_THREADS + tid;
      if (q < pieces) {
        const int r = q / per_row, k = (q - r * per_row) * 8, t = r / B, b = r - t * B;
        *(uint4*)(act + (t * BT + b) * RS + k) = piece[i];
      }
    }
  }

  // ---- 3. the new rows: LayerNorm + activation of the raw input, one wave per sequence (a row is C <= 2048 fp32
  // values: up to 8 chunks of 256, 4 per lane); mean, then the centred second moment, on registers
  {
    float v[RPW][2 * NCH][4];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int b = wave + ARG_WAVES * i;
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) {
        const int idx = (c * 64 + lane) * 4;
        float4 raw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B && idx < C) raw = *(const float4*)(p.in + (long)b * p.in_pitch + idx);
        v[i][c][0] = raw.x; v[i][c][1] = raw.y; v[i][c][2] = raw.z; v[i][c][3] = raw.w;
      }
    }
    // gamma / beta ahead of the statistics (their latency then hides under the row's), registers permitting
    constexpr bool GB_EARLY = NTAPS * NMT * 4 + RPW * 8 * NCH + 16 * NCH <= 208;
    float4 gm_r[GB_EARLY ? 2 * NCH : 1], bt_r[GB_EARLY ? 2 * NCH : 1];
    if constexpr (GB_EARLY) {
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) {
        const int idx = (c * 64 + lane) * 4;
        gm_r[c] = idx < C ? *(const float4*)(p.gamma + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
        bt_r[c] = idx < C ? *(const float4*)(p.beta + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int b = wave + ARG_WAVES * i;
      if (b >= B) continue;  // (wave-uniform)
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) s += (v[i][c][0] + v[i][c][1]) + (v[i][c][2] + v[i][c][3]);
      const float mean = wave_sum(s) / (float)C;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) {
        const bool in_row = (c * 64 + lane) * 4 < C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = in_row ? v[i][c][j] - mean : 0.f;
          v[i][c][j] = d;
          q = fmaf(d, d, q);
        }
      }
      const float rstd = rsqrtf(wave_sum(q) / (float)C + LN_EPS);
      bf16_t* dst = act + ((NTAPS - 1) * BT + b) * RS;
      bf16_t* crow = p.cache ? p.cache + ((long)b * p.cache_pitch + p.cache_row0 + pos) * C : nullptr;
#pragma unroll
      for (int c = 0; c < 2 * NCH; ++c) {
        const int idx = (c * 64 + lane) * 4;
        if (idx >= C) continue;
        float4 gm, bt;
        if constexpr (GB_EARLY) {
          gm = gm_r[c];
          bt = bt_r[c];
        } else {
          gm = *(const float4*)(p.gamma + idx);
          bt = *(const float4*)(p.beta + idx);
        }
        float y[4] = {fmaf(v[i][c][0] * rstd, gm.x, bt.x), fmaf(v[i][c][1] * rstd, gm.y, bt.y),
                      fmaf(v[i][c][2] * rstd, gm.z, bt.z), fmaf(v[i][c][3] * rstd, gm.w, bt.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = p.act == 1 ? gelu_sigmoid_fit(y[j]) : (p.act == 2 ? fmaxf(y[j], 0.f) : y[j]);
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        const bf16x4 pk = {(__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};  // v_cvt_pk_bf16_f32 (RNE)
        *(bf16x4*)(dst + idx) = pk;
        // the convolution's new cache row (one writer: nobody in this launch reads it from memory, tap NTAPS-1 is in LDS)
        if (crow != nullptr && blockIdx.x == 0) *(bf16x4*)(crow + idx) = pk;
      }
    }
  }
  __syncthreads();

  // ---- 4. column groups: workgroup g owns groups g, g + (workgroups of the launch), ...
  auto group = [&](int g, const auto& w, float add) {
    f32x4 acc[4];  // independent chains, added up at the end
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        if (wave * S + 4 * m >= q_end) continue;  // (wave-uniform)
        const int q = q0 + 4 * m;
        // pieces past the wave's range are zero on BOTH sides (0 x whatever the LDS holds there could be a NaN); lanes
        // of sequences the batch does not have feed output columns nobody reads
        uint4 a = make_uint4(0, 0, 0, 0);
        if (q < q_end && l16 < BT) a = *(const uint4*)(act + (t * BT + l16) * RS + q * 8);
        acc[(t * NMT + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8, w[t][m]), __builtin_bit_cast(bf16x8, a), acc[(t * NMT + m) & 3], 0, 0, 0);
      }
    const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // lane l holds columns (l / 16) * 4 + r of sequence l % 16
    if (l16 < BT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][l16 * ARG_NC + (lane >> 4) * 4 + r] = sum[r];
    }
    __syncthreads();
    if (tid < BT * ARG_NC) {
      const int b = tid / ARG_NC, j = tid - b * ARG_NC, n = g * ncol + j;
      if (b < B && j < ncol && n < N) {
        float r = red[0][tid];
#pragma unroll
        for (int wv = 1; wv < ARG_WAVES; ++wv) r += red[wv][tid];  // fixed order
        p.out[(long)b * N + n] = r + add;
      }
    }
  };
  if constexpr (COLWAVE) {
    auto column = [&](int n, const uint4 (&w)[NTAPS][NCH], float add) {
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int k = c * 512 + lane * 8;
          if (k >= C) continue;
          const uint4 a = *(const uint4*)(act + t * RS + k);
          acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[t][c].x), __builtin_bit_cast(bf16x2, a.x), acc, false);
          acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[t][c].y), __builtin_bit_cast(bf16x2, a.y), acc, false);
          acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[t][c].z), __builtin_bit_cast(bf16x2, a.z), acc, false);
          acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[t][c].w), __builtin_bit_cast(bf16x2, a.w), acc, false);
        }
      const float r = wave_sum(acc);
      if (lane == 0) p.out[n] = r + add;
    };
    if (gw < N) column(gw, c0, cadd0);
    for (int n = gw + n_waves; n < N; n += n_waves) {
      uint4 w[NTAPS][NCH];
      load_column(n, w);
      column(n, w, p.bias[n] + (p.resid ? p.resid[n] : 0.f));
    }
  } else {
    if ((int)blockIdx.x < ngroups) group(blockIdx.x, w0, add0);
    for (int g = blockIdx.x + gridDim.x; g < ngroups; g += gridDim.x) {
      __syncthreads();  // the sums of the pass before have been read
      uint4 w[NTAPS][NMT];
      float add;
      load_group(g, w);
      load_tail(g, add);
      group(g, w, add);
    }
  }
}

template <int NCH, int NTAPS>
hipError_t launch_bt(const ArGemv& p, int grid, hipStream_t stream) {
  if (p.B <= 1) hipLaunchKernelGGL((ar_gemv_kernel<NCH, NTAPS, 1>), dim3(grid), dim3(ARG_THREADS), 0, stream, p);
  else if (p.B <= 2) hipLaunchKernelGGL((ar_gemv_kernel<NCH, NTAPS, 2>), dim3(grid), dim3(ARG_THREADS), 0, stream, p);
  else if (p.B <= 4) hipLaunchKernelGGL((ar_gemv_kernel<NCH, NTAPS, 4>), dim3(grid), dim3(ARG_THREADS), 0, stream, p);
  else if constexpr (NTAPS * 8 * NCH * 512 * 2 + ARG_WAVES * 8 * ARG_NC * 4 <= 160 * 1024)  // (the LDS rows of 8 sequences must fit)
    hipLaunchKernelGGL((ar_gemv_kernel<NCH, NTAPS, 8>), dim3(grid), dim3(ARG_THREADS), 0, stream, p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <int NTAPS>
hipError_t launch_nch(const ArGemv& p, int grid, hipStream_t stream) {
  switch ((p.C + 511) / 512) {
    case 1: return launch_bt<1, NTAPS>(p, grid, stream);
    case 2: return launch_bt<2, NTAPS>(p, grid, stream);
    case 3: return launch_bt<3, NTAPS>(p, grid, stream);
    case 4: return launch_bt<4, NTAPS>(p, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool ar_gemv_supported(const ArGemv& p) {
  if (p.B < 1 || p.B > AR_GEMV_MAX_B || p.N < 1) return false;
  if (p.C < 8 || p.C > 2048 || p.C % 8 != 0) return false;
  if (p.ntaps != 1 && p.ntaps != 5) return false;
  if (p.ntaps > 1 && (!p.cache || p.dilation < 1 || p.cache_row0 < (long)(p.ntaps - 1) * p.dilation)) return false;
  if (p.act < 0 || p.act > 2) return false;
  if (p.B > 4 && (long)p.ntaps * 8 * ((p.C + 511) / 512) * 512 * 2 + ARG_WAVES * 8 * ARG_NC * 4 > 160 * 1024) return false;  // LDS rows of 8 sequences
  return p.in && p.gamma && p.beta && p.W && p.bias && p.out;
}

hipError_t launch_ar_gemv(const ArGemv& problem, int n_cus, hipStream_t stream) {
  if (!ar_gemv_supported(problem)) return hipErrorInvalidValue;
  ArGemv p = problem;
  // one workgroup per CU at most (the convolution's LDS rows allow no more); a workgroup takes as many columns at a time
  // as spread the launch over every CU (1280 columns: 5 x 256), further groups in passes
  const int cus = n_cus > 0 ? n_cus : 256;
  p.ncol = std::max(1, std::min(ARG_NC, (p.N + cus - 1) / cus));
  const int grid = p.B == 1 ? std::min(cus, (p.N + ARG_WAVES - 1) / ARG_WAVES)   // (one sequence: a wave per column)
                            : std::min(cus, (p.N + p.ncol - 1) / p.ncol);
  return p.ntaps == 1 ? launch_nch<1>(p, grid, stream) : launch_nch<5>(p, grid, stream);
}

}  // namespace ed
