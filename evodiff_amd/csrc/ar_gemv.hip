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
//   - a wave owns whole output columns: it streams the column's weight rows (16 bytes per lane and load, all loads of a
//     column in flight together), multiplies them with the LDS rows of every sequence and reduces across its lanes -
//     fp32 accumulation, no split-K, no partial sums;
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

// x * sigmoid(x (a + b x^2 + c x^4)): the bf16 path's fit of the exact-erf GELU (elementwise.hip, 2.6e-5 absolute)
__device__ __forceinline__ float gelu_sigmoid_fit(float x) {
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float A = -1.59501577f * LOG2E, B = -7.40112924e-02f * LOG2E, C = 7.03033587e-04f * LOG2E;
  const float xc = __builtin_amdgcn_fmed3f(x, -9.0f, 9.0f);
  const float x2 = xc * xc;
  const float w = xc * fmaf(x2, fmaf(x2, C, B), A);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// NCH: 512-channel chunks per row (C <= NCH * 512); NTAPS: 1 or 5; BT: sequences (B <= BT)
//
// Order inside a workgroup - every launch is a chain of memory latencies (rows written by the previous launch, cold
// weights), so everything that depends on nothing is issued first and all of it is in flight together:
//   1. the weights and the bias of each wave's first column (registers);
//   2. the older taps' cache rows (registers, flattened over the workgroup's threads);
//   3. the raw input rows with gamma / beta, then the LayerNorm of each row by one wave, into LDS;
//   4. barrier; columns.
template <int NCH, int NTAPS, int BT>
__global__ __launch_bounds__(ARG_THREADS) void ar_gemv_kernel(ArGemv p) {
  __shared__ __attribute__((aligned(16))) bf16_t act[NTAPS * BT * NCH * 512];  // [tap][b][NCH * 512]
  constexpr int RS = NCH * 512;                                                // LDS row stride (elements)
  constexpr int RPW = (BT + ARG_WAVES - 1) / ARG_WAVES;                       // rows a wave normalises
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.C, B = p.B, N = p.N;
  const int pos = p.pos.get();
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
  // ---- 1. first column of this wave (gw < N always holds for the grid the launcher picks, except on tiny N)
  uint4 w0[NTAPS][NCH];
  float bias0 = 0.f;
  if (gw < N) {
    load_column(gw, w0);
    bias0 = p.bias[gw];
  }

  // ---- 2. older taps of the convolution: rows pos - (NTAPS-1-t) * dilation of the cache (real zero rows in front of
  // every sequence stand for the positions before its start), 16 bytes per thread and piece, all pieces in flight
  if constexpr (NTAPS > 1) {
    const int per_row = C / 8, pieces = (NTAPS - 1) * B * per_row;
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
    float4 gm[2 * NCH], bt[2 * NCH];
#pragma unroll
    for (int c = 0; c < 2 * NCH; ++c) {
      const int idx = (c * 64 + lane) * 4;
      gm[c] = idx < C ? *(const float4*)(p.gamma + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
      bt[c] = idx < C ? *(const float4*)(p.beta + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        const bool on = (c * 64 + lane) * 4 < C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = on ? v[i][c][j] - mean : 0.f;
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
        float y[4] = {fmaf(v[i][c][0] * rstd, gm[c].x, bt[c].x), fmaf(v[i][c][1] * rstd, gm[c].y, bt[c].y),
                      fmaf(v[i][c][2] * rstd, gm[c].z, bt[c].z), fmaf(v[i][c][3] * rstd, gm[c].w, bt[c].w)};
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

  // ---- 4. columns: wave gw owns columns gw, gw + (waves of the launch), ...
  auto column = [&](int n, const uint4 (&w)[NTAPS][NCH], float bias) {
    float acc[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = 0.f;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int k = c * 512 + lane * 8;
        if (k >= C) continue;
        // v_dot2c_f32_bf16: two bf16 products (exact in fp32) added to the fp32 sum per instruction, on the packed pairs
        // as they come from memory and from LDS
        const bf16x2 wp[4] = {__builtin_bit_cast(bf16x2, w[t][c].x), __builtin_bit_cast(bf16x2, w[t][c].y),
                              __builtin_bit_cast(bf16x2, w[t][c].z), __builtin_bit_cast(bf16x2, w[t][c].w)};
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          if (b >= B) continue;
          const uint4 a = *(const uint4*)(act + (t * BT + b) * RS + k);
          acc[b] = __builtin_amdgcn_fdot2_f32_bf16(wp[0], __builtin_bit_cast(bf16x2, a.x), acc[b], false);
          acc[b] = __builtin_amdgcn_fdot2_f32_bf16(wp[1], __builtin_bit_cast(bf16x2, a.y), acc[b], false);
          acc[b] = __builtin_amdgcn_fdot2_f32_bf16(wp[2], __builtin_bit_cast(bf16x2, a.z), acc[b], false);
          acc[b] = __builtin_amdgcn_fdot2_f32_bf16(wp[3], __builtin_bit_cast(bf16x2, a.w), acc[b], false);
        }
      }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b >= B) continue;
      const float r = wave_sum(acc[b]);
      if (lane == 0) {
        const long o = (long)b * N + n;
        p.out[o] = r + bias + (p.resid ? p.resid[o] : 0.f);
      }
    }
  };
  if (gw < N) column(gw, w0, bias0);
  for (int n = gw + n_waves; n < N; n += n_waves) {
    uint4 w[NTAPS][NCH];
    load_column(n, w);
    column(n, w, p.bias[n]);
  }
}

template <int NCH, int NTAPS>
hipError_t launch_bt(const ArGemv& p, int grid, hipStream_t stream) {
  if (p.B <= 1) hipLaunchKernelGGL((ar_gemv_kernel<NCH, NTAPS, 1>), dim3(grid), dim3(ARG_THREADS), 0, stream, p);
  else if (p.B <= 4) hipLaunchKernelGGL((ar_gemv_kernel<NCH, NTAPS, 4>), dim3(grid), dim3(ARG_THREADS), 0, stream, p);
  else if constexpr (NTAPS * 8 * NCH * 512 * 2 <= 160 * 1024)  // (the LDS rows of 8 sequences must fit)
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
  if (p.B > 4 && (long)p.ntaps * 8 * ((p.C + 511) / 512) * 512 * 2 > 160 * 1024) return false;  // LDS rows of 8 sequences
  return p.in && p.gamma && p.beta && p.W && p.bias && p.out;
}

hipError_t launch_ar_gemv(const ArGemv& p, int n_cus, hipStream_t stream) {
  if (!ar_gemv_supported(p)) return hipErrorInvalidValue;
  // one workgroup per CU at most (the convolution's LDS rows allow no more), never more waves than columns
  const int grid = (int)std::max(1, std::min(n_cus > 0 ? n_cus : 256, (p.N + ARG_WAVES - 1) / ARG_WAVES));
  return p.ntaps == 1 ? launch_nch<1>(p, grid, stream) : launch_nch<5>(p, grid, stream);
}

}  // namespace ed
