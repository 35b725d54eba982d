// Row-wise kernels of the ByteNetLMTime step (gfx950): embedding + timestep encoding + up-projection,
// LayerNorm + activation, and the three step tails (full logits, OA-DM draw, D3PM posterior draw).
// All are HBM-bound: one 64-lane wave owns one row, loads are 16 bytes (fp32) / 8 bytes (bf16) per lane,
// statistics are reduced with cross-lane shuffles, nothing is staged through LDS except small tables.
#include "kernels.hpp"

#include <type_traits>

namespace ed {
namespace {

constexpr float LN_EPS = 1e-5f;  // nn.LayerNorm default, as constructed in ByteNetBlock / model.py:166

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float bf16_bits_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// exact-erf GELU to ~2e-7 absolute (Abramowitz-Stegun 7.1.26 + hardware exp/rcp): a third of erff's
// instructions.  Used where the result is rounded to bf16 (2^-9 relative) right after.
__device__ __forceinline__ float gelu_bf16_grade(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = p * t * __expf(-z * z);   // 1 - erf(z)
  const float erf_abs = 1.0f - e;
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// ---------------------------------------------------------------------------------------------------
// embed: evodiff/model.py:131-140
// ---------------------------------------------------------------------------------------------------
// One thread per channel keeps its row of the up-projection (E <= 16 weights) in registers and walks ROWS_PER_BLOCK
// rows: the weights are read once per block instead of once per row, the stores are coalesced along the channels.
constexpr int EMBED_ROWS_PER_BLOCK = 32;
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tokens, const int64_t* tvec,
                                                    const StepState* state, const float* __restrict__ emb,
                                                    const float* __restrict__ pe, int n_timesteps,
                                                    const float* __restrict__ up_w, const float* __restrict__ up_b,
                                                    float* __restrict__ X, int rows, int B, int D, int E,
                                                    int n_tokens) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  const bool c_on = c < D;
  float w[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) w[j] = (c_on && j < E) ? up_w[c * E + j] : 0.f;
  const float bias = c_on ? up_b[c] : 0.f;
  const int row_end = min(rows, (int)(blockIdx.x + 1) * EMBED_ROWS_PER_BLOCK);
  for (int row = blockIdx.x * EMBED_ROWS_PER_BLOCK; row < row_end; ++row) {
    long tok = tokens[row];
    if (tok < 0 || tok >= n_tokens) tok = n_tokens - 1;  // host validates; never index out of the table
    long t = 0;
    if (pe != nullptr) {
      // quirk Q1 (model.py:136-137): expand(L,B,E).reshape(B,L,E) => row r receives pe[y[r % B]]
      t = tvec ? tvec[row % B] : (long)(state->t_start - state->step);
      if (t < 0) t = 0;
      if (t >= n_timesteps) t = n_timesteps - 1;
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < E) {
        float e = emb[tok * E + j];               // wave-uniform addresses: broadcast loads
        if (pe != nullptr) e = pe[t * E + j] + e;
        acc = fmaf(w[j], e, acc);
      }
    if (c_on) X[(long)row * D + c] = acc + bias;
  }
}

// Models without timestep encoding: the embedded row only depends on the token, so the n_tokens possible rows are
// computed once (by embed_kernel itself, at commit) and a step just copies them: one wave per row.
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ tokens,
                                                           const float* __restrict__ table, float* __restrict__ X,
                                                           int rows, int D, int n_tokens) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  long tok = tokens[row];
  if (tok < 0 || tok >= n_tokens) tok = n_tokens - 1;  // host validates; never index out of the table
  const float4* src = (const float4*)(table + tok * D);
  float4* dst = (float4*)(X + row * D);
  for (int c = lane; c < D / 4; c += 64) dst[c] = src[c];
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm (+ activation), one wave per row, NCH chunks of 256 channels held in registers.
// ---------------------------------------------------------------------------------------------------
template <int NCH, bool IN_BF16>
__device__ __forceinline__ void load_row(const void* in, long row, int C, int lane, float (&v)[NCH][4]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = (c * 64 + lane) * 4;
    if (idx < C) {
      if constexpr (IN_BF16) {
        const uint2 raw = *(const uint2*)((const bf16_t*)in + row * C + idx);
        v[c][0] = __uint_as_float(raw.x << 16);
        v[c][1] = __uint_as_float(raw.x & 0xffff0000u);
        v[c][2] = __uint_as_float(raw.y << 16);
        v[c][3] = __uint_as_float(raw.y & 0xffff0000u);
      } else {
        const float4 raw = *(const float4*)((const float*)in + row * C + idx);
        v[c][0] = raw.x; v[c][1] = raw.y; v[c][2] = raw.z; v[c][3] = raw.w;
      }
    } else {
      v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f;
    }
  }
}

// Fast-path statistics: sum and sum of squares in ONE cross-lane reduction round (two independent shuffle
// chains instead of two dependent ones); var = E[x^2] - mean^2 in fp32 is exact enough ahead of a bf16 store.
template <int NCH>
__device__ __forceinline__ void normalize_row_onepass(float (&v)[NCH][4], int C, int lane) {
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s += v[c][j];
      q = fmaf(v[c][j], v[c][j], q);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  const float inv = 1.0f / (float)C;
  const float mean = s * inv;
  const float rstd = rsqrtf(fmaxf(fmaf(-mean, mean, q * inv), 0.f) + LN_EPS);
  const float shift = -mean * rstd;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[c][j] = fmaf(v[c][j], rstd, shift);
}

// (v - mean) * rstd in place; mean and biased variance over C as nn.LayerNorm computes them.
template <int NCH>
__device__ __forceinline__ void normalize_row(float (&v)[NCH][4], int C, int lane) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const bool on = (c * 64 + lane) * 4 < C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = on ? v[c][j] - mean : 0.f;
      v[c][j] = d;
      q = fmaf(d, d, q);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + LN_EPS);
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[c][j] *= rstd;
}

template <int NCH, bool IN_BF16, bool OUT_BF16>
__global__ __launch_bounds__(256) void ln_act_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int rows, int C, LnRowMap map, int act) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long seq = row / map.rows_per_seq, m = row - seq * map.rows_per_seq;
  const long irow = map.in_row0 + seq * map.in_pitch + m * map.in_mul + map.in_sel.get();
  float v[NCH][4];
  load_row<NCH, IN_BF16>(in, irow, C, lane, v);
  if constexpr (OUT_BF16) normalize_row_onepass<NCH>(v, C, lane);
  else normalize_row<NCH>(v, C, lane);
  const long orow = map.out_row0 + seq * map.out_pitch + m + map.out_sel.get();
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = (c * 64 + lane) * 4;
    if (idx >= C) continue;
    const float4 gm = *(const float4*)(gamma + idx);
    const float4 bt = *(const float4*)(beta + idx);
    float y[4] = {fmaf(v[c][0], gm.x, bt.x), fmaf(v[c][1], gm.y, bt.y), fmaf(v[c][2], gm.z, bt.z),
                  fmaf(v[c][3], gm.w, bt.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (act == 1) y[j] = OUT_BF16 ? gelu_bf16_grade(y[j]) : gelu_exact(y[j]);
      else if (act == 2) y[j] = fmaxf(y[j], 0.f);
    }
    if constexpr (OUT_BF16) {
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      const bf16x4 pk = {(__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};  // v_cvt_pk_bf16_f32 (RNE)
      *(bf16x4*)((bf16_t*)out + orow * C + idx) = pk;
    } else {
      *(float4*)((float*)out + orow * C + idx) = make_float4(y[0], y[1], y[2], y[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Fast-path LayerNorm + activation (bf16 out).  Three LayerNorms per block make this kernel a fifth of a step;
// plain streaming kernels move its bytes in 13 us (bf16 in) / 18 us (fp32 in) at config-2 size, so everything
// besides the traffic has to hide under that:
//  - a wave walks RPW consecutive rows and issues the loads of the next row before it works on the current one;
//  - gamma / beta stay in registers across the rows of a wave;
//  - the two row sums go through DPP adds inside each 16-lane row and four readlanes across them (the generic
//    shuffle goes through the LDS crossbar: 12 dependent round trips per row);
//  - GELU as x * sigmoid(x * (a + b x^2 + c x^4)), fitted to the exact-erf GELU to 2.6e-5 absolute on the whole
//    axis (|x| clamped to 9 inside the polynomial, where the sigmoid is saturated): 7 plain + 2 quarter-rate
//    instructions per element, and bf16 rounding (2^-9 relative) of the result is two orders coarser.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_sigmoid_fit(float x) {
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float A = -1.59501577f * LOG2E, B = -7.40112924e-02f * LOG2E, C = 7.03033587e-04f * LOG2E;
  const float xc = __builtin_amdgcn_fmed3f(x, -9.0f, 9.0f);
  const float x2 = xc * xc;
  const float w = xc * fmaf(x2, fmaf(x2, C, B), A);        // -x (a + b x^2 + c x^4) log2(e)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
}

// two elements at a time: hipcc selects the packed fp32 instructions (v_pk_mul/fma/add_f32) for <2 x float>
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_sigmoid_fit2(f32x2 x) {
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float A = -1.59501577f * LOG2E, B = -7.40112924e-02f * LOG2E, C = 7.03033587e-04f * LOG2E;
  f32x2 xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -9.0f, 9.0f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -9.0f, 9.0f);
  const f32x2 x2 = xc * xc;
  const f32x2 w = xc * __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, (f32x2)(C), (f32x2)(B)), (f32x2)(A));
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(w.x);
  e.y = __builtin_amdgcn_exp2f(w.y);
  const f32x2 d = e + 1.0f;
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return x * r;
}

// sum over the 64 lanes, result in every lane (all lanes active)
__device__ __forceinline__ float wave_sum_dpp(float v) {
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

// Row layout of a wave: NC8 chunks of 512 channels at 8 per lane (16-byte bf16 / 2 x 16-byte fp32 accesses), then
// NC4 chunks of 256 channels at 4 per lane for the rest (masked past C).  1280 = 2 x 512 + 256.
template <int NC8, int NC4, bool IN_BF16, int ACT>
__global__ __launch_bounds__(256) void ln_act_fast_kernel(const void* __restrict__ in, bf16_t* __restrict__ out,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int rows, int C,
                                                          LnRowMap map) {
  constexpr int NV = NC8 * 8 + NC4 * 4;  // values per lane
  const int lane = threadIdx.x & 63;
  const int row = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (row >= rows) return;
  const unsigned rps = (unsigned)map.rows_per_seq;
  const unsigned seq = (unsigned)row / rps, m = (unsigned)row - seq * rps;
  const long irow = map.in_row0 + (long)seq * map.in_pitch + (long)m * map.in_mul + map.in_sel.get();
  const long orow = map.out_row0 + (long)seq * map.out_pitch + m + map.out_sel.get();
  // channel index of value slot k (a multiple of 4 slots shares one 16-byte piece of gamma / beta)
  auto chan = [&](int k) { return k < NC8 * 8 ? ((k >> 3) * 64 + lane) * 8 + (k & 7) : NC8 * 512 + (((k - NC8 * 8) >> 2) * 64 + lane) * 4 + (k & 3); };

  float v[NV];
  if constexpr (IN_BF16) {
    const bf16_t* src = (const bf16_t*)in + irow * C;
#pragma unroll
    for (int c = 0; c < NC8; ++c) {
      const uint4 raw = *(const uint4*)(src + chan(c * 8));
      const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[c * 8 + 2 * j] = __uint_as_float(w[j] << 16);
        v[c * 8 + 2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int c = 0; c < NC4; ++c) {
      const int k = NC8 * 8 + c * 4;
      uint2 raw = make_uint2(0, 0);
      if (chan(k) < C) raw = *(const uint2*)(src + chan(k));
      v[k] = __uint_as_float(raw.x << 16);
      v[k + 1] = __uint_as_float(raw.x & 0xffff0000u);
      v[k + 2] = __uint_as_float(raw.y << 16);
      v[k + 3] = __uint_as_float(raw.y & 0xffff0000u);
    }
  } else {
    const float* src = (const float*)in + irow * C;
#pragma unroll
    for (int k = 0; k < NV; k += 4) {
      float4 raw = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < NC8 * 8 || chan(k) < C) raw = *(const float4*)(src + chan(k));
      v[k] = raw.x; v[k + 1] = raw.y; v[k + 2] = raw.z; v[k + 3] = raw.w;
    }
  }
  f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NV; k += 2) {  // masked slots hold zeros
    const f32x2 v2 = {v[k], v[k + 1]};
    s2 += v2;
    q2 = __builtin_elementwise_fma(v2, v2, q2);
  }
  const float s = wave_sum_dpp(s2.x + s2.y);
  const float q = wave_sum_dpp(q2.x + q2.y);
  const float inv = 1.0f / (float)C;
  const float mean = s * inv;
  const float rstd = rsqrtf(fmaxf(fmaf(-mean, mean, q * inv), 0.f) + LN_EPS);
  const float shift = -mean * rstd;
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
  bf16_t* dst = out + orow * C;
#pragma unroll
  for (int k = 0; k < NV; k += 4) {
    const bool on = k < NC8 * 8 || chan(k) < C;
    if (on) {
      const float4 gm = *(const float4*)(gamma + chan(k));
      const float4 bt = *(const float4*)(beta + chan(k));
      const f32x2 g2[2] = {{gm.x, gm.y}, {gm.z, gm.w}}, b2[2] = {{bt.x, bt.y}, {bt.z, bt.w}};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x2 v2 = {v[k + 2 * j], v[k + 2 * j + 1]};
        f32x2 y = __builtin_elementwise_fma(__builtin_elementwise_fma(v2, (f32x2)(rstd), (f32x2)(shift)), g2[j], b2[j]);
        if constexpr (ACT == 1) y = gelu_sigmoid_fit2(y);
        else if constexpr (ACT == 2) y = __builtin_elementwise_max(y, (f32x2)(0.f));
        v[k + 2 * j] = y.x;
        v[k + 2 * j + 1] = y.y;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NC8; ++c) {
    const bf16x8v pk = {(__bf16)v[c * 8], (__bf16)v[c * 8 + 1], (__bf16)v[c * 8 + 2], (__bf16)v[c * 8 + 3],
                        (__bf16)v[c * 8 + 4], (__bf16)v[c * 8 + 5], (__bf16)v[c * 8 + 6], (__bf16)v[c * 8 + 7]};
    *(bf16x8v*)(dst + chan(c * 8)) = pk;  // v_cvt_pk_bf16_f32 (RNE)
  }
#pragma unroll
  for (int c = 0; c < NC4; ++c) {
    const int k = NC8 * 8 + c * 4;
    if (chan(k) < C) {
      const bf16x4 pk = {(__bf16)v[k], (__bf16)v[k + 1], (__bf16)v[k + 2], (__bf16)v[k + 3]};
      *(bf16x4*)(dst + chan(k)) = pk;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Split-K reduction + LayerNorm + activation in one launch (small problems of the fast path).  A GEMM whose K loop
// was cut across workgroups leaves ksplit raw partial sums per output element; the LayerNorm that consumes its result
// adds them up itself (in the fixed order z = 0, 1, ..., plus the bias, rounded to bf16 when the GEMM's result would
// have been stored as bf16), so the reduction launch and the round trip of the result through memory disappear.
// One WORKGROUP per row, one 16-byte piece per thread: the ksplit loads of a thread are independent and in flight
// together (a wave-per-row form of this kernel reads ksplit x 5 KB serially per wave and lost to the two launches).
// With a residual (PFF2 of a block followed by the first LayerNorm of the next block) the reduced fp32 row is also
// stored: it is the residual stream.
// ---------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(1024) void reduce_ln_kernel(const float* __restrict__ partial, long pstride, int ksplit,
                                                         const float* __restrict__ bias, int round_bf16,
                                                         const float* resid, int resid_mul, PosBit resid_sel,
                                                         float* xout, bf16_t* __restrict__ out,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int C, LnRowMap map) {
  __shared__ float red[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = (int)blockDim.x >> 6;
  const unsigned row = blockIdx.x, rps = (unsigned)map.rows_per_seq;
  const unsigned seq = row / rps, m = row - seq * rps;
  const long irow = map.in_row0 + (long)seq * map.in_pitch + (long)m * map.in_mul + map.in_sel.get();
  const long orow = map.out_row0 + (long)seq * map.out_pitch + m + map.out_sel.get();
  const int idx = tid * 4;
  const bool on = idx < C;
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on) {
    const float* p = partial + irow * C + idx;
#pragma unroll 8
    for (int z = 0; z < ksplit; ++z) {
      const float4 x = *(const float4*)(p + z * pstride);
      t.x += x.x; t.y += x.y; t.z += x.z; t.w += x.w;
    }
    float4 b = *(const float4*)(bias + idx);
    if (resid) {  // PFF2: the reduced value is the new residual stream (fp32, stored) AND this LayerNorm's input
      const float4 x = *(const float4*)(resid + (irow * resid_mul + resid_sel.get()) * C + idx);
      b.x += x.x; b.y += x.y; b.z += x.z; b.w += x.w;
    }
    t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
    if (xout) *(float4*)(xout + irow * C + idx) = t;
    if (round_bf16) {
      t.x = bf16_bits_to_f32(f32_to_bf16_bits(t.x)); t.y = bf16_bits_to_f32(f32_to_bf16_bits(t.y));
      t.z = bf16_bits_to_f32(f32_to_bf16_bits(t.z)); t.w = bf16_bits_to_f32(f32_to_bf16_bits(t.w));
    }
  }
  float s = (t.x + t.y) + (t.z + t.w);
  float q = fmaf(t.x, t.x, fmaf(t.y, t.y, fmaf(t.z, t.z, t.w * t.w)));
  s = wave_sum_dpp(s);
  q = wave_sum_dpp(q);
  if (lane == 0) {
    red[0][wid] = s;
    red[1][wid] = q;
  }
  __syncthreads();
  s = 0.f;
  q = 0.f;
  for (int w = 0; w < nw; ++w) {
    s += red[0][w];
    q += red[1][w];
  }
  if (!on) return;
  const float inv = 1.0f / (float)C;
  const float mean = s * inv;
  const float rstd = rsqrtf(fmaxf(fmaf(-mean, mean, q * inv), 0.f) + LN_EPS);
  const float shift = -mean * rstd;
  const float4 gm = *(const float4*)(gamma + idx);
  const float4 bt = *(const float4*)(beta + idx);
  f32x2 y0 = {fmaf(fmaf(t.x, rstd, shift), gm.x, bt.x), fmaf(fmaf(t.y, rstd, shift), gm.y, bt.y)};
  f32x2 y1 = {fmaf(fmaf(t.z, rstd, shift), gm.z, bt.z), fmaf(fmaf(t.w, rstd, shift), gm.w, bt.w)};
  if constexpr (ACT == 1) {
    y0 = gelu_sigmoid_fit2(y0);
    y1 = gelu_sigmoid_fit2(y1);
  } else if constexpr (ACT == 2) {
    y0 = __builtin_elementwise_max(y0, (f32x2)(0.f));
    y1 = __builtin_elementwise_max(y1, (f32x2)(0.f));
  }
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const bf16x4 pk = {(__bf16)y0.x, (__bf16)y0.y, (__bf16)y1.x, (__bf16)y1.y};
  *(bf16x4*)(out + orow * C + idx) = pk;
}

// last_norm + decoder for one row held by one wave: returns logit k in lane k (k < n_out <= 64).
template <int NCH>
__device__ __forceinline__ float row_logits(const float* X, long row, int D, int lane, const float* gamma,
                                            const float* beta, const float* dec_w, const float* dec_b, int n_out) {
  float v[NCH][4];
  load_row<NCH, false>(X, row, D, lane, v);
  normalize_row<NCH>(v, D, lane);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = (c * 64 + lane) * 4;
    if (idx < D) {
      const float4 gm = *(const float4*)(gamma + idx);
      const float4 bt = *(const float4*)(beta + idx);
      v[c][0] = fmaf(v[c][0], gm.x, bt.x); v[c][1] = fmaf(v[c][1], gm.y, bt.y);
      v[c][2] = fmaf(v[c][2], gm.z, bt.z); v[c][3] = fmaf(v[c][3], gm.w, bt.w);
    }
  }
  float mine = 0.f;
  // four outputs at a time: their weight loads are independent, so one memory round trip serves four dot products
  for (int k0 = 0; k0 < n_out; k0 += 4) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u < n_out ? k0 + u : n_out - 1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int idx = (c * 64 + lane) * 4;
        if (idx < D) {
          const float4 w = *(const float4*)(dec_w + (long)k * D + idx);
          s[u] = fmaf(v[c][0], w.x, s[u]); s[u] = fmaf(v[c][1], w.y, s[u]);
          s[u] = fmaf(v[c][2], w.z, s[u]); s[u] = fmaf(v[c][3], w.w, s[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float t = wave_sum_dpp(s[u]);
      if (k0 + u < n_out && lane == k0 + u) mine = t + dec_b[k0 + u];
    }
  }
  return mine;
}

template <int NCH>
__global__ __launch_bounds__(256) void final_logits_kernel(const float* __restrict__ X, const float* gamma,
                                                           const float* beta, const float* dec_w, const float* dec_b,
                                                           float* __restrict__ logits, int rows, int D, int n_tokens) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float mine = row_logits<NCH>(X, row, D, lane, gamma, beta, dec_w, dec_b, n_tokens);
  if (lane < n_tokens) logits[row * n_tokens + lane] = mine;
}

// softmax over lanes [0, n): returns p in lane k, computed as exp(x - max) / sum like torch.softmax.
__device__ __forceinline__ float lane_softmax(float logit, int lane, int n) {
  const float x = lane < n ? logit : -INFINITY;
  const float m = wave_max(x);
  const float e = lane < n ? expf(x - m) : 0.f;
  return e / wave_sum(e);
}

// ---------------------------------------------------------------------------------------------------
// Scoring tail: analysis/sequence_perp.py:127-154.  last_norm + decoder + cross entropy of the first n_classes logits
// against the target token, one wave per (sequence, position); nll = logsumexp(x) - x[target] computed as torch's
// log_softmax does (x - max - log(sum exp(x - max))).  Rows with select[row] == 0 are skipped and score 0.
// ---------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void final_nll_kernel(const float* __restrict__ X, const float* gamma,
                                                        const float* beta, const float* dec_w, const float* dec_b,
                                                        const int64_t* __restrict__ targets,
                                                        const uint8_t* __restrict__ select, float* __restrict__ nll,
                                                        int rows, int D, int n_classes) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (select != nullptr && select[row] == 0) {  // wave-uniform
    if (lane == 0) nll[row] = 0.f;
    return;
  }
  const float logit = row_logits<NCH>(X, row, D, lane, gamma, beta, dec_w, dec_b, n_classes);
  const float x = lane < n_classes ? logit : -INFINITY;
  const float m = wave_max(x);
  const float ssum = wave_sum(lane < n_classes ? expf(x - m) : 0.f);
  long t = targets[row];
  if (t < 0 || t >= n_classes) t = 0;           // host validates; keep the lane index in range regardless
  const float xt = __shfl(logit, (int)t);
  if (lane == 0) nll[row] = -((xt - m) - logf(ssum));
}

// ---------------------------------------------------------------------------------------------------
// OA-DM tail: generate.py:198-211.  One workgroup per sequence; only row order[step] is decoded (the logits spread over
// its waves, softmax and draw on the first).
// ---------------------------------------------------------------------------------------------------
// The same logits spread over the OADM_WAVES waves of a workgroup: every wave normalises the row itself (the loads hit
// the same lines) and takes the outputs k = wave, wave + OADM_WAVES, ... - at most four, so their weight loads are all
// in flight together; each dot product runs exactly as in row_logits.  Results go to `out[k]` (LDS).
constexpr int OADM_WAVES = 8;
template <int NCH>
__device__ __forceinline__ void row_logits_spread(const float* X, long row, int D, int lane, int wave,
                                                  const float* gamma, const float* beta, const float* dec_w,
                                                  const float* dec_b, int n_out, float* out) {
  float v[NCH][4];
  load_row<NCH, false>(X, row, D, lane, v);
  normalize_row<NCH>(v, D, lane);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int idx = (c * 64 + lane) * 4;
    if (idx < D) {
      const float4 gm = *(const float4*)(gamma + idx);
      const float4 bt = *(const float4*)(beta + idx);
      v[c][0] = fmaf(v[c][0], gm.x, bt.x); v[c][1] = fmaf(v[c][1], gm.y, bt.y);
      v[c][2] = fmaf(v[c][2], gm.z, bt.z); v[c][3] = fmaf(v[c][3], gm.w, bt.w);
    }
  }
  for (int k0 = wave; k0 < n_out; k0 += 4 * OADM_WAVES) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + u * OADM_WAVES, k = kk < n_out ? kk : n_out - 1;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int idx = (c * 64 + lane) * 4;
        if (idx < D) {
          const float4 w = *(const float4*)(dec_w + (long)k * D + idx);
          s[u] = fmaf(v[c][0], w.x, s[u]); s[u] = fmaf(v[c][1], w.y, s[u]);
          s[u] = fmaf(v[c][2], w.z, s[u]); s[u] = fmaf(v[c][3], w.w, s[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = k0 + u * OADM_WAVES;
      const float t = wave_sum_dpp(s[u]);
      if (kk < n_out && lane == 0) out[kk] = t + dec_b[kk];
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(64 * OADM_WAVES) void final_oadm_kernel(const float* __restrict__ X, const float* gamma,
                                                        const float* beta, const float* dec_w, const float* dec_b,
                                                        int64_t* tokens, const int* __restrict__ order,
                                                        const float* __restrict__ uniforms, StepState* state,
                                                        float* probs, int B, int L, int D, int n_draw_max,
                                                        int x_rows_per_seq, int x_shift, int write_shift, int advance,
                                                        const int* __restrict__ n_draw_step,
                                                        const uint8_t* __restrict__ keep) {
  __shared__ float logits[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
  const int step = state->step;
  const int pos = order[step];
  const long row = (long)b * L + pos;
  const long xrow = (long)b * x_rows_per_seq + (pos >> x_shift);
  row_logits_spread<NCH>(X, xrow, D, lane, wave, gamma, beta, dec_w, dec_b, n_draw_max, logits);
  __syncthreads();
  if (wave != 0) return;
  // classes of this step's draw: the first n_draw_step[step] logits (scaffolded autoregressive sampling changes the
  // range along the sequence, conditional_generation.py:549-550,562), clamped into [1, n_draw_max]
  int n_draw = n_draw_max;
  if (n_draw_step != nullptr) {
    const int n = n_draw_step[step];
    n_draw = n < 1 ? 1 : (n > n_draw_max ? n_draw_max : n);
  }
  const float logit = lane < n_draw ? logits[lane] : 0.f;
  const float p = lane_softmax(logit, lane, n_draw);
  if (probs != nullptr && lane < n_draw_max) probs[((long)step * B + b) * n_draw_max + lane] = lane < n_draw ? p : 0.f;
  // inverse CDF: index = #{k : cdf[k] <= u * cdf[n-1]}, cdf = left-to-right fp32 running sum
  float cdf = 0.f, mycdf = 0.f;
  for (int k = 0; k < n_draw; ++k) {
    cdf += __shfl(p, k);
    if (lane == k) mycdf = cdf;
  }
  const float thr = uniforms[(long)step * B + b] * cdf;
  const unsigned long long below = __ballot(lane < n_draw && mycdf <= thr);
  int idx = __popcll(below);
  if (idx > n_draw - 1) idx = n_draw - 1;
  // write_shift = 1 (autoregressive steps, generate.py:241-244): the draw from position pos is the NEXT token
  // keep: positions whose token is given (a motif spliced into the sequence): the draw is made and dropped
  const bool given = keep != nullptr && pos + write_shift < L && keep[pos + write_shift] != 0;
  if (lane == 0 && pos + write_shift < L && !given) tokens[row + write_shift] = idx;
  // the step counter moves on once the LAST workgroup is done: each one read it when it started, so none reads the
  // new value (saves the one-thread advance launch at the end of every step)
  if (advance && lane == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(&state->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == gridDim.x - 1) {
      __hip_atomic_store(&state->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&state->step, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// D3PM tail: generate.py:279-298.  One wave per (sequence, position); Q[t] and Qbar[t-1] (K x K fp64)
// are staged in LDS once per workgroup.
// ---------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void final_d3pm_kernel(const float* __restrict__ X, const float* gamma,
                                                         const float* beta, const float* dec_w, const float* dec_b,
                                                         int64_t* tokens, const double* __restrict__ Q,
                                                         const double* __restrict__ Qbar,
                                                         const float* __restrict__ uniforms, const StepState* state,
                                                         double* post, int B, int L, int D, int K, int n_timesteps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* sQ = (double*)smem;          // Q[t]       [K][K]
  double* sQb = sQ + K * K;            // Qbar[t-1]  [K][K]
  const int step = state->step;
  const int t = state->t_start - step;
  for (int i = threadIdx.x; i < K * K; i += 256) {
    sQ[i] = Q[(long)t * K * K + i];
    sQb[i] = Qbar[(long)(t - 1) * K * K + i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long rows = (long)B * L;
  const long row = (long)blockIdx.x * 4 + wid;
  if (row >= rows) return;
  const float logit = row_logits<NCH>(X, row, D, lane, gamma, beta, dec_w, dec_b, K);
  const double p = (double)lane_softmax(logit, lane, K);
  const double p2 = p * p;                        // p enters twice (quirk Q3, generate.py:287,289)
  long xt = tokens[row];
  if (xt < 0 || xt >= K) xt = K - 1;               // host validates; keep LDS reads in range regardless
  double marg = 0.0;
  for (int j = 0; j < K; ++j) {
    const double pj = __shfl(p2, j);
    if (lane < K) marg = fma(pj, sQb[j * K + lane], marg);
  }
  if (lane < K) marg *= sQ[lane * K + xt];        // A[l,k] = Q_t[k, x_t[l]]  (generate.py:285)
  // normalise, then inverse CDF, all in fp64 and left to right; every lane walks the same sums
  double total = 0.0;
  for (int k = 0; k < K; ++k) total += __shfl(marg, k);
  const double pk = lane < K ? marg / total : 0.0;
  if (post != nullptr && lane < K) post[((long)step * rows + row) * K + lane] = pk;
  const int n = (t == 1) ? K - 6 : K;              // generate.py:293-294: last step draws canonical AAs only
  const long urow = (t == 1) ? (long)(n_timesteps - 1) : (long)(n_timesteps - 1 - t);
  double cdf = 0.0, mycdf = 0.0;
  for (int k = 0; k < n; ++k) {
    cdf += __shfl(pk, k);
    if (lane == k) mycdf = cdf;
  }
  const double thr = (double)uniforms[urow * rows + row] * cdf;
  int idx = __popcll(__ballot(lane < n && mycdf <= thr));
  if (idx > n - 1) idx = n - 1;
  if (lane == 0) tokens[row] = idx;
}

__global__ void advance_step_kernel(StepState* s) { s->step += 1; }
__global__ void init_state_kernel(StepState* s, int step, int t_start) {
  s->step = step;
  s->t_start = t_start;
  s->done = 0;
}

// Validation on the device (no copy to the host, no synchronisation): out-of-range values are clamped, so that no
// kernel ever indexes outside a table or a sequence, and reported through a host-visible word the next call checks.
__global__ __launch_bounds__(256) void prepare_order_kernel(const int* __restrict__ src, int* __restrict__ dst, int n,
                                                            int hi, unsigned* err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int v = src ? src[i] : i;
  if (v < 0 || v >= hi) {
    err[ED_AERR_ORDER] = 1u;
    v = v < 0 ? 0 : hi - 1;
  }
  dst[i] = v;
}
__global__ __launch_bounds__(256) void prepare_ids_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst,
                                                          long n, int hi, unsigned* err, int code) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long v = src[i];
  if (v < 0 || v >= hi) {
    err[code] = 1u;
    v = hi - 1;
  }
  dst[i] = v;
}
__global__ __launch_bounds__(256) void check_targets_kernel(const int64_t* __restrict__ targets,
                                                            const uint8_t* __restrict__ select, long n, int n_classes,
                                                            unsigned* err) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (select != nullptr && select[i] == 0) return;
  const long v = targets[i];
  if (v < 0 || v >= n_classes) err[ED_AERR_TARGET] = 1u;
}

}  // namespace

hipError_t launch_embed(const int64_t* tokens, const int64_t* tvec, const StepState* state, const float* emb,
                        const float* pe, int n_timesteps, const float* up_w, const float* up_b, float* X,
                        int rows, int B, int D, int E, int n_tokens, hipStream_t stream) {
  if (E > 16 || rows <= 0) return hipErrorInvalidValue;
  const dim3 grid((rows + EMBED_ROWS_PER_BLOCK - 1) / EMBED_ROWS_PER_BLOCK, (D + 255) / 256);
  hipLaunchKernelGGL(embed_kernel, grid, dim3(256), 0, stream, tokens, tvec, state, emb, pe, n_timesteps,
                     up_w, up_b, X, rows, B, D, E, n_tokens);
  return hipGetLastError();
}

__global__ __launch_bounds__(64) void embed_gather_rows_kernel(const int64_t* __restrict__ tokens,
                                                               const float* __restrict__ table, float* __restrict__ X,
                                                               int L, int D, int n_tokens, PosBit sel, bool compact) {
  const long row = (long)blockIdx.x * L + sel.get();
  long tok = tokens[row];
  if (tok < 0 || tok >= n_tokens) tok = n_tokens - 1;
  const float4* src = (const float4*)(table + tok * D);
  float4* dst = (float4*)(X + (compact ? (long)blockIdx.x : row) * D);
  for (int c = threadIdx.x; c < D / 4; c += 64) dst[c] = src[c];
}

hipError_t launch_embed_gather_rows(const int64_t* tokens, const float* table, float* X, int B, int L, int D,
                                    int n_tokens, PosBit sel, hipStream_t stream, bool compact) {
  if (D % 4 != 0 || B <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_gather_rows_kernel, dim3(B), dim3(64), 0, stream, tokens, table, X, L, D, n_tokens, sel,
                     compact);
  return hipGetLastError();
}

hipError_t launch_embed_gather(const int64_t* tokens, const float* table, float* X, int rows, int D, int n_tokens,
                               hipStream_t stream) {
  if (D % 4 != 0 || rows <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_gather_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, tokens, table, X, rows, D,
                     n_tokens);
  return hipGetLastError();
}

#define ED_NCH_DISPATCH(C, CALL)              \
  do {                                        \
    const int nch_ = ((C) + 255) / 256;       \
    if (nch_ <= 1) { CALL(1); }               \
    else if (nch_ <= 2) { CALL(2); }          \
    else if (nch_ <= 4) { CALL(4); }          \
    else if (nch_ <= 5) { CALL(5); }          \
    else if (nch_ <= 8) { CALL(8); }          \
    else return hipErrorInvalidValue;         \
  } while (0)

hipError_t launch_ln_act(const void* in, bool in_bf16, void* out, bool out_bf16, const float* gamma,
                         const float* beta, int rows, int C, const LnRowMap& map, int act, hipStream_t stream) {
  if (C % 4 != 0 || rows <= 0) return hipErrorInvalidValue;
  if (out_bf16) {
    // one row per wave measured fastest (2/4/8 rows per wave with a register prefetch of the next row: +1..+7 us
    // at config-2 size; 1-16 waves per block: no difference)
    const dim3 fgrid((rows + 3) / 4), fblock(256);
    const int nc8 = C / 512, nc4 = (C % 512 + 255) / 256;
#define ED_FAST_ACT(A8, A4, AC)                                                                                  \
    if (act == AC) {                                                                                             \
      if (in_bf16)                                                                                               \
        hipLaunchKernelGGL((ln_act_fast_kernel<A8, A4, true, AC>), fgrid, fblock, 0, stream, in, (bf16_t*)out,   \
                           gamma, beta, rows, C, map);                                                           \
      else                                                                                                       \
        hipLaunchKernelGGL((ln_act_fast_kernel<A8, A4, false, AC>), fgrid, fblock, 0, stream, in, (bf16_t*)out,  \
                           gamma, beta, rows, C, map);                                                           \
    }
#define ED_FAST(A8, A4)                                                                                          \
    if (nc8 == A8 && nc4 == A4) {                                                                                \
      ED_FAST_ACT(A8, A4, 0) ED_FAST_ACT(A8, A4, 1) ED_FAST_ACT(A8, A4, 2)                                       \
      return hipGetLastError();                                                                                  \
    }
    ED_FAST(0, 1) ED_FAST(0, 2) ED_FAST(1, 0) ED_FAST(1, 1) ED_FAST(1, 2) ED_FAST(2, 0) ED_FAST(2, 1) ED_FAST(2, 2)
    ED_FAST(3, 0) ED_FAST(3, 1) ED_FAST(3, 2) ED_FAST(4, 0)
#undef ED_FAST
#undef ED_FAST_ACT
    return hipErrorInvalidValue;
  }
  const dim3 grid((rows + 3) / 4), block(256);
#define ED_CALL(N)                                                                                            \
  if (in_bf16 && out_bf16)                                                                                    \
    hipLaunchKernelGGL((ln_act_kernel<N, true, true>), grid, block, 0, stream, in, out, gamma, beta, rows, C, \
                       map, act);                                                          \
  else if (in_bf16)                                                                                           \
    hipLaunchKernelGGL((ln_act_kernel<N, true, false>), grid, block, 0, stream, in, out, gamma, beta, rows,   \
                       C, map, act);                                                       \
  else if (out_bf16)                                                                                          \
    hipLaunchKernelGGL((ln_act_kernel<N, false, true>), grid, block, 0, stream, in, out, gamma, beta, rows,   \
                       C, map, act);                                                       \
  else                                                                                                        \
    hipLaunchKernelGGL((ln_act_kernel<N, false, false>), grid, block, 0, stream, in, out, gamma, beta, rows,   \
                       C, map, act)
  ED_NCH_DISPATCH(C, ED_CALL);
#undef ED_CALL
  return hipGetLastError();
}

bool reduce_ln_supported(const GemmProblem& q) {
  // the GEMM's output rows must be its flat rows (the LayerNorm's input map then indexes the partial sums directly)
  const bool plain_out = q.seq_rows == 0 || (q.out_pitch == q.seq_rows && q.out_sel.order == nullptr);
  if (q.resid && (q.seq_rows != 0 || q.out_bf16)) return false;
  return q.ksplit > 1 && q.partial && plain_out && q.N % 4 == 0 && q.N <= 4096;
}

hipError_t launch_reduce_ln(const GemmProblem& q, bf16_t* out, const float* gamma, const float* beta, int rows,
                            const LnRowMap& map, int act, hipStream_t stream) {
  if (!reduce_ln_supported(q) || rows <= 0 || act < 0 || act > 2) return hipErrorInvalidValue;
  const int C = q.N;
  const dim3 grid((unsigned)rows), block((unsigned)(((C / 4) + 63) / 64 * 64));
  const long pstride = (long)q.B * q.L * q.N;
#define ED_RL(AC)                                                                                                 \
  hipLaunchKernelGGL((reduce_ln_kernel<AC>), grid, block, 0, stream, (const float*)q.partial, pstride, q.ksplit,  \
                     q.bias, q.out_bf16 ? 1 : 0, q.resid, q.resid_mul, q.resid_sel, q.resid ? (float*)q.out : nullptr, \
                     out, gamma, beta, C, map)
  if (act == 0) ED_RL(0);
  else if (act == 1) ED_RL(1);
  else ED_RL(2);
#undef ED_RL
  return hipGetLastError();
}

hipError_t launch_final_logits(const float* X, const float* gamma, const float* beta, const float* dec_w,
                               const float* dec_b, float* logits, int rows, int D, int n_tokens,
                               hipStream_t stream) {
  if (D % 4 != 0 || n_tokens > 64 || rows <= 0) return hipErrorInvalidValue;
  const dim3 grid((rows + 3) / 4), block(256);
#define ED_CALL(N) \
  hipLaunchKernelGGL((final_logits_kernel<N>), grid, block, 0, stream, X, gamma, beta, dec_w, dec_b, logits, rows, D, n_tokens)
  ED_NCH_DISPATCH(D, ED_CALL);
#undef ED_CALL
  return hipGetLastError();
}

hipError_t launch_final_nll(const float* X, const float* gamma, const float* beta, const float* dec_w,
                            const float* dec_b, const int64_t* targets, const uint8_t* select, float* nll, int rows,
                            int D, int n_classes, hipStream_t stream) {
  if (D % 4 != 0 || n_classes > 64 || n_classes < 1 || rows <= 0) return hipErrorInvalidValue;
  const dim3 grid((rows + 3) / 4), block(256);
#define ED_CALL(N)                                                                                                \
  hipLaunchKernelGGL((final_nll_kernel<N>), grid, block, 0, stream, X, gamma, beta, dec_w, dec_b, targets, select, \
                     nll, rows, D, n_classes)
  ED_NCH_DISPATCH(D, ED_CALL);
#undef ED_CALL
  return hipGetLastError();
}

hipError_t launch_final_oadm(const float* X, int x_rows_per_seq, int x_shift, const float* gamma, const float* beta,
                             const float* dec_w, const float* dec_b, int64_t* tokens, const int* order,
                             const float* uniforms, StepState* state, float* probs, int B, int L, int D,
                             int n_draw, hipStream_t stream, int write_shift, bool advance, const int* n_draw_step,
                             const uint8_t* keep) {
  if (D % 4 != 0 || n_draw > 64 || n_draw < 1 || B <= 0) return hipErrorInvalidValue;
  const dim3 grid(B), block(64 * OADM_WAVES);
#define ED_CALL(N)                                                                                             \
  hipLaunchKernelGGL((final_oadm_kernel<N>), grid, block, 0, stream, X, gamma, beta, dec_w, dec_b, tokens, order, \
                     uniforms, state, probs, B, L, D, n_draw, x_rows_per_seq, x_shift, write_shift, advance ? 1 : 0, \
                     n_draw_step, keep)
  ED_NCH_DISPATCH(D, ED_CALL);
#undef ED_CALL
  return hipGetLastError();
}

hipError_t launch_final_d3pm(const float* X, const float* gamma, const float* beta, const float* dec_w,
                             const float* dec_b, int64_t* tokens, const double* Q, const double* Qbar,
                             const float* uniforms, const StepState* state, double* post, int B, int L, int D,
                             int K, int n_timesteps, hipStream_t stream) {
  if (D % 4 != 0 || K > 62 || K < 7 || B <= 0) return hipErrorInvalidValue;
  const long rows = (long)B * L;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const size_t smem = (size_t)(2 * K * K) * sizeof(double);
#define ED_CALL(N)                                                                                              \
  hipLaunchKernelGGL((final_d3pm_kernel<N>), grid, block, smem, stream, X, gamma, beta, dec_w, dec_b, tokens, Q, \
                     Qbar, uniforms, state, post, B, L, D, K, n_timesteps)
  ED_NCH_DISPATCH(D, ED_CALL);
#undef ED_CALL
  return hipGetLastError();
}

hipError_t launch_prepare_order(const int* src, int* dst, int n, int hi, unsigned* err, hipStream_t stream) {
  if (n <= 0 || hi <= 0 || !dst || !err) return hipErrorInvalidValue;
  hipLaunchKernelGGL(prepare_order_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, dst, n, hi, err);
  return hipGetLastError();
}

hipError_t launch_prepare_ids(const int64_t* src, int64_t* dst, long n, int hi, unsigned* err, int code,
                              hipStream_t stream) {
  if (n <= 0 || hi <= 0 || !src || !dst || !err) return hipErrorInvalidValue;
  hipLaunchKernelGGL(prepare_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, n, hi, err, code);
  return hipGetLastError();
}

hipError_t launch_check_targets(const int64_t* targets, const uint8_t* select, long n, int n_classes, unsigned* err,
                                hipStream_t stream) {
  if (n <= 0 || !targets || !err) return hipErrorInvalidValue;
  hipLaunchKernelGGL(check_targets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, targets, select, n,
                     n_classes, err);
  return hipGetLastError();
}

hipError_t launch_init_state(StepState* state, int step, int t_start, hipStream_t stream) {
  hipLaunchKernelGGL(init_state_kernel, dim3(1), dim3(1), 0, stream, state, step, t_start);
  return hipGetLastError();
}

hipError_t launch_advance_step(StepState* state, hipStream_t stream) {
  hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, stream, state);
  return hipGetLastError();
}

}  // namespace ed
