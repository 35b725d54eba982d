// LayerNorm + activation of a GEMM's result inside the GEMM's own epilogue (tuned bf16 kernels, gfx950).
//
// Every GEMM of a ByteNet block feeds a LayerNorm (PFF1 -> LN2, conv -> LN3, PFF2 -> LN1 of the next block;
// evodiff/model.py:113-117 + sequence_models ByteNetBlock).  A LayerNorm needs whole rows of N channels while a tuned
// tile holds BN = 320 / 256 of them, so the tiles_n workgroups that share a row tile exchange their per-row partial
// statistics through memory INSIDE the launch and then each normalises, activates and stores its own columns straight
// from the fp32 accumulators: the stand-alone LayerNorm launch, its kernel boundary and the round trip of the raw GEMM
// result through memory disappear (and the statistics are taken on the fp32 sums, not on their bf16 roundings).
//
// Statistics.  Per row: a lane's 8 NT values give (mean, M2 about that mean); the 8 lane partials of a row inside a
// workgroup and then the tiles_n workgroup partials are merged with the equal-count form of Chan's update
//   mean = avg(mean_i),  M2 = sum(M2_i) + n * sum((mean_i - mean)^2)
// in a fixed order, so the result does not depend on timing, tile height or dispatch order.
//
// Exchange (cdna_hip_programming.md Guideline 16; MI355X_MICROARCH.md "Valid forms", first row of the sc1 table).  The
// memory protocol assumes nothing about workgroup -> XCD placement or about which partner arrives first.  Forward
// progress does rest on one assumption: a waiting workgroup's partners must become resident while it spins.  The
// tiles_n workgroups of a row tile have block ids at most 8 * (tiles_n - 1) <= 24 apart (tile order of the tile bodies),
// and the hardware dispatches the workgroups of a launch in block-id order, so the workgroup with the lowest id that
// is still waiting always has every partner dispatched already (or next in line as soon as any workgroup ends): the
// resident set is a contiguous window of ids that moves forward.  A dispatcher that did not go in id order could strand
// a launch with more workgroups than fit on the chip; the bounded poll below then turns that into a reported error.
// The hand-off itself:
//   producer: each of the first BM threads stores its row's 8-byte {mean, M2} with a write-through (sc1) store; every
//             storing wave drains its stores (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds 1 to the row tile's
//             counter (agent-scope atomic);
//   consumer: that lane polls the counter with relaxed agent-scope (sc1) loads until all tiles_n workgroups of the row
//             tile have arrived; workgroup barrier; every load of the handed-off bytes is an sc1 load to registers.
// Counters are never reset: a launch adds exactly tiles_n (a power of two) to each counter it touches, launches on a
// stream run one after the other, so a counter is a multiple of tiles_n between launches and the value the add returns
// tells the arrival target, (old & ~(tiles_n - 1)) + tiles_n; wrap-around at 2^32 keeps that true.  A captured graph
// therefore needs no memset node per launch.  The poll is bounded (LN_SPIN_TICKS of the 100-MHz clock): if a partner
// never arrives the workgroup gives up, raises the error word (host-visible memory) and finishes with whatever
// statistics it has - wrong results with a reported error instead of a hung GPU.
#pragma once
#include "kernels.hpp"

namespace ed {

typedef __attribute__((ext_vector_type(4))) float lnf32x4;
typedef __attribute__((ext_vector_type(2))) float lnf32x2;
typedef __attribute__((address_space(1))) unsigned long long ln_gu64;
typedef __attribute__((address_space(1))) unsigned int ln_gu32;

constexpr float LN_FUSED_EPS = 1e-5f;            // nn.LayerNorm default (as elementwise.hip)
constexpr unsigned long long LN_SPIN_TICKS = 200000000ull;  // 2 s of the 100-MHz clock: far beyond any delay CU contention
                                                          // can cause (a launch is ~0.2 ms), short enough not to look hung

// x * sigmoid(x (a + b x^2 + c x^4)): the fit of the exact-erf GELU the stand-alone bf16 LayerNorm uses
// (elementwise.hip, 2.6e-5 absolute), two elements at a time.
__device__ __forceinline__ lnf32x2 ln_gelu2(lnf32x2 x) {
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float A = -1.59501577f * LOG2E, B = -7.40112924e-02f * LOG2E, C = 7.03033587e-04f * LOG2E;
  lnf32x2 xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -9.0f, 9.0f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -9.0f, 9.0f);
  const lnf32x2 x2 = xc * xc;
  const lnf32x2 w = xc * __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, (lnf32x2)(C), (lnf32x2)(B)), (lnf32x2)(A));
  lnf32x2 e;
  e.x = __builtin_amdgcn_exp2f(w.x);
  e.y = __builtin_amdgcn_exp2f(w.y);
  const lnf32x2 d = e + 1.0f;
  lnf32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return x * r;
}

// The epilogue in two halves, both called by ALL threads of the workgroup after the K loop's last barrier (every LDS
// stage is free and no copy is in flight).  acc[jn][ip] holds the final fp32 values (bias and residual included) of
//   position  l0 + wr*64 + ip*16 + fr,  channels  n0 + wc*(BN/2) + 8*fq + 32*(jn >> 1) + 4*(jn & 1) + 0..3
// as in the kernels' own epilogues.  ln_publish: lane partials -> workgroup partial per row -> write-through store ->
// arrival; returns the arrival target (meaningful in thread 0).  Between the halves the kernel may store its raw
// result (PFF2: the fp32 residual stream), which hides the other workgroups' arrival.  ln_finish: wait, whole-row
// statistics, normalise + GELU + bf16 store.  `row_base` = flat row (b*L + l0) of the tile's first position.
template <int NT, int WM, typename Acc>
__device__ __forceinline__ unsigned ln_publish(const GemmProblem& g, Acc& acc, char* lds, int tile_m, int tile_n,
                                               int tiles_n, int tid_in) {
  constexpr int BM = WM * 64, NJ = 2 * NT, NL = 4 * NJ;  // NL values per lane and position
  const LnFuse& f = g.ln;
  // every lane-derived index is formed here, from one laundered register: nothing of the epilogue stays live across
  // the K loop (the fully unrolled loops run at the register limit)
  int tid = tid_in;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 1, wc = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  lnf32x2* const red = (lnf32x2*)lds;                 // [BM][8] lane partials {mean, M2}
  // this tile's gamma / beta go to LDS ([2][BN] behind the row statistics): the loads fly under the statistics and
  // ln_finish reads them at LDS latency instead of five rounds of global loads
  constexpr int BN = NT * 64;
  float* const gb = (float*)(lds + BM * 64 + BM * 8);
  lnf32x4 gv = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
  if (tid < BN / 4) {
    gv = *(const lnf32x4*)(as_global(f.gamma) + tile_n * BN + 4 * tid);
    bv = *(const lnf32x4*)(as_global(f.beta) + tile_n * BN + 4 * tid);
  }

  // ---- 1. lane partials
#pragma unroll
  for (int ip = 0; ip < 4; ++ip) {
    if (f.dbg & 8) {
      red[(wr * 64 + ip * 16 + fr) * 8 + wc * 4 + fq] = lnf32x2{0.f, 1.f};
      continue;
    }
    // one pass, two elements per instruction (v_pk_add_f32 / v_pk_fma_f32): sum and sum of squares of the lane's NL
    // values; M2 about the lane's own mean is q - s^2 / NL (a cancellation over NL = 32 / 40 values only: the merge
    // of the partials below is the stable form)
    lnf32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
      const lnf32x2 a = {acc[jn][ip][0], acc[jn][ip][1]}, c = {acc[jn][ip][2], acc[jn][ip][3]};
      s2 += a;
      q2 = __builtin_elementwise_fma(a, a, q2);
      s2 += c;
      q2 = __builtin_elementwise_fma(c, c, q2);
    }
    const float sl = s2.x + s2.y;
    const float m = sl * (1.0f / NL);
    const float q = fmaxf(fmaf(-sl, m, q2.x + q2.y), 0.f);
    red[(wr * 64 + ip * 16 + fr) * 8 + wc * 4 + fq] = lnf32x2{m, q};
  }
  if (tid < BN / 4) {
    *(lnf32x4*)(gb + 4 * tid) = gv;
    *(lnf32x4*)(gb + BN + 4 * tid) = bv;
  }
  __syncthreads();

  // ---- 2. workgroup partial of each row, published write-through
  ln_gu64* const slab = (ln_gu64*)f.stats + ((long)tile_m * tiles_n) * BM;
  if (tid < BM) {
    float ms = 0.f, qs = 0.f;
    lnf32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      p[i] = red[tid * 8 + i];
      ms += p[i].x;
      qs += p[i].y;
    }
    const float mw = ms * 0.125f;
    float dv = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) dv = fmaf(p[i].x - mw, p[i].x - mw, dv);
    const float m2 = fmaf((float)NL, dv, qs);
    const unsigned long long bits = (unsigned long long)__float_as_uint(mw) | ((unsigned long long)__float_as_uint(m2) << 32);
    __hip_atomic_store(slab + (long)tile_n * BM + tid, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier the signalling lane joins
  __syncthreads();

  // ---- 3. arrive
  unsigned target = 0;
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add((ln_gu32*)f.counters + tile_m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    target = (old & ~(unsigned)(tiles_n - 1)) + (unsigned)tiles_n;
  }
  return target;
}

// RAW: the raw fp32 result (PFF2: the residual stream) is stored here too, to g.out, interleaved with the normalised
// rows AFTER the wait and the statistics loads: issued in front of them (as a separate store loop between the halves)
// the 40 16-byte stores of a lane sit in front of the poll's and the statistics' loads in the wave's memory queue,
// and the epilogue waits for their drain twice.
template <int NT, int WM, bool RAW = false, typename Acc>
__device__ __forceinline__ void ln_finish(const GemmProblem& g, Acc& acc, char* lds, unsigned target, int tile_m,
                                          int tiles_n, long row_base, int rows_valid, int n0, int tid_in,
                                          long long* stamp = nullptr) {  // stamp: diagnostic builds, two 100-MHz times
  constexpr int BM = WM * 64, BN = NT * 64;
  const LnFuse& f = g.ln;
  int tid = tid_in;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 1, wc = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  lnf32x2* const rowstat = (lnf32x2*)(lds + BM * 64);  // [BM] {mean, rstd}; behind ln_publish's [BM][8] partials
  ln_gu64* const slab = (ln_gu64*)f.stats + ((long)tile_m * tiles_n) * BM;
  // ---- wait for the other workgroups of the row tile (one lane polls, relaxed; bounded)
  if (tid == 0 && tiles_n > 1 && !(f.dbg & 1)) {
    ln_gu32* const ctr = (ln_gu32*)f.counters + tile_m;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > LN_SPIN_TICKS) {  // a partner never arrived: report, do not hang
        __hip_atomic_store((ln_gu32*)f.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  if (stamp && tid == 0) stamp[0] = (long long)__builtin_amdgcn_s_memrealtime();  // all partners arrived

  // ---- 4. whole-row statistics from the tiles_n workgroup partials (sc1 loads), fixed order
  if (tid < BM) {
    float ms = 0.f, qs = 0.f, dv = 0.f;
    float mean;
    if (tiles_n <= 8) {  // one batch of loads in flight, one wait (same order of additions as the loop form)
      unsigned long long bits[8];
#pragma unroll
      for (int w = 0; w < 8; ++w)
        bits[w] = w < tiles_n ? __hip_atomic_load(slab + (long)w * BM + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
      for (int w = 0; w < 8; ++w)
        if (w < tiles_n) {
          ms += __uint_as_float((unsigned)bits[w]);
          qs += __uint_as_float((unsigned)(bits[w] >> 32));
        }
      mean = ms / (float)tiles_n;
#pragma unroll
      for (int w = 0; w < 8; ++w)
        if (w < tiles_n) {
          const float d = __uint_as_float((unsigned)bits[w]) - mean;
          dv = fmaf(d, d, dv);
        }
    } else {
      for (int w = 0; w < tiles_n; ++w) {
        const unsigned long long bits = __hip_atomic_load(slab + (long)w * BM + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ms += __uint_as_float((unsigned)bits);
        qs += __uint_as_float((unsigned)(bits >> 32));
      }
      mean = ms / (float)tiles_n;
      for (int w = 0; w < tiles_n; ++w) {
        const unsigned long long bits = __hip_atomic_load(slab + (long)w * BM + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float d = __uint_as_float((unsigned)bits) - mean;
        dv = fmaf(d, d, dv);
      }
    }
    const float var = fmaf((float)BN, dv, qs) / (float)(BN * tiles_n);
    rowstat[tid] = lnf32x2{mean, rsqrtf(var + LN_FUSED_EPS)};
  }
  __syncthreads();
  if (stamp && tid == 0) stamp[2] = (long long)__builtin_amdgcn_s_memrealtime();  // whole-row statistics in LDS

  // ---- 5. normalise, scale / shift, activate, round to bf16, store: eight consecutive channels per 16-byte store
  const int nb = n0 + wc * (BN / 2) + 8 * fq;
  typedef __attribute__((ext_vector_type(8))) __bf16 lnbf16x8;
  float rs[4], sh[4];
  long orow[4];
#pragma unroll
  for (int ip = 0; ip < 4; ++ip) {
    const int r = wr * 64 + ip * 16 + fr;
    const lnf32x2 st = rowstat[r];
    rs[ip] = st.y;
    sh[ip] = -st.x * st.y;
    long R = row_base + r;                       // flat output row of the GEMM
    if (f.seq_len > 0) {
      const long bs = R / f.seq_len;
      R = bs * f.out_pitch + f.out_row0 + (R - bs * f.seq_len);
    }
    orow[ip] = r < rows_valid ? R : -1;
  }
  float* const raw = RAW ? as_global((float*)g.out) + (row_base + wr * 64 + fr) * g.N + nb : nullptr;
  const float* const gb = (const float*)(lds + BM * 64 + BM * 8) + wc * (BN / 2) + 8 * fq;  // staged by ln_publish
#pragma unroll
  for (int jp = 0; jp < NT; ++jp) {
    const lnf32x4 g0 = *(const lnf32x4*)(gb + 32 * jp), g1 = *(const lnf32x4*)(gb + 32 * jp + 4);
    const lnf32x4 b0 = *(const lnf32x4*)(gb + BN + 32 * jp), b1 = *(const lnf32x4*)(gb + BN + 32 * jp + 4);
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
      const lnf32x4 v = acc[2 * jp][ip], w = acc[2 * jp + 1][ip];
      if constexpr (RAW) {
        if (orow[ip] >= 0 && !(f.dbg & 16)) {
          *(lnf32x4*)(raw + (long)(ip * 16) * g.N + 32 * jp) = v;
          *(lnf32x4*)(raw + (long)(ip * 16) * g.N + 32 * jp + 4) = w;
        }
      }
      lnf32x2 y[4] = {{v[0], v[1]}, {v[2], v[3]}, {w[0], w[1]}, {w[2], w[3]}};
      const lnf32x2 gm[4] = {{g0[0], g0[1]}, {g0[2], g0[3]}, {g1[0], g1[1]}, {g1[2], g1[3]}};
      const lnf32x2 bt[4] = {{b0[0], b0[1]}, {b0[2], b0[3]}, {b1[0], b1[1]}, {b1[2], b1[3]}};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        y[k] = __builtin_elementwise_fma(__builtin_elementwise_fma(y[k], (lnf32x2)(rs[ip]), (lnf32x2)(sh[ip])), gm[k], bt[k]);
        if (!(f.dbg & 2)) y[k] = ln_gelu2(y[k]);
      }
      if (orow[ip] >= 0 && !(f.dbg & 4)) {
        const lnbf16x8 pk = {(__bf16)y[0].x, (__bf16)y[0].y, (__bf16)y[1].x, (__bf16)y[1].y,
                             (__bf16)y[2].x, (__bf16)y[2].y, (__bf16)y[3].x, (__bf16)y[3].y};
        *(lnbf16x8*)(as_global(f.out) + orow[ip] * g.N + nb + 32 * jp) = pk;
      }
    }
  }
}

}  // namespace ed
