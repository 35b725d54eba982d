// Micro-benchmark: what does the chip sustain on a bare bf16 MFMA loop?  256 workgroups x 512 threads (two waves per
// SIMD, as the tuned kernels run) or x 256 threads (one per SIMD); every wave issues v_mfma_f32_16x16x32_bf16
// back to back on register-resident operands, 40 independent accumulators (the tuned kernels' register tile), random or
// zero data.  Prints TFLOP/s, MFMA issue rate and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int NACC = 40, ITERS = 4000;

__global__ __launch_bounds__(512) void mfma_loop(const uint4* in, float* out, long long* stamps) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  bf16x8 a[4], b[10];
  for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, in[(tid * 16 + i) & 0xffff]);
  for (int i = 0; i < 10; ++i) b[i] = __builtin_bit_cast(bf16x8, in[(tid * 16 + 4 + i) & 0xffff]);
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j * 4 + i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[tid] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = (long long)(c1 - c0);
    stamps[2 * blockIdx.x + 1] = (long long)(r1 - r0);
  }
}

// the same register tile (160 accumulator registers) on v_mfma_f32_32x32x16_bf16: 10 accumulators of 16, twice the
// FLOPs per operand register read
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void mfma_loop32(const uint4* in, float* out, long long* stamps) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  bf16x8 a[2], b[5];
  for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, in[(tid * 16 + i) & 0xffff]);
  for (int i = 0; i < 5; ++i) b[i] = __builtin_bit_cast(bf16x8, in[(tid * 16 + 4 + i) & 0xffff]);
  f32x16 acc[10];
  for (int i = 0; i < 10; ++i)
    for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j * 2 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[j * 2 + i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x16 s = acc[0];
  for (int i = 1; i < 10; ++i) s += acc[i];
  float t = 0.f;
  for (int k = 0; k < 16; ++k) t += s[k];
  out[tid] = t;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = (long long)(c1 - c0);
    stamps[2 * blockIdx.x + 1] = (long long)(r1 - r0);
  }
}

int main() {
  uint4* in;
  float* out;
  long long* stamps;
  hipMalloc(&in, 65536 * 16);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&stamps, 256 * 16);
  std::vector<unsigned short> h(65536 * 8);
  for (int zero = 0; zero < 2; ++zero) {
    srand(1);
    for (auto& v : h) {  // bf16 uniform in [-1, 1): random sign, exponent 118..126, random mantissa
      const unsigned r = (unsigned)rand();
      v = zero ? 0 : (unsigned short)(((r & 1) << 15) | ((118 + (r >> 1) % 9) << 7) | ((r >> 8) & 0x7f));
    }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int shape32 = 0; shape32 < 2; ++shape32)
    for (int threads : {512, 256}) {
      auto* kern = shape32 ? mfma_loop32 : mfma_loop;
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, in, out, stamps);  // > 2 s of load
      hipEventRecord(a);
      const int reps = 20;
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, in, out, stamps);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      long long hs[512];
      hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
      double clk = 0;
      for (int i = 0; i < 256; ++i) clk += (double)hs[2 * i] / (double)hs[2 * i + 1] * 100.0;
      const int n_mfma = shape32 ? 20 : NACC;  // per iteration; both forms are 2 * 16 * 16 * 32 * 40 FLOPs per iteration
      const double flops = 2.0 * 16 * 16 * 32 * NACC * (double)ITERS * (threads / 64) * 256 * reps;
      const double per_mfma = (double)hs[0] / ((double)n_mfma * ITERS * (threads / 256));  // shader cycles per MFMA issued on a SIMD
      printf("%s %s data, %d waves per SIMD: %7.1f TFLOP/s  (%4.1f %% of 2500), in-kernel clock %4.0f MHz, %5.2f cycles per MFMA on a SIMD\n",
             shape32 ? "32x32x16" : "16x16x32", zero ? "zero  " : "random", threads / 256, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 25.0, clk / 256,
             per_mfma);
    }
  }
  return 0;
}
