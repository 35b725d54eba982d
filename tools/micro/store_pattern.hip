// Micro-benchmark: how fast can 256 workgroups (512 threads, one per CU) each store a 256 x 320 tile of fp32 / bf16 from
// registers, with the tuned kernels' epilogue pattern (a lane stores 16 B; the four lanes of a row give 64 contiguous
// bytes (bf16) or four 16-byte pieces at a 32-byte stride (fp32)) against row-contiguous patterns?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int N = 1280, ROWS = 16384, BM = 256, BN = 320;

// mode 0: tuned-kernel pattern fp32 (two 16-B stores per lane and column-tile pair, 32-B lane stride)
// mode 1: tuned-kernel pattern bf16 (one 16-B store per lane and pair: 64 B contiguous per row)
// mode 2: row-contiguous fp32: a wave writes its 64 x 160 sub-tile row by row, 16 B per lane (40 lanes per row)
// mode 3: row-contiguous bf16: 20 lanes per row
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(float* out, float seed) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
  const int tile = blockIdx.x, tile_m = tile / 4, tile_n = tile % 4;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 v = {seed, seed + 1, seed + 2, seed + 3};
  if (MODE == 0 || MODE == 1) {
    const int nb = tile_n * BN + wc * 160 + 8 * fq;
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
      const long row = (long)tile_m * BM + wr * 64 + ip * 16 + fr;
#pragma unroll
      for (int jp = 0; jp < 5; ++jp) {
        if (MODE == 0) {
          *(f32x4*)(out + row * N + nb + 32 * jp) = v;
          *(f32x4*)(out + row * N + nb + 32 * jp + 4) = v;
        } else {
          *(f32x4*)((char*)out + (row * N + nb + 32 * jp) * 2) = v;
        }
      }
    }
  } else if (MODE == 2) {
    // 64 rows x 160 floats = 2560 16-B chunks per wave = 40 instructions; chunk c -> row c / 40, piece c % 40
#pragma unroll 8
    for (int k = 0; k < 40; ++k) {
      const int c = k * 64 + lane, r = c / 40, p = c - r * 40;
      const long row = (long)tile_m * BM + wr * 64 + r;
      *(f32x4*)(out + row * N + tile_n * BN + wc * 160 + 4 * p) = v;
    }
  } else {
#pragma unroll 4
    for (int k = 0; k < 20; ++k) {
      const int c = k * 64 + lane, r = c / 20, p = c - r * 20;
      const long row = (long)tile_m * BM + wr * 64 + r;
      *(f32x4*)((char*)out + (row * N + tile_n * BN + wc * 160 + 8 * p) * 2) = v;
    }
  }
}

template <int MODE>
void run(float* buf, const char* what, double bytes) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(256), dim3(512), 0, 0, buf, (float)i);
  hipEventRecord(a);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(256), dim3(512), 0, 0, buf, (float)i);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("%-52s %6.1f us/launch  %5.2f TB/s\n", what, ms / 50 * 1e3, bytes / (ms / 50 * 1e-3) / 1e12);
}

int main() {
  float* buf;
  hipMalloc(&buf, (size_t)ROWS * N * 4 + 4096);
  run<0>(buf, "fp32, epilogue pattern (16 B at 32-B stride x2)", (double)ROWS * N * 4);
  run<2>(buf, "fp32, row-contiguous (640 B runs)", (double)ROWS * N * 4);
  run<1>(buf, "bf16, epilogue pattern (64-B runs)", (double)ROWS * N * 2);
  run<3>(buf, "bf16, row-contiguous (320 B runs)", (double)ROWS * N * 2);
  return 0;
}
