// Tuned bf16 implicit-GEMM kernel for the ByteNet block on gfx950 (MI355X): the launchers and the kernel that wraps the
// tile body of gemm_bf16_tile.hpp (tiling, K loop, LDS image and epilogue are described there).
#include "gemm_bf16_tile.hpp"

namespace ed {

namespace {

using gemm_tile::BK;
using gemm_tile::BM;

template <int NT, bool CONV, bool OUT_BF16, bool RESID, int ABLATE = 0, int RT = 0, int WM = 4, bool SPLIT = false,
          bool LNF = false>
__global__ __launch_bounds__(WM * 128) void gemm_bf16_kernel(GemmProblem g, int tiles_n, int tiles_total) {
  __shared__ __attribute__((aligned(1024))) char lds[gemm_tile::lds_bytes(NT, WM)];
  gemm_tile::tile<NT, CONV, OUT_BF16, RESID, ABLATE, RT, WM, SPLIT, LNF>(g, tiles_n, tiles_total, lds, (int)blockIdx.x,
                                                                        (int)blockIdx.y);
}

template <int NT, bool CONV, int WM>
hipError_t launch_nt(const GemmProblem& p, hipStream_t stream) {
  constexpr int BMT = WM * 64;
  const int tiles_n = p.N / (NT * 64);
  const int tiles_total = p.B * ((p.L + BMT - 1) / BMT) * tiles_n;
  const dim3 grid(tiles_total), block(WM * 128);
  if (p.ln.gamma) {  // LayerNorm + GELU in the epilogue
    if (!gemm_bf16_ln_supported(p) || !p.ln.beta || !p.ln.out || !p.ln.stats || !p.ln.counters || !p.ln.error)
      return hipErrorInvalidValue;
    if (p.ablate == 7 && p.stamps) {  // diagnostic: the fused kernels of the 640M geometry with clock stamps
      if constexpr (!CONV && NT == 5 && WM == 4) {
        const bool whole = ((long)p.B * p.L) % BMT == 0 && (p.B == 1 || p.L % BMT == 0) && p.resid_mul == 1;
        if (p.resid && whole && p.K == 20 * BK)
          hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 3, 20, WM, false, true>), grid, block, 0, stream, p, tiles_n, tiles_total);
        else if (!p.resid)
          hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, true, false, 3, 0, WM, false, true>), grid, block, 0, stream, p, tiles_n, tiles_total);
        else
          return hipErrorInvalidValue;
        return hipGetLastError();
      } else {
        return hipErrorInvalidValue;
      }
    }
    if (p.resid) {
      if constexpr (!CONV) {
        const bool whole = ((long)p.B * p.L) % BMT == 0 && (p.B == 1 || p.L % BMT == 0) && p.resid_mul == 1;
        if (whole && p.K == 20 * BK)
          hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 0, 20, WM, false, true>), grid, block, 0, stream, p, tiles_n, tiles_total);
        else if (whole && NT == 4 && p.K == 8 * BK)
          hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 0, 8, WM, false, true>), grid, block, 0, stream, p, tiles_n, tiles_total);
        else
          hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 0, 0, WM, false, true>), grid, block, 0, stream, p, tiles_n, tiles_total);
      } else {
        return hipErrorInvalidValue;
      }
    } else {
      hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, true, false, 0, 0, WM, false, true>), grid, block, 0, stream, p, tiles_n, tiles_total);
    }
    return hipGetLastError();
  }
  if (p.ksplit > 1) {  // K loop cut across workgroups: raw partial sums
    if (!p.partial || p.ablate != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, false, 0, 0, WM, true>), dim3(tiles_total, p.ksplit), block, 0,
                       stream, p, tiles_n, tiles_total);
    return hipGetLastError();
  }
#define ED_GO(OB, RS, AB) \
  hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, OB, RS, AB, 0, WM>), grid, block, 0, stream, p, tiles_n, tiles_total)
  if ((p.ablate == 1 || p.ablate == 2) && (p.out_bf16 || p.resid)) return hipErrorInvalidValue;  // (fp32 stores only)
  if (p.ablate == 1) ED_GO(false, false, 1);
  else if (p.ablate == 2) ED_GO(false, false, 2);
  else if (p.ablate == 7 && p.stamps) {
    if (p.resid && p.K == 20 * BK && p.resid_mul == 1 && !CONV)
      hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 3, 20, WM>), grid, block, 0, stream, p, tiles_n, tiles_total);
    else if (p.resid) ED_GO(false, true, 3);
    else ED_GO(true, false, 3);
  }
  else if (p.resid) {
    // hidden-residual form: 1x1, every tile full, K loop of a known length
    const bool whole = !CONV && ((long)p.B * p.L) % BMT == 0 && (p.B == 1 || p.L % BMT == 0) && p.resid_mul == 1;
    if (whole && p.K == 20 * BK)
      hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 0, 20, WM>), grid, block, 0, stream, p, tiles_n, tiles_total);
    else if (whole && NT == 4 && p.K == 8 * BK)
      hipLaunchKernelGGL((gemm_bf16_kernel<NT, CONV, false, true, 0, 8, WM>), grid, block, 0, stream, p, tiles_n, tiles_total);
    else ED_GO(false, true, 0);
  }
  else if (p.out_bf16) ED_GO(true, false, 0);
  else ED_GO(false, false, 0);
#undef ED_GO
  return hipGetLastError();
}

template <int NT, bool CONV>
hipError_t launch_nt(const GemmProblem& p, hipStream_t stream) {
  if (gemm_bf16_tile_rows(p) == 128) return launch_nt<NT, CONV, 2>(p, stream);
  return launch_nt<NT, CONV, 4>(p, stream);
}

}  // namespace

// Shape contract of the tuned kernel.  The caller guarantees the memory contract: A rows
// [a_row0 - halo, a_row0 + ceil(L/256)*256 + halo) of every sequence are readable, with halo =
// (ntaps/2)*dilation (causal problems, tap_center = ntaps - 1: (ntaps-1)*dilation rows in front only), and rows outside
// [0, L) hold zeros whenever ntaps > 1.
bool gemm_bf16_supported(const GemmProblem& p) {
  if (!p.a_bf16 || p.K % BK != 0 || p.K < BK || p.seq_rows > 0) return false;
  if (p.N % 320 != 0 && p.N % 256 != 0) return false;
  if ((long)p.B * p.L < BM) return false;               // less than one 256-row tile: generic kernel
  if ((long)p.K * 2 * 64 >= (1L << 31)) return false;   // per-lane DMA offsets are 32-bit
  if (p.ntaps > 1 && p.a_row0 < (long)p.center() * p.dilation) return false;
  if (p.resid && p.out_bf16) return false;
  return true;
}

hipError_t launch_gemm_bf16(const GemmProblem& p, hipStream_t stream) {
  if (!gemm_bf16_supported(p)) return hipErrorInvalidValue;
  if (p.ksplit < 1 || p.ksplit > 64) return hipErrorInvalidValue;
  // (a cut K loop always takes the per-tap form: its step order is the one the ranges are defined on)
  if (p.ksplit == 1 && (p.ablate == 0 || p.ablate >= 4) && conv_window_supported(p)) return launch_conv_window(p, stream);
  if (p.ntaps > 1) return p.N % 320 == 0 ? launch_nt<5, true>(p, stream) : launch_nt<4, true>(p, stream);
  return p.N % 320 == 0 ? launch_nt<5, false>(p, stream) : launch_nt<4, false>(p, stream);
}

}  // namespace ed
