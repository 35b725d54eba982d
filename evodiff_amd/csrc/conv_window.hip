// Dilated k=5 convolution with the activation window kept in LDS: the launchers, the window geometry and the kernel that
// wraps the tile body of conv_window_tile.hpp (described there).
#include "conv_window_tile.hpp"

namespace ed {

namespace {

using conv_tile::NTAPS;
using conv_tile::SEQ_ROWS;
using conv_tile::WIN_ROWS;
using conv_tile::WindowPlan;

template <int NT, bool OUT_BF16, bool STAMP = false, int WM = 4, bool LNF = false>
__global__ __launch_bounds__(WM * 128) void conv_window_kernel(GemmProblem g, int tiles_n, int tiles_total,
                                                               WindowPlan wp) {
  __shared__ __attribute__((aligned(1024))) char lds[conv_tile::lds_bytes(NT)];
  conv_tile::tile<NT, OUT_BF16, STAMP, WM, LNF>(g, tiles_n, tiles_total, wp, lds, (int)blockIdx.x);
}

}  // namespace

// Shape contract (on top of gemm_bf16_supported): five taps, no residual, and either
//  - the window of 256 + 4*dilation rows fits the 320-row LDS window (dilation <= 16), or
//  - whole-sequence form: L <= 256 (every row outside [0, 256) is zero padding) and dilation 32 or a multiple of
//    64 - then each 64-row wave slice of a tap is either entirely padding (skipped) or inside the window
//    [-wofs, 256 + wofs).
// Memory contract: A rows [a_row0 - 2*dilation, a_row0 + ceil(L/256)*256 + 2*dilation + 8) of every sequence are
// readable, and rows outside [0, L) hold zeros.
static bool window_plan(const GemmProblem& p, int wm, WindowPlan* wp) {
  if (!gemm_bf16_supported(p) || p.ntaps != NTAPS || p.resid) return false;
  const int d = p.dilation, bm = wm * 64, nw = 2 * wm;
  // (the same dilations whatever the tile height: the window forms add up slice-major, the per-tap form tap-major,
  // and a launch's results must not depend on which tile height its size selects)
  if (SEQ_ROWS + 4 * d <= WIN_ROWS) {
    const int groups = (bm + 4 * d + 7) / 8;
    *wp = WindowPlan{groups, p.center() * d, 0, NTAPS - 1, 0, (groups + nw * NTAPS - 1) / (nw * NTAPS), p.center()};
    return wp->na <= 3;
  }
  if (p.center() == NTAPS / 2 && p.L <= SEQ_ROWS && (d == 32 || d % 64 == 0)) {
    const int wofs = d == 32 ? 32 : 0;
    int lo = 0, hi = NTAPS - 1;
    while ((2 - lo) * d >= SEQ_ROWS) ++lo;
    while ((hi - 2) * d >= SEQ_ROWS) --hi;
    const int groups = (SEQ_ROWS + 2 * wofs) / 8, nt = hi - lo + 1;
    *wp = WindowPlan{groups, wofs, lo, hi, 1, (groups + nw * nt - 1) / (nw * nt), NTAPS / 2};
    return wp->na <= (wm == 4 ? 2 : 3);
  }
  return false;
}

// 128-row tiles when 256-row tiles would cover at most half of the CUs (same rule as gemm_bf16.hip)
static int window_wm(const GemmProblem& p, WindowPlan* wp) {
  if (gemm_bf16_tile_rows(p) == 128 && window_plan(p, 2, wp)) return 2;
  return window_plan(p, 4, wp) ? 4 : 0;
}

bool conv_window_supported(const GemmProblem& p) {
  WindowPlan wp;
  return window_wm(p, &wp) != 0;
}

hipError_t launch_conv_window(const GemmProblem& p, hipStream_t stream) {
  WindowPlan wp;
  const int wm = window_wm(p, &wp);
  if (!wm) return hipErrorInvalidValue;
  if (p.ln.gamma && (!gemm_bf16_ln_supported(p) || !p.ln.beta || !p.ln.out || !p.ln.stats || !p.ln.counters || !p.ln.error))
    return hipErrorInvalidValue;
#define ED_GO(NTV, OB, WMV)                                                                               \
  {                                                                                                       \
    const int tiles_n = p.N / (NTV * 64);                                                                 \
    const int tiles_total = p.B * ((p.L + WMV * 64 - 1) / (WMV * 64)) * tiles_n;                          \
    if (p.ln.gamma && p.ablate == 7 && p.stamps) {                                                        \
      if constexpr (NTV == 5 && WMV == 4)                                                                 \
        hipLaunchKernelGGL((conv_window_kernel<NTV, true, true, WMV, true>), dim3(tiles_total), dim3(WMV * 128), 0, \
                           stream, p, tiles_n, tiles_total, wp);                                          \
      else                                                                                                \
        return hipErrorInvalidValue;                                                                      \
    } else if (p.ln.gamma)                                                                                \
      hipLaunchKernelGGL((conv_window_kernel<NTV, true, false, WMV, true>), dim3(tiles_total), dim3(WMV * 128), 0, \
                         stream, p, tiles_n, tiles_total, wp);                                            \
    else if (p.ablate == 7 && p.stamps)                                                                   \
      hipLaunchKernelGGL((conv_window_kernel<NTV, OB, true, WMV>), dim3(tiles_total), dim3(WMV * 128), 0, \
                         stream, p, tiles_n, tiles_total, wp);                                            \
    else                                                                                                  \
      hipLaunchKernelGGL((conv_window_kernel<NTV, OB, false, WMV>), dim3(tiles_total), dim3(WMV * 128), 0, \
                         stream, p, tiles_n, tiles_total, wp);                                            \
  }
#define ED_GO2(NTV, OB) \
  if (wm == 2) ED_GO(NTV, OB, 2) else ED_GO(NTV, OB, 4)
  if (p.N % 320 == 0) {
    if (p.out_bf16) ED_GO2(5, true) else ED_GO2(5, false)
  } else {
    if (p.out_bf16) ED_GO2(4, true) else ED_GO2(4, false)
  }
#undef ED_GO2
#undef ED_GO
  return hipGetLastError();
}

}  // namespace ed
