// Block chain: PFF1 -> dilated convolution -> PFF2 of up to CHAIN_MAX_BLOCKS ByteNet blocks (evodiff/model.py:142-147 +
// sequence_models ByteNetBlock) as ONE persistent launch on gfx950.  Every workgroup keeps its (row tile, column tile)
// through all phases and runs the tile bodies of the tuned kernels (gemm_bf16_tile.hpp, conv_window_tile.hpp) one after
// the other in the same LDS; each phase applies the LayerNorm + GELU behind it in its epilogue (ln_epilogue.hpp) and
// hands the normalised rows to the next phase inside the launch (chain.hpp: why, and the hand-off protocol).
#include "conv_window_tile.hpp"
#include "gemm_bf16_tile.hpp"

#include <cstddef>

namespace ed {

namespace {

struct ChainArgs {
  ChainBlock blk[CHAIN_MAX_BLOCKS];
  int n_blocks;
  int tiles_n, tiles_total;
  unsigned* ctr;
  unsigned* error;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// A reference to a member of the kernel's argument block, formed from the kernarg segment pointer behind an empty asm:
// loads through it cannot move above this point.  (Read straight from the by-value parameter, every field of every
// phase's problem is loaded at kernel entry and kept in SGPRs across all phases: 86 - 329 SGPR spills measured; and an
// argument block indexed at run time is copied to private memory, where its pointers stop being wave-uniform.)
typedef const __attribute__((address_space(4))) char* kernarg_ptr;
template <typename T>
__device__ __forceinline__ const T& kernarg_ref(unsigned offset) {
  kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offset;
  asm volatile("" : "+s"(p));
  return *(const T*)p;
}

// One block = three phases.
template <int NT, int NB, int K>
__device__ __forceinline__ void run_block(char* lds) {
  constexpr unsigned BLK = (unsigned)offsetof(ChainArgs, blk) + (unsigned)K * (unsigned)sizeof(ChainBlock);
  const int bid = (int)blockIdx.x;
  auto ctl = [](unsigned phase, bool last) {
    const ChainArgs& a = kernarg_ref<ChainArgs>(0);
    ChainCtl cc;
    cc.ctr = a.ctr;
    cc.error = a.error;
    cc.wait_target = phase * (unsigned)a.tiles_n;
    cc.final_total = last ? 3u * NB * (unsigned)a.tiles_n : 0u;
    return cc;
  };
  {  // phase 3K: PFF1 (+ LayerNorm 2); its input was written by the launch before (K = 0) or by PFF2 of block K - 1
    const ChainArgs& a = kernarg_ref<ChainArgs>(0);
    gemm_tile::tile<NT, false, true, false, 0, 0, 4, false, true, true>(
        kernarg_ref<GemmProblem>(BLK + (unsigned)offsetof(ChainBlock, g1)), a.tiles_n, a.tiles_total, lds, bid, 0, ctl(3u * K, false));
  }
  {  // phase 3K + 1: convolution (+ LayerNorm 3)
    const ChainArgs& a = kernarg_ref<ChainArgs>(0);
    conv_tile::tile<NT, true, false, 4, true, true>(kernarg_ref<GemmProblem>(BLK + (unsigned)offsetof(ChainBlock, gc)), a.tiles_n,
                                                    a.tiles_total,
                                                    kernarg_ref<conv_tile::WindowPlan>(BLK + (unsigned)offsetof(ChainBlock, wp)), lds,
                                                    bid, ctl(3u * K + 1u, false));
  }
  {  // phase 3K + 2: PFF2 + residual (+ LayerNorm 1 of the next block)
    const ChainArgs& a = kernarg_ref<ChainArgs>(0);
    gemm_tile::tile<NT, false, false, true, 0, 20, 4, false, true, true>(
        kernarg_ref<GemmProblem>(BLK + (unsigned)offsetof(ChainBlock, g2)), a.tiles_n, a.tiles_total, lds, bid, 0,
        ctl(3u * K + 2u, K + 1 == NB));
  }
  if constexpr (K + 1 < NB) run_block<NT, NB, K + 1>(lds);
}

template <int NT, int NB>
__global__ __launch_bounds__(512) void block_chain_kernel(ChainArgs a) {
  __shared__ __attribute__((aligned(1024))) char lds[cmax(gemm_tile::lds_bytes(NT, 4), conv_tile::lds_bytes(NT))];
  run_block<NT, NB, 0>(lds);
}

bool ln_complete(const GemmProblem& p) {
  return p.ln.gamma && p.ln.beta && p.ln.out && p.ln.stats && p.ln.counters && p.ln.error;
}

}  // namespace

bool block_chain_supported(const ChainBlock& b, int n_cus) {
  const GemmProblem &g1 = b.g1, &gc = b.gc, &g2 = b.g2;
  // all three with the LayerNorm behind them in the epilogue, on 256 x 320 tiles with the same tile grid
  if (!ln_complete(g1) || !ln_complete(gc) || !ln_complete(g2)) return false;
  if (!gemm_bf16_ln_supported(g1) || !gemm_bf16_ln_supported(gc) || !gemm_bf16_ln_supported(g2)) return false;
  if (g1.ablate || gc.ablate || g2.ablate) return false;
  if (g1.N % 320 || gc.N != g1.N || g2.N != g1.N) return false;
  if (g1.ntaps != 1 || g2.ntaps != 1 || gc.ntaps != 5 || g1.resid || gc.resid || !g2.resid) return false;
  // a row tile is a whole sequence of exactly 256 positions: the convolution then reads only rows its own row tile's
  // workgroups produced, and the 1x1 projections (one flat sequence of B * 256 rows) tile the same way
  if (gc.L != 256 || gc.B < 1) return false;
  const long rows = (long)gc.B * 256;
  if (g1.B != 1 || g2.B != 1 || g1.L != rows || g2.L != rows) return false;
  // PFF2 in its unrolled form with the residual streamed inside the K loop; write-through offsets are 32-bit
  if (g2.K != 20 * 64 || g2.resid_mul != 1 || g2.out_bf16) return false;
  if ((long)gc.B * (gc.ln.out_pitch > 0 ? gc.ln.out_pitch : 256) * gc.N * 2 >= (1L << 31) ||
      (long)g1.B * g1.L * g1.N * 2 >= (1L << 31) || (g1.ln.seq_len > 0 && (g1.L / g1.ln.seq_len + 1) * g1.ln.out_pitch * g1.N * 2 >= (1L << 31)))
    return false;
  if (gemm_bf16_tile_rows(g1) != 256 || gemm_bf16_tile_rows(gc) != 256 || gemm_bf16_tile_rows(g2) != 256) return false;
  conv_tile::WindowPlan wp;
  if (!conv_tile::plan_256(gc, &wp)) return false;
  // every workgroup of the launch resident at once: one per CU (a workgroup that waits keeps its CU)
  const long tiles = gemm_bf16_full_tiles(g1);
  return tiles == gemm_bf16_full_tiles(gc) && tiles == gemm_bf16_full_tiles(g2) && tiles <= n_cus;
}

hipError_t launch_block_chain(ChainBlock* blocks, int n_blocks, unsigned* ctr, unsigned* error, hipStream_t stream) {
  if (!blocks || n_blocks < 1 || n_blocks > CHAIN_MAX_BLOCKS || !ctr || !error) return hipErrorInvalidValue;
  static_assert(sizeof(ChainArgs) <= 4000, "kernel arguments are limited to 4 KiB");
  ChainArgs a;
  for (int k = 0; k < n_blocks; ++k) {
    if (!block_chain_supported(blocks[k], 1 << 30)) return hipErrorInvalidValue;
    if (!conv_tile::plan_256(blocks[k].gc, &blocks[k].wp)) return hipErrorInvalidValue;
    if (gemm_bf16_full_tiles(blocks[k].g1) != gemm_bf16_full_tiles(blocks[0].g1)) return hipErrorInvalidValue;
    a.blk[k] = blocks[k];
  }
  a.n_blocks = n_blocks;
  a.tiles_n = blocks[0].g1.N / 320;
  a.tiles_total = (int)gemm_bf16_full_tiles(blocks[0].g1);
  a.ctr = ctr;
  a.error = error;
  if (n_blocks == 1) hipLaunchKernelGGL((block_chain_kernel<5, 1>), dim3(a.tiles_total), dim3(512), 0, stream, a);
  else if (n_blocks == 2) hipLaunchKernelGGL((block_chain_kernel<5, 2>), dim3(a.tiles_total), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((block_chain_kernel<5, 3>), dim3(a.tiles_total), dim3(512), 0, stream, a);
  return hipGetLastError();
}

}  // namespace ed
