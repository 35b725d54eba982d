// In-launch hand-off between the phases of a block chain (block_chain.hip): PFF1 -> convolution -> PFF2 of one or more
// ByteNet blocks run as ONE persistent launch, each workgroup keeping its (row tile, column tile) through all phases.
//
// MEASURED OUTCOME (round 2, config 2, same-process A/B `tools/opt_ab.py block_chain 0 1 2 3`): bit-identical results,
// and NOT faster - 15.40 ms per diffusion step with one launch per GEMM, 15.56 / 15.95 / 16.06 with chains of 1 / 2 / 3
// blocks.  Timing-only builds with parts of the hand-off removed (results invalid) show where it goes: without the
// consumer's acquire 15.31 against 15.74, without any wait 15.24, with plain instead of write-through stores (and the
// acquire) 15.60, with plain stores + the producer's agent release 15.87.  A hand-off in one of the forms that are
// valid for any placement costs as much as the launch boundary it replaces; the forms that would be cheaper rely on
// the workgroups of a row tile sharing an XCD, which the programming guide rules out.  ED_OPT_BLOCK_CHAIN therefore
// defaults to 0; the code stays as the measured form of the idea and as a second user of the tile bodies.
//
// Why it was tried: between two launches a CU sits idle from the exit of its workgroup to the first MFMA of the next one - 3.5-4.5 us
// until the next launch's workgroup enters (dispatch, LDS and register allocation; measured with in-kernel clocks,
// tools/kernel_timeline.py; two streams feeding the CUs without any ordering do not shorten it), ~3 us of prologue
// until the first operands are in LDS, and the launch ends with its slowest XCD.  A phase change inside a launch costs
// the hand-off below instead.
//
// What is handed over: a phase's A operand rows of a row tile are the normalised rows the tiles_n workgroups of that
// SAME row tile stored in the phase before (1x1 projections are row-local; the convolution is when a row tile is a whole
// sequence, which is what the launcher checks).  Protocol (cdna_hip_programming.md Guideline 16; MI355X_MICROARCH.md
// "Valid forms": write-through stores, every storing wave drained, one signalling lane behind a workgroup barrier;
// consumer: one relaxed poll, ONE agent acquire, wait, barrier, then plain loads / LDS-DMA copies):
//   producer: the normalised rows leave as write-through (sc1) 16-byte stores; every wave `s_waitcnt vmcnt(0)`;
//             workgroup barrier; ONE lane adds 1 to the row tile's chain counter (agent-scope atomic);
//   consumer: ONE lane polls that counter (relaxed, agent scope, bounded) until phase_index * tiles_n arrivals are in;
//             its wave runs the agent acquire (buffer_inv sc1: this CU's L1 holds lines of the same buffers from the
//             block before) and waits for it; workgroup barrier; only then the phase begins.  (The wait is the very
//             first thing a phase does: behind the first weight copies, which depend on nothing, it would hide ~1 us
//             more, but a barrier between a phase's address set-up and its unrolled K loop costs 21 spilled registers.)
// Counters are zero between launches: the arrival that completes the last phase of a launch stores 0 (every wait on the
// counter has been satisfied by then).  Nothing is assumed about dispatch order or workgroup -> XCD placement; what IS
// assumed is that all workgroups of the launch are resident at once (grid <= number of CUs, one workgroup per CU by its
// LDS), which the launcher checks - a workgroup that waits keeps its CU.
#pragma once
#include "kernels.hpp"

namespace ed {

struct ChainCtl {
  unsigned* ctr = nullptr;     // one counter per row tile
  unsigned* error = nullptr;   // host-visible word (ED_AERR_LN_WAIT slot), set when a bounded wait runs out
  unsigned wait_target = 0;    // arrivals that must be in before the phase's A operand may be read (0: none)
  unsigned final_total = 0;    // > 0: this is the launch's last phase; the arrival that makes the counter this resets it
};

constexpr unsigned long long CHAIN_SPIN_TICKS = 5000000ull;  // 50 ms of the 100-MHz clock

// Called by ALL threads of the workgroup as the first thing of a phase.
__device__ __forceinline__ void chain_wait(const ChainCtl& cc, int tile_m, int tid) {
  if (cc.wait_target == 0) return;  // (uniform)
  typedef __attribute__((address_space(1))) unsigned int gu32;
  if (tid < 64) {
    if (tid == 0) {
      gu32* const c = (gu32*)cc.ctr + tile_m;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while ((int)(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - cc.wait_target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > CHAIN_SPIN_TICKS) {  // a partner never arrived: report, do not hang
          __hip_atomic_store((gu32*)cc.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the polling wave, after its poll matched
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the invalidate completes asynchronously)
  }
  __syncthreads();
}

// Called by ALL threads after the phase's last store.
__device__ __forceinline__ void chain_signal(const ChainCtl& cc, int tile_m, int tid) {
  typedef __attribute__((address_space(1))) unsigned int gu32;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier the signalling lane joins
  __syncthreads();
  if (tid == 0) {
    gu32* const c = (gu32*)cc.ctr + tile_m;
    const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cc.final_total && old + 1 == cc.final_total) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// 16-byte write-through (sc1) stores of the chain's hand-off data, as buffer stores off a resource built once per
// epilogue: base pointer in SGPRs, 32-bit byte offset per lane.  (A hand-written `global_store_dwordx4 ... sc1` in
// inline asm stored wrong rows here - the compiler's hazard handling does not see inside it; the builtin is tracked like
// any other store.)
struct WtBuffer {
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ WtBuffer wt_buffer(void* base) {
  // raw buffer (stride 0), no bounds (the callers' offsets are in range by construction); dword 3 as for gfx90a / gfx94x / gfx950
  return WtBuffer{__builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000)};
}
template <typename V>
__device__ __forceinline__ void store16_wt(const WtBuffer& b, unsigned byte_offset, V v) {
  static_assert(sizeof(V) == 16, "16-byte vector");
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b.rsrc, (int)byte_offset, 0, /*aux: sc1*/ 16);
}

}  // namespace ed
