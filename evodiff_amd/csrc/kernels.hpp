// Internal launcher interface between the C-ABI layer (api.hip) and the gfx950 kernels.
// All launchers are asynchronous on `stream`, allocate nothing and never synchronise, so a
// whole denoiser step can be captured into a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ed {

typedef unsigned short bf16_t;  // storage type of a bfloat16

__host__ __device__ inline bf16_t f32_to_bf16_bits(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  if ((v.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((v.u >> 16) | 0x40);  // keep NaN a NaN
  return (bf16_t)((v.u + 0x7fffu + ((v.u >> 16) & 1u)) >> 16);                // round to nearest even
}

// Device-resident per-call scalars read by kernels inside a captured graph.
struct StepState {
  int step;       // index of the current sampling step (advanced by advance_step_kernel, or by the step tail itself)
  int t_start;    // D3PM: t of step 0; the current t is t_start - step
  unsigned done;  // workgroups of the OA-DM step tail that have finished (the last one advances `step` and clears this)
};

// Device-side row selector (order == nullptr: always 0); lets row maps that depend on the OA-DM positions live
// inside a captured graph.
//  - bit field (maxstart < 0): bits [shift, shift + log2(mask + 1)) of order[step], the position the step fills;
//  - window (maxstart >= 0): first row of the window of rows that changed since the previous step, i.e. that lie
//    within `radius` of order[step - lag], clamped to [0, maxstart]: lag = 1 for OA-DM (the position the previous
//    step filled), lag = 0 for autoregressive steps (the position this step extends the sequence from).
struct PosBit {
  const int* order = nullptr;
  const StepState* state = nullptr;
  int shift = 0;
  int mask = 0;
  int radius = 0;
  int maxstart = -1;
  int lag = 1;
  __device__ __forceinline__ int get() const {
    if (!order) return 0;
    if (maxstart < 0) return (order[state->step] >> shift) & mask;
    const int s = order[state->step - lag] - radius;
    return s < 0 ? 0 : (s > maxstart ? maxstart : s);
  }
  static PosBit window(const int* order, const StepState* state, int radius, int maxstart, int lag = 1) {
    PosBit p;
    p.order = order; p.state = state; p.radius = radius; p.maxstart = maxstart; p.lag = lag;
    return p;
  }
};

// A pointer read from a structure in memory is a generic ("flat") pointer to the compiler; the tile bodies mark the
// device-memory pointers of a problem as global ones, so that loads and stores through them stay global_load /
// global_store however the problem reaches them.
#if defined(__HIPCC__)
template <typename T>
__device__ __forceinline__ T* as_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}
#endif

// LayerNorm + GELU of a GEMM's result fused into the tuned kernels' epilogue (ln_epilogue.hpp).  gamma == nullptr: off.
struct LnFuse {
  const float* gamma = nullptr;  // [N] of the LayerNorm that consumes this GEMM's result
  const float* beta = nullptr;
  bf16_t* out = nullptr;         // GELU(LayerNorm(result)) as bf16 rows of N channels; flat GEMM row R goes to row R, or with
  int seq_len = 0;               //   seq_len > 0 to row (R / seq_len) * out_pitch + out_row0 + R % seq_len (padded conv input)
  long out_pitch = 0, out_row0 = 0;
  void* stats = nullptr;         // exchange buffer: 8 bytes per (row tile, column tile, tile row)
  unsigned* counters = nullptr;  // one arrival counter per row tile; never reset (see ln_epilogue.hpp)
  unsigned* error = nullptr;     // host-visible word, set to 1 when a bounded wait ran out
  int dbg = 0;                   // diagnostics (ed_debug_gemm 400 + dbg; results invalid): 1 no wait for the partners, 2 no GELU,
                                 //   4 no stores of the normalised rows, 8 no statistics pass, 16 raw result stored in
                                 //   front of the wait (the older order; results valid)
  // The GEMM's own result is still stored to GemmProblem::out when it has a residual (PFF2: the fp32 residual stream);
  // otherwise only the normalised rows leave the kernel.
};

// One implicit-GEMM problem: out[b*L + l][n] = bias[n] (+ resid) + sum_{tap,k} A[row(b, l + off(tap))][k] * W[tap][n][k]
// with off(tap) = (tap - ntaps/2) * dilation and rows outside [0, L) contributing zero.
struct GemmProblem {
  const void* A;        // activations, row-major rows of K elements (fp32 or bf16)
  long a_seq_pitch;     // rows between the starts of consecutive sequences in A
  long a_row0;          // row of position 0 inside a sequence (halo padding in front)
  const void* W;        // [ntaps][N][K], same element type as A
  const float* bias;    // [N]
  void* out;            // [B*L][N], fp32 or bf16
  const float* resid;   // nullptr, or fp32 [B*L][N] added to the result (may alias out)
  int B, L, N, K, ntaps, dilation;
  bool a_bf16;          // element type of A and W
  bool out_bf16;        // element type of out (resid != nullptr requires fp32 out)
  int resid_mul = 1;    // the residual of output row R is row R * resid_mul + resid_sel.get() of `resid`
  PosBit resid_sel;     //   (resid_mul > 1: every resid_mul-th row of a longer buffer; then resid must not alias out)
  // Row groups (generic kernel and reduction only; the tuned kernels refuse such problems).  seq_rows > 0: the problem
  // is flat (B = 1) and its row R belongs to sequence b = R / seq_rows at offset l = R % seq_rows, with
  //   A row   = b * a_seq_pitch + a_row0 + a_sel + l (+ tap shift), read without bounds checks when a_unbounded
  //             (the rows around the window are real data or real zero padding),
  //   out row = b * out_pitch + out_sel + l,   resid row = b * resid_pitch + l * resid_mul + resid_sel.
  int seq_rows = 0;
  bool a_unbounded = false;
  PosBit a_sel, out_sel;
  long out_pitch = 0, resid_pitch = 0;
  int tile = 128;       // generic kernel only: 128 x 128 or 64 x 64 output tile per workgroup
  int ksplit = 1;       // generic kernel only: the K loop is cut into ksplit ranges (grid.z); range z writes its raw fp32
  float* partial = nullptr;  //   sums to partial[z][B*L][N] and launch_reduce_partials adds them up in a fixed order
  long long* stamps = nullptr;  // diagnostics (ablate == 7): eight clock stamps per tile (evodiff_hip.h, ed_debug_set_stamp_buffer)
  int ablate = 0;       // diagnostics of the tuned kernel: 1 = no MFMAs, 2 = no DMA copies (results are wrong), 3 = no window form
  int tap_center = -1;  // tap t reads position l + (t - center) * dilation; -1 = ntaps / 2 ('same' padding, MaskedConv1d),
                        //   ntaps - 1 = causal (left padding only, MaskedCausalConv1d)
  LnFuse ln;            // tuned kernels only: the LayerNorm behind this GEMM, applied in its epilogue
  __host__ __device__ int center() const { return tap_center >= 0 ? tap_center : ntaps / 2; }
};

// Reference-quality tile kernel: any shape with K % 32 == 0 (bf16) / K % 16 == 0 (fp32).
hipError_t launch_gemm_generic(const GemmProblem& p, hipStream_t stream);
// out = bias (+ resid through the problem's row map) + sum_z partial[z], for a problem run with ksplit > 1.
hipError_t launch_reduce_partials(const GemmProblem& p, hipStream_t stream);
// Tuned bf16 kernel (see gemm_bf16.hip for its shape requirements); false = shape not supported.
bool gemm_bf16_supported(const GemmProblem& p);
// Tile height the tuned kernels use for a problem: 256 rows, or 128 when 256-row tiles would cover at most half of
// the 256 CUs (ablate != 0, the diagnostic forms, always use 256).
inline long gemm_bf16_full_tiles(const GemmProblem& p) {
  return (long)p.B * ((p.L + 255) / 256) * (p.N % 320 == 0 ? p.N / 320 : p.N / 256);
}
inline int gemm_bf16_tile_rows(const GemmProblem& p) {
  return p.ablate == 0 && gemm_bf16_full_tiles(p) <= 128 ? 128 : 256;
}
hipError_t launch_gemm_bf16(const GemmProblem& p, hipStream_t stream);
// Whether the tuned kernels can apply p.ln themselves: the column tiles of a row tile must be a power of two (see
// ln_epilogue.hpp), no cut K loop, no diagnostic form.  The caller decides on the activation (GELU only).
inline bool gemm_bf16_ln_supported(const GemmProblem& p) {
  const int tn = p.N % 320 == 0 ? p.N / 320 : p.N / 256;
  return gemm_bf16_supported(p) && p.ksplit == 1 && (p.ablate == 0 || p.ablate == 7) && tn >= 1 && (tn & (tn - 1)) == 0;
}
// Window-in-LDS form of the dilated convolution (conv_window.hip): five taps, dilation <= 16; launch_gemm_bf16
// picks it by itself when the shape allows (GemmProblem::ablate == 3 keeps the per-tap form, for comparisons).
namespace conv_tile {
// Geometry of the window (host-computed, the same for every tile of a launch).
struct WindowPlan {
  int win_groups;  // 8-row copy groups of the window
  int wofs;        // window rows in front of the tile's first position (2d; whole-sequence form: 32 or 0)
  int tap_lo, tap_hi;  // taps in the K loop (whole-sequence form drops taps no position of the tile can reach)
  int clip;        // whole-sequence form: a wave skips the taps that move all of its 64 rows outside [0, 256);
                   // the window then starts at position -wofs of the SEQUENCE, whichever tile the workgroup owns
  int na;          // window groups each wave copies per step (1 .. 3)
  int center;      // tap t reads position l + (t - center) * dilation (2 = 'same' padding, 4 = causal)
};
}  // namespace conv_tile
bool conv_window_supported(const GemmProblem& p);

hipError_t launch_conv_window(const GemmProblem& p, hipStream_t stream);

// tokens (B*L) int64 -> X fp32 [B*L][D]:  up_b + up_w @ (emb[tok] + pe[t])   (model.py:131-140)
// tvec: int64 (B,) per-sequence timesteps with the reference's reshape quirk (row r uses tvec[r % B]),
// or nullptr to use state->t_start - state->step for every row.  pe == nullptr: no time encoding.
hipError_t launch_embed(const int64_t* tokens, const int64_t* tvec, const StepState* state, const float* emb,
                        const float* pe, int n_timesteps, const float* up_w, const float* up_b, float* X,
                        int rows, int B, int D, int E, int n_tokens, hipStream_t stream);

// X[r] = table[tokens[r]] for a table of n_tokens embedded rows (models without timestep encoding).
hipError_t launch_embed_gather(const int64_t* tokens, const float* table, float* X, int rows, int D, int n_tokens,
                               hipStream_t stream);

// X[b*L + sel] = table[tokens[b*L + sel]] for every sequence b: the one row per sequence that changed.
// compact: X holds ONE row per sequence (row b) instead of L (row b*L + position)
hipError_t launch_embed_gather_rows(const int64_t* tokens, const float* table, float* X, int B, int L, int D,
                                    int n_tokens, PosBit sel, hipStream_t stream, bool compact = false);

// Row maps of the LayerNorm kernel: logical row r = (b, m), m < rows_per_seq, is read from row
// in_row0 + b*in_pitch + m*in_mul + in_sel.get() and written to row out_row0 + b*out_pitch + m + out_sel.get().
struct LnRowMap {
  int rows_per_seq;
  long in_pitch, in_row0;
  int in_mul;
  PosBit in_sel;
  long out_pitch, out_row0;
  PosBit out_sel;  // added to the output row
  static LnRowMap plain(int rows) { return LnRowMap{rows, rows, 0, 1, PosBit{}, rows, 0, PosBit{}}; }
};
// out = act(LayerNorm(in)) row-wise; in, out fp32/bf16 rows of C channels.  act: 0 none, 1 exact GELU, 2 ReLU.
hipError_t launch_ln_act(const void* in, bool in_bf16, void* out, bool out_bf16, const float* gamma,
                         const float* beta, int rows, int C, const LnRowMap& map, int act, hipStream_t stream);

// The same LayerNorm (bf16 out) reading the result of a split-K GEMM `q` straight from its partial sums (q.partial,
// q.ksplit; launch_reduce_partials is then NOT run): in = bias + sum_z partial[z], rounded to bf16 if q.out_bf16.
// The map's input rows index q's flat output rows.
bool reduce_ln_supported(const GemmProblem& q);
hipError_t launch_reduce_ln(const GemmProblem& q, bf16_t* out, const float* gamma, const float* beta, int rows,
                            const LnRowMap& map, int act, hipStream_t stream);

// logits[r][k] = dec_b[k] + dec_w[k] . LayerNorm(X[r])  for all rows (ByteNetLMTime.forward tail).
hipError_t launch_final_nll(const float* X, const float* gamma, const float* beta, const float* dec_w,
                            const float* dec_b, const int64_t* targets, const uint8_t* select, float* nll, int rows,
                            int D, int n_classes, hipStream_t stream);
hipError_t launch_final_logits(const float* X, const float* gamma, const float* beta, const float* dec_w,
                               const float* dec_b, float* logits, int rows, int D, int n_tokens,
                               hipStream_t stream);

// OA-DM step tail (generate.py:198-211): one categorical draw per sequence at position pos = order[step].
// X holds x_rows_per_seq rows per sequence; the row of position pos is pos >> x_shift.
// n_draw_step (may be null): the draw of step s is over the first n_draw_step[s] classes (<= n_draw; probs rows keep
// n_draw entries, zero beyond).  keep (may be null): L bytes, a non-zero byte marks a position whose token is given -
// the step that would write it leaves it alone (generate_autoreg_scaffold's motif, conditional_generation.py:552-557).
hipError_t launch_final_oadm(const float* X, int x_rows_per_seq, int x_shift, const float* gamma, const float* beta,
                             const float* dec_w, const float* dec_b, int64_t* tokens, const int* order,
                             const float* uniforms, StepState* state, float* probs, int B, int L, int D,
                             int n_draw, hipStream_t stream, int write_shift = 0, bool advance = false,
                             const int* n_draw_step = nullptr, const uint8_t* keep = nullptr);
// Batches up to this size advance the step counter inside the OA-DM step tail (the workgroup that finishes last does
// it: every workgroup has read the counter by then) instead of a launch of its own; larger ones would queue on the
// ticket word for longer than that launch takes.
constexpr int ED_ADVANCE_IN_TAIL_MAX_B = 128;

// D3PM step tail (generate.py:279-298): softmax over K, fp64 posterior with Q[t] / Qbar[t-1], one draw per
// position, in place on tokens.
hipError_t launch_final_d3pm(const float* X, const float* gamma, const float* beta, const float* dec_w,
                             const float* dec_b, int64_t* tokens, const double* Q, const double* Qbar,
                             const float* uniforms, const StepState* state, double* post, int B, int L, int D,
                             int K, int n_timesteps, hipStream_t stream);

hipError_t launch_advance_step(StepState* state, hipStream_t stream);

// Weight-streaming GEMM on a few rows (ar_gemv.hip): out[b][n] = bias[n] (+ resid[b][n]) + sum_{t, k} W[t][n][k] * a[b][t][k]
// for b < B <= AR_GEMV_MAX_B, where a[b][ntaps-1] = act(LayerNorm(in[b])) * gamma + beta rounded to bf16 (the LayerNorm in
// front of the GEMM, applied by the consumer) and, for the causal convolution (ntaps = 5), a[b][t] for t < ntaps-1 are
// the rows pos - (ntaps-1-t) * dilation of the layer's cache of such rows, to which the new row is appended at pos.
constexpr int AR_GEMV_MAX_B = 8;
struct ArGemv {
  const float* in;       // [B] rows of C fp32 values, in_pitch elements apart: the raw input of the LayerNorm
  long in_pitch;
  const float *gamma, *beta;  // [C]
  int C, act;            // act: 0 none, 1 GELU (the bf16 path's fit), 2 ReLU
  const bf16_t* W;       // [ntaps][N][C]
  const float* bias;     // [N]
  const float* resid;    // nullptr, or fp32 [B][N] (may alias out: an element is read and written by the same lane)
  float* out;            // fp32 [B][N]
  int B, N, ntaps, dilation;
  bf16_t* cache;         // ntaps > 1: [B][cache_pitch][C] bf16, position 0 at row cache_row0 (zero rows in front)
  long cache_pitch, cache_row0;
  PosBit pos;            // position of the new row
  int ncol = 0;          // (set by the launcher: output columns of a workgroup and pass)
};
bool ar_gemv_supported(const ArGemv& p);
hipError_t launch_ar_gemv(const ArGemv& p, int n_cus, hipStream_t stream);

// Asynchronous error words (host-visible memory, one 4-byte word per code): kernels raise them instead of the host
// validating device data with a copy and a synchronisation per call; the C ABI reports them on the next call.
enum { ED_AERR_LN_WAIT = 0, ED_AERR_TOKEN = 1, ED_AERR_TIMESTEP = 2, ED_AERR_ORDER = 3, ED_AERR_TARGET = 4, ED_AERR_WORDS = 16 };

// dst[i] = src[i] clamped into [0, hi) (src == nullptr: dst[i] = i), raising err[code] when a value was outside.
hipError_t launch_prepare_order(const int* src, int* dst, int n, int hi, unsigned* err, hipStream_t stream);
// dst[i] = src[i], values outside [0, hi) replaced by hi - 1 and reported in err[code] (tokens, timesteps)
hipError_t launch_prepare_ids(const int64_t* src, int64_t* dst, long n, int hi, unsigned* err, int code, hipStream_t stream);
// raises err[ED_AERR_TARGET] when a selected target is outside [0, n_classes) (the scoring tail itself clamps)
hipError_t launch_check_targets(const int64_t* targets, const uint8_t* select, long n, int n_classes, unsigned* err,
                                hipStream_t stream);
// *state = {step, t_start}
hipError_t launch_init_state(StepState* state, int step, int t_start, hipStream_t stream);

}  // namespace ed
