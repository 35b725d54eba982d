/*
 * evodiff_hip.h — C ABI of libevodiff_hip.so: the MI355X (gfx950) implementation of EvoDiff's
 * ByteNetLMTime denoiser forward and per-step samplers.
 *
 * The reference (microsoft/evodiff) has no FFI: this path sits behind plain Python
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces; the
 * ctypes binding a maintainer would add on the reference side is shown in INTEGRATION.md.
 *
 * Conventions
 *   - Every function returns 0 on success and a negative ED_E* code on failure; the message for
 *     the calling thread's last failure is returned by ed_last_error().
 *   - Pointers named *_dev are device (HBM) pointers on the model's device; *_host are host
 *     pointers.  Buffers are owned by the caller; the model owns its packed weights, workspace
 *     and captured graphs.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous
 *     with respect to the host unless stated otherwise.
 *   - One model handle serves one device; a handle is not thread-safe (the reference is
 *     single-threaded on this path as well).
 *   - No CPU fallback exists anywhere in this library.
 *   - Device data is validated on the device, never by a copy to the host: token ids, timesteps, OA-DM positions and
 *     scoring targets outside their range are CLAMPED by the kernels (nothing ever indexes outside a table or a
 *     sequence) and reported asynchronously - the NEXT call on the model (or ed_model_check_errors) returns ED_EINVAL
 *     with a message naming what was out of range; the results of the offending call are not valid.  No compute entry
 *     point synchronises the stream.
 */
#ifndef EVODIFF_HIP_H
#define EVODIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ED_ABI_VERSION 1

enum {
  ED_OK = 0,
  ED_EINVAL = -1,     /* bad argument / shape */
  ED_ESTATE = -2,     /* call order (e.g. forward before commit, missing tensor) */
  ED_EHIP = -3,       /* a HIP runtime call failed */
  ED_ENOMEM = -4
};

enum { ED_PREC_FP32 = 0, ED_PREC_BF16 = 1 };

/* Hyper-parameters of ByteNetLMTime as constructed by evodiff/pretrained.py:39-42 from
 * config/config{38M,640M}.json (n_tokens=31, final_ln=True, tie_weights=False, causal=False). */
typedef struct ed_config {
  int32_t n_tokens;     /* len(MSA_ALPHABET) = 31 */
  int32_t d_embed;      /* 8 */
  int32_t d_model;      /* 1024 / 1280 */
  int32_t d_hidden;     /* d_model/2 if slim else d_model (evodiff/model.py:110-112) */
  int32_t n_layers;     /* 16 / 56 */
  int32_t kernel_size;  /* 5 */
  int32_t r;            /* 128: dilation of layer n is 2**(n % (log2(r)+1)) (model.py:108-109) */
  int32_t timesteps;    /* 0 = no timestep encoding (OA-DM); 500 for D3PM (model.py:93-94,133) */
  int32_t activation;   /* 1 = GELU (the only activation on this path), 0 = ReLU.  ED_PREC_FP32 evaluates the exact-erf
                           GELU (erff); ED_PREC_BF16 evaluates x * sigmoid(x (a + b x^2 + c x^4)), a fit of the exact-erf
                           GELU with at most 2.6e-5 absolute error, ahead of the rounding to bf16 (2^-9 relative) */
  int32_t causal;       /* 0 = MaskedConv1d ('same' padding); 1 = MaskedCausalConv1d, (k-1)*dilation zeros in front
                           only (ByteNetBlock(causal=True), model.py:114: the LR_AR_* models) */
} ed_config;

typedef struct ed_model ed_model;

/* Per-kernel-class probe record (see ed_probe_*): device time by HIP events around launches. */
/* ED_K_CONV: launches of the windowed convolution kernel (conv_window.hip, dilation <= 16), the dominant kernel;
 * ED_K_CONV_OTHER: convolutions that went to the per-tap or the small-tile kernel. */
enum { ED_K_EMBED = 0, ED_K_LNACT = 1, ED_K_GEMM1X1 = 2, ED_K_CONV = 3, ED_K_FINAL = 4, ED_K_CONV_OTHER = 5,
       ED_K_COUNT = 6 };
typedef struct ed_probe_stat {
  double ms;          /* summed device time of the probed launches */
  double flops;       /* summed algorithmic FLOPs (in-range taps only) of those launches */
  double bytes;       /* summed algorithmic HBM bytes (operands read once, results written once) */
  int64_t launches;
} ed_probe_stat;

int ed_abi_version(void);
const char* ed_last_error(void);

/* Replaces: ByteNetLMTime.__init__ (evodiff/model.py:152-168).  Selects `device`. */
int ed_model_create(const ed_config* cfg, int device, ed_model** out);
void ed_model_destroy(ed_model* m);

/* Replaces: nn.Module.load_state_dict as used by load_sequence_checkpoint
 * (evodiff/pretrained.py:47-52).  `key` is a reference state_dict key without the "module."
 * prefix, e.g. "embedder.layers.3.conv.weight"; `host` is contiguous fp32 in
 * the PyTorch layout; `numel` must match.  The extra key "time_encoding.table" (timesteps x
 * d_embed) carries the sin/cos table of PositionalEncoding1D (model.py:29-33), which the
 * reference rebuilds on every call. */
int ed_model_load_tensor(ed_model* m, const char* key, const float* host, int64_t numel);
/* Checks that every tensor is present and packs the device weights (fp32 and bf16 images). */
int ed_model_commit(ed_model* m);
/* Number of parameters packed (for cross-checks against 37,890,327 / 642,951,959). */
int64_t ed_model_num_params(const ed_model* m);
/* The hyper-parameters the handle was created with (the attributes a caller of the reference reads off the module:
 * model.embedder.timesteps, decoder width, ...): lets a binding size and check its buffers against the model instead of
 * trusting its own caller (csrc/torch_ops.cpp). */
int ed_model_get_config(const ed_model* m, ed_config* out);

/* Replaces: ByteNetLMTime.forward(x, y, input_mask=None) (evodiff/model.py:170-173).
 * tokens_dev int64 (B, L); timesteps_dev int64 (B,) (ignored when cfg.timesteps == 0);
 * logits_dev fp32 (B, L, n_tokens).
 * Token ids and timesteps are range-checked on the device: an id outside [0, n_tokens) is embedded as id n_tokens-1, a
 * timestep outside [0, timesteps) as timesteps-1, and the next call reports it (see Conventions).  The Python layer
 * (evodiff_amd.model) checks the ranges up front (one fused check) and raises IndexError like nn.Embedding would. */
int ed_forward(ed_model* m, const int64_t* tokens_dev, const int64_t* timesteps_dev, int B, int L,
               float* logits_dev, int precision, void* stream);
/* Debug/validation variant: additionally copies the residual stream after the embedding and after
 * every block into acts_dev, fp32 (n_layers + 1, B, L, d_model). */
int ed_forward_collect(ed_model* m, const int64_t* tokens_dev, const int64_t* timesteps_dev, int B, int L,
                       float* logits_dev, float* acts_dev, int precision, void* stream);

/* Replaces: the model call and the loss of sum_nll_mask (analysis/sequence_perp.py:125-154): one forward, then per
 * position the cross entropy of the first n_classes logits against targets[b][l] (torch CrossEntropyLoss as used
 * by OAMaskedCrossEntropyLoss, evodiff/losses.py:57-61, and by the D3PM branch, sequence_perp.py:133-134); the
 * logits never leave the GPU.  select_dev (B*L bytes, may be NULL = every position): positions with a zero byte
 * are skipped and score 0 (the OA-DM mask).  nll_dev receives B*L fp32 values; the caller sums them.
 * Targets of selected positions must be in [0, n_classes): others score against class 0 and are reported by the next
 * call (see Conventions). */
int ed_score_nll(ed_model* m, const int64_t* tokens_dev, const int64_t* timesteps_dev, const int64_t* targets_dev,
                 const uint8_t* select_dev, int B, int L, int n_classes, float* nll_dev, int precision, void* stream);

/* Replaces: the loop of generate_oaardm (evodiff/generate.py:193-211, penalty=None).
 * tokens_dev int64 (B, L), in/out (the caller fills it with mask_id for unconditional sampling);
 * order_dev int32 (n_steps,) = positions to fill (np.random.shuffle'd arange in the reference);
 * uniforms_dev fp32 (n_steps, B) in [0,1): the categorical draw at step s, sequence b is the
 * inverse CDF of softmax(logits[b, order[s], :n_draw]) at uniforms[s, b] (torch.multinomial in
 * the reference; north_star fixes pre-drawn uniforms on both sides).  n_draw = 21 in the
 * reference (generate.py:198).  probs_dev: optional fp32 (n_steps, B, n_draw) receiving the
 * softmax rows (NULL to skip).  use_graph != 0 replays one captured hipGraph per step. */
int ed_oadm_sample(ed_model* m, int64_t* tokens_dev, const int32_t* order_dev, const float* uniforms_dev,
                   int B, int L, int n_steps, int n_draw, float* probs_dev, int precision, int use_graph,
                   void* stream);

/* Replaces: the per-sequence loop of generate_autoreg (evodiff/generate.py:229-250) for a causal model (LR_AR_*):
 * tokens_dev (B, L) holds the START token at position 0 of every row; step s = 0 .. n_steps-1 takes the logits at
 * position s (softmax over the first n_draw of them, generate.py:241-243 uses all), draws token s+1 per sequence by
 * the inverse CDF at uniforms_dev[s][b] and writes it to position s+1 (n_steps <= L-1).  Where the reference runs the
 * model again on the whole growing prefix at every step, this keeps each layer's convolution input of the past
 * positions (they cannot change: every tap looks back) and computes ONE new row per sequence, layer and step.
 * The loop does not stop at the STOP token: what follows it does not influence what precedes it, so the caller cuts
 * the sequence there (evodiff_amd/generate.py::generate_autoreg).  probs_dev (n_steps, B, n_draw) may be NULL.
 * ED_ESTATE unless cfg.causal = 1 and cfg.timesteps = 0. */
int ed_ar_sample(ed_model* m, int64_t* tokens_dev, const float* uniforms_dev, int B, int L, int n_steps, int n_draw,
                 float* probs_dev, int precision, int use_graph, void* stream);

/* Replaces: the loop of generate_autoreg_scaffold (evodiff/conditional_generation.py:545-572): ed_ar_sample with a motif
 * spliced into the growing sequence.  given_dev: L bytes, a non-zero byte at position l means tokens_dev[b][l] is given
 * (the motif, and START at position 0): the step that would draw it runs - the caches of every layer need its row - and
 * its draw is dropped.  n_draw_step_dev int32 (n_steps,): step s draws over the first n_draw_step[s] classes (the
 * reference excludes STOP and the specials until the last motif residue is placed, :549-550,562), clamped into
 * [1, n_draw_max]; probs_dev rows hold n_draw_max entries, zero beyond the step's range.  Otherwise as ed_ar_sample. */
int ed_ar_sample_scaffold(ed_model* m, int64_t* tokens_dev, const float* uniforms_dev, const uint8_t* given_dev,
                          const int32_t* n_draw_step_dev, int B, int L, int n_steps, int n_draw_max, float* probs_dev,
                          int precision, int use_graph, void* stream);

/* Replaces: the loop of generate_d3pm (evodiff/generate.py:274-298).
 * tokens_dev int64 (B, L) in/out = x_t on entry; Q_dev / Qbar_dev fp64 (timesteps, K, K) as returned by
 * Tokenizer.q_blosum_schedule / q_random_schedule (evodiff/utils.py:216-249); runs n_steps steps
 * t = t_start, t_start-1, ... (the reference runs t = timesteps-1 ... 1).  uniforms_dev fp32
 * (timesteps, B, L): the step at time t consumes row timesteps-1-t; at t == 1 the reference's second,
 * restricted draw over the first K-6 categories (generate.py:293-294) consumes row timesteps-1.
 * post_dev: optional fp64 (n_steps, B, L, K) receiving the normalised posterior (NULL to skip). */
int ed_d3pm_sample(ed_model* m, int64_t* tokens_dev, const double* Q_dev, const double* Qbar_dev,
                   const float* uniforms_dev, int B, int L, int K, int t_start, int n_steps,
                   double* post_dev, int precision, int use_graph, void* stream);

/* Reports (and clears) what kernels of earlier calls on the model raised asynchronously: out-of-range tokens,
 * timesteps, positions or targets (ED_EINVAL), or a fused-LayerNorm exchange that timed out (ED_EHIP); ED_OK if
 * nothing is pending.  synchronize != 0 waits for `stream` first, so that everything launched so far is covered.
 * No counterpart in the reference (its errors are Python exceptions raised by the indexing op itself). */
int ed_model_check_errors(ed_model* m, void* stream, int synchronize);

/* Execution options (no counterpart in the reference).  They select kernels and what is recomputed, never the
 * function: with ED_PREC_FP32 the results are bit-identical under every setting; with ED_PREC_BF16 different settings
 * select kernels with different (fixed) orders of rounding, and results agree within the bf16 tolerance.
 *   ED_OPT_PRUNE (default 1): ed_oadm_sample reads the network output at ONE position per step
 *     (generate.py:198), so the trailing blocks whose dilations double from 1 are evaluated only on
 *     the positions that output depends on ("dependency-cone pruning").  0 = every block
 *     on every position.  ed_forward and ed_d3pm_sample always evaluate every position.
 *   ED_OPT_INCREMENTAL (default 1): between two steps of ed_oadm_sample one token per sequence changes, so
 *     the leading blocks (dilations 1, 2, 4 at config 2) keep their outputs from the previous step and recompute
 *     only the rows within their receptive field of the changed position.  0 = recompute every row.
 *   ED_OPT_MIN_TUNED_TILES (default 130): GEMMs with fewer tiles than this (tiles of the height the tuned kernel
 *     would pick: 128 rows up to 128 tiles of 256 rows, 256 rows above) use the small-tile kernel instead of the
 *     tuned one (0 = tuned kernel whenever the shape allows).
 *   ED_OPT_SPLIT_TARGET (default 1024): bf16 GEMMs on the small-tile kernel with fewer workgroups than this
 *     have their K loop cut across workgroups, up to this many (at most ~4 per tile when there are few tiles, and 16
 *     ways); partial sums are added in a fixed order; 0 = never.
 *   ED_OPT_FUSE_REDUCE_LN (default 1): the LayerNorm behind such a GEMM adds up the partial sums itself (one
 *     launch instead of reduction + LayerNorm; same order of additions).  0 = separate reduction launch.
 *   ED_OPT_SPLIT_MIN_ITERS (default 4): K-steps of 64 channels every workgroup of a cut K loop keeps at least.
 *   ED_OPT_TILE64_MAX (default 320): bf16 GEMMs on the small-tile kernel with at most this many 128 x 128 tiles use
 *     64 x 64 tiles (four times the workgroups before any K loop is cut); 0 = never.
 *   ED_OPT_TUNED_SPLIT_MIN_ROWS (default 0): bf16 GEMMs of at least this many rows with fewer tuned tiles than
 *     ED_OPT_MIN_TUNED_TILES run the tuned kernel (128-row tiles) with its K loop cut across workgroups; 0 = never.
 *   ED_OPT_PARTIALS_MB (default 32): size in MiB of the partial-sum workspace (a cut needs ksplit x rows x N x 4 B).
 *   ED_OPT_FUSE_LN (default 1): bf16 GEMMs on the tuned kernels apply the LayerNorm + GELU that follows them (PFF1 -> LN2,
 *     convolution -> LN3, PFF2 -> LN1 of the next block) in their own epilogue - the column tiles of a row tile exchange
 *     per-row partial statistics inside the launch - instead of a separate LayerNorm launch; the statistics are then
 *     taken on the fp32 sums rather than on their bf16 roundings.  0 = separate launches.  If a workgroup ever gives up
 *     waiting for its partners (bounded wait), the NEXT call on the model returns ED_EHIP.
 *   ED_OPT_AR_GEMV (default 8, range 0..8): ed_ar_sample / ed_ar_sample_scaffold with at most this many sequences (bf16,
 *     kernel size 5) run every block as three weight-streaming launches, each applying the LayerNorm in front of its
 *     GEMM itself (no K loop cut across workgroups, no partial sums in memory); 0 = always the tile kernels, as larger
 *     batches use.
 *   ED_OPT_AR_GEMV_WGS (default 0 = one per compute unit): the most workgroups a weight-streaming launch uses; with fewer
 *     than its column groups a workgroup works through several groups in passes (the tests use it to reach that path,
 *     which the widths the kernels accept never reach on 256 CUs).
 *   (Option 10, the round-2 block chain, was measured slower than one launch per GEMM and is gone; the number stays
 *   unused.) */
enum { ED_OPT_PRUNE = 0, ED_OPT_MIN_TUNED_TILES = 1, ED_OPT_SPLIT_TARGET = 2, ED_OPT_INCREMENTAL = 3,
       ED_OPT_FUSE_REDUCE_LN = 4, ED_OPT_SPLIT_MIN_ITERS = 5, ED_OPT_PARTIALS_MB = 6, ED_OPT_TILE64_MAX = 7,
       ED_OPT_TUNED_SPLIT_MIN_ROWS = 8, ED_OPT_FUSE_LN = 9, ED_OPT_AR_GEMV = 11,
       ED_OPT_AR_GEMV_WGS = 12 };
int ed_model_set_option(ed_model* m, int option, int value);

/* Measurement hooks (no counterpart in the reference).  With a probe interval n > 0, every n-th
 * sampling step (and every ed_forward) runs eagerly with a HIP-event pair around each kernel
 * launch on the caller's stream; ed_probe_read synchronises the stream, accumulates the event
 * times into stats[ED_K_COUNT] and clears the pending events. */
int ed_probe_config(ed_model* m, int every_n_steps);
int ed_probe_read(ed_model* m, ed_probe_stat* stats, void* stream);

/* Kernel-level test/benchmark hook (no counterpart in the reference): runs ONE implicit-GEMM problem
 *   out[b*L + l][n] = bias[n] (+ resid[b*L + l][n]) + sum_{tap,k} A[b*a_seq_pitch + a_row0 + l + (tap - ntaps/2)*dilation][k] * W[tap][n][k]
 * on caller-owned device buffers (A, W bf16 when a_bf16 != 0, else fp32; rows of A outside [0, L) of a
 * sequence must be readable and zero when ntaps > 1).  kernel: 0 = the library's own choice, 1 = generic
 * tile kernel, 2 = tuned bf16 kernel (fails with ED_EINVAL if the shape is outside its contract); 3 and above
 * select diagnostic variants of the tuned kernels (tools/gemm_bench.py lists them). */
int ed_debug_gemm(const void* A_dev, int64_t a_seq_pitch, int64_t a_row0, const void* W_dev, const float* bias_dev,
                  void* out_dev, const float* resid_dev, int B, int L, int N, int K, int ntaps, int dilation,
                  int a_bf16, int out_bf16, int kernel, void* stream);

/* Allocation-free variant of ed_debug_gemm(kernel = 100 + ksplit / 200 + ksplit): the small-tile kernel (tile = 128 or
 * 64) with its K loop cut `ksplit` ways across workgroups, then the fixed-order reduction with bias (+ residual).
 * It has NO counterpart in the reference; it exists for callers that cannot allocate - code under hipGraph capture,
 * tight serving loops, harnesses that forbid hipMalloc: where ed_debug_gemm keeps the partial sums in a buffer it
 * allocates, this entry point keeps them in a static device array of ED_NOALLOC_SCRATCH_BYTES (each GPU the library
 * runs on holds its own copy until the process ends) and calls no hipMalloc / hipFree or other allocating routine.
 * bf16 operands only (A, W bf16; out fp32 or bf16); same argument meaning and the same results, bit for bit, as
 * ed_debug_gemm.  ED_EINVAL when ksplit * B * L * N * 4 exceeds ED_NOALLOC_SCRATCH_BYTES, or ksplit is outside
 * [2, 16].  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 * serialise them.  The device must be current (hipSetDevice) as for ed_debug_gemm. */
#define ED_NOALLOC_SCRATCH_BYTES (16u << 20)
int ed_debug_gemm_splitk_noalloc(const void* A_dev, int64_t a_seq_pitch, int64_t a_row0, const void* W_dev,
                                 const float* bias_dev, void* out_dev, const float* resid_dev, int B, int L, int N, int K,
                                 int ntaps, int dilation, int out_bf16, int tile, int ksplit, void* stream);

/* Diagnostic hook for ed_debug_gemm(kernel = 9, or 500 + bits: the fused-LayerNorm form): a device buffer of n int64
 * that receives EIGHT values per 256-row tile of the tuned kernels: {prologue, K loop, epilogue} shader cycles and the
 * 100-MHz ticks they span (the windowed convolution stamps its K loop only: {0, loop cycles, 0, loop ticks}), then
 * the absolute 100-MHz times of workgroup entry, K-loop start, K-loop end (fused convolution: statistics published)
 * and exit with the stores drained.  NULL detaches the buffer. */
int ed_debug_set_stamp_buffer(int64_t* stamps_dev, int64_t n);

/* Kernel-level test/benchmark hook (no counterpart in the reference): out = act(LayerNorm(in)) row-wise,
 * in/out fp32 or bf16 [rows][C], act 0 none / 1 exact-erf GELU / 2 ReLU. */
int ed_debug_ln_act(const void* in_dev, int in_bf16, void* out_dev, int out_bf16, const float* gamma_dev,
                    const float* beta_dev, int rows, int C, int act, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVODIFF_HIP_H */
