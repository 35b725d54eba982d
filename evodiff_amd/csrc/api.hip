// C ABI of libevodiff_hip.so (declared in include/evodiff_hip.h): model handle, weight packing,
// workspace, per-step launch sequence, hipGraph capture/replay and the HIP-event probe.
#include "../../include/evodiff_hip.h"
#include "kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

using namespace ed;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess) return fail(ED_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                      __FILE__, __LINE__);                                          \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  hipError_t ensure(size_t n) {
    if (n <= bytes) return hipSuccess;
    release();
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <typename T> T* as() const { return (T*)p; }
};

struct LayerW {
  int dilation;
  const float *ln1_g, *ln1_b, *b1, *ln2_g, *ln2_b, *bc, *ln3_g, *ln3_b, *b2;  // fp32 vectors
  const float *w1_f, *wc_f, *w2_f;                                            // fp32 matrices
  const bf16_t *w1_h, *wc_h, *w2_h;                                           // bf16 matrices
};

// Dependency-cone pruning (OA-DM only): a step reads the network output at ONE position pos, so a layer only has to
// produce the positions its successors read.  Walking back from the output, a layer with dilation d whose output
// is needed on {p : p = pos mod s_out} reads its input on {p : p = pos mod gcd(s_out, d)}; such position sets are
// stored compactly (L / s rows per sequence, row m = position (pos mod s) + m*s).
struct LayerPlan {
  int s_in = 1, s_out = 1;  // 1 = every position
  long cp_row = 0;          // first row of this layer's compact convolution input inside Cprune (s_in > 1)
};

struct GraphKey {
  int mode, B, L, prec, n, variant;
  bool operator<(const GraphKey& o) const {
    return std::tie(mode, B, L, prec, n, variant) < std::tie(o.mode, o.B, o.L, o.prec, o.n, o.variant);
  }
};

struct PendingEvent {
  hipEvent_t a, b;
  int kind;
  double flops, bytes;
};

}  // namespace

struct ed_model {
  ed_config cfg;
  int device = 0;
  bool committed = false;
  int64_t n_params = 0;
  std::unordered_map<std::string, std::vector<float>> staged;

  DevBuf w32, w16;  // packed weight images
  DevBuf emb_table; // timesteps == 0: the n_tokens possible embedded rows, fp32 [n_tokens][d_model]
  const float *emb = nullptr, *up_w = nullptr, *up_b = nullptr, *pe = nullptr;
  const float *last_g = nullptr, *last_b = nullptr, *dec_w = nullptr, *dec_b = nullptr;
  std::vector<LayerW> layers;

  // workspace
  int ws_B = 0, ws_L = 0, ws_prec = -1;
  DevBuf X, X2, Abuf, Ybuf, Cpad, Cprune, state, tok, order, uni, qbuf, tvec;
  // Incremental leading layers (OA-DM, ED_OPT_INCREMENTAL): between two steps ONE token per sequence changes, so
  // layer k only changes within r_k = r_{k-1} + 2 d_k positions of it.  The first n_inc layers keep their outputs
  // (Xc[k] = input of layer k, Xc[0] = embedding) and convolution inputs (Cc[k]) from the previous step and
  // recompute the window of 2 r_k + 1 rows per sequence only.
  static constexpr int MAX_INC = 4;
  bool incremental = true;
  int n_inc = 0;
  int inc_r[MAX_INC + 1] = {0, 0, 0, 0, 0};  // inc_r[k + 1] = r_k; inc_r[0] = 0 (the changed token itself)
  DevBuf Xc[MAX_INC + 1], Cc[MAX_INC];
  DevBuf A1c;  // LayerNorm 1 of the first NON-incremental layer, kept across steps: only the window its input changed in is redone
  // Autoregressive sampling of a causal model (ed_ar_sample): the convolution inputs of EVERY layer are kept (past rows
  // never change), a step computes one new row per sequence and layer.  Car[k]: [B][conv_pitch][d_hidden].
  std::vector<DevBuf> Car;
  int ar_B = 0, ar_L = 0, ar_prec = -1;
  DevBuf ar_keep, ar_ndraw;   // ed_ar_sample_scaffold: given positions (L bytes), classes per step (n_steps ints)
  int ar_gemv = 8;            // ED_OPT_AR_GEMV: autoregressive batches up to this many sequences take the weight-streaming
                              // kernels (ar_gemv.hip); 0 = never, at most AR_GEMV_MAX_B.  Measured per step, 640M: 1.19 / 1.45 /
                              // 1.71 ms at B = 1 / 4 / 8 against 1.99 / 2.01 / 2.04 on the tile kernels
  int n_cus = 0;              // compute units of the model's device
  int ar_gemv_wgs = 0;        // ED_OPT_AR_GEMV_WGS: workgroups a weight-streaming launch uses at most (0 = one per CU)
  bool ar_forced = false;     // the current autoregressive call uses them
  // LayerNorm fused into the tuned kernels' epilogues (ED_OPT_FUSE_LN, ln_epilogue.hpp): exchange buffer for the per-row
  // partial statistics, arrival counters (one array per GEMM of the block: a counter must always see the same number of
  // column tiles), and a host-visible error word the kernels raise when a bounded wait runs out.
  bool fuse_ln = true;
  DevBuf Abuf1, ln_stats, ln_counters;
  long ln_counter_pitch = 0;
  unsigned* ln_err_host = nullptr;  // ED_AERR_WORDS host-visible words the kernels raise (kernels.hpp), checked by the
  unsigned* ln_err_dev = nullptr;   //   next call on the model: no per-call validation copies or synchronisations
  DevBuf partials;             // split-K workspace of the small-problem path (run_gemm)
  int split_target = 1024;     // ED_OPT_SPLIT_TARGET: workgroups a small GEMM is cut up to (0 = never split)
  int tuned_split_min_rows = 0;  // ED_OPT_TUNED_SPLIT_MIN_ROWS: problems of at least this many rows that are too small for the
                                 //   tuned kernel as it is run it with a cut K loop (0 = never)
  int tile64_max = 320;        // ED_OPT_TILE64_MAX: bf16 problems with at most this many 128 x 128 tiles use 64 x 64 tiles
  int partials_mb = 32;        // ED_OPT_PARTIALS_MB: size of that workspace (a cut needs ksplit x rows x N x 4 bytes)
  int split_min_iters = 4;     // ED_OPT_SPLIT_MIN_ITERS: K-steps (of 64 channels) a workgroup keeps at least
  bool fuse_reduce_ln = true;  // ED_OPT_FUSE_REDUCE_LN: the LayerNorm behind a split-K GEMM adds up the partial sums itself
  bool prune = true;           // ED_OPT_PRUNE
  int min_tuned_tiles = 130;   // ED_OPT_MIN_TUNED_TILES (tiles of the height the tuned kernel would use)
  std::vector<LayerPlan> plan; // per layer, for the current (B, L)
  int plan_s_final = 1;        // stride of the position set the last block produces
  long conv_pitch = 0, conv_row0 = 0;  // layout of Cpad: [B][conv_pitch][d_hidden], position 0 at row conv_row0
  hipStream_t cap_stream = nullptr;
  std::map<GraphKey, hipGraphExec_t> graphs;

  // probe
  int probe_every = 0;
  bool probing_now = false;
  std::vector<PendingEvent> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
  hipStream_t probe_stream = nullptr;

  ~ed_model() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
    for (auto& e : event_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto& e : pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    if (ln_err_host) (void)hipHostFree(ln_err_host);
  }
  void drop_graphs() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
    graphs.clear();
  }
};

namespace {

std::vector<int> dilations(const ed_config& c) {
  int period = 1;
  while ((1 << period) <= c.r) ++period;  // int(log2(r)) + 1   (model.py:108)
  std::vector<int> d(c.n_layers);
  for (int n = 0; n < c.n_layers; ++n) d[n] = 1 << (n % period);
  return d;
}

struct TensorSpec {
  std::string key;
  int64_t numel;
};

std::vector<TensorSpec> expected_tensors(const ed_config& c) {
  const int64_t D = c.d_model, H = c.d_hidden, E = c.d_embed, V = c.n_tokens, KS = c.kernel_size;
  std::vector<TensorSpec> t;
  t.push_back({"embedder.embedder.weight", V * E});
  t.push_back({"embedder.up_embedder.conv.weight", D * E});
  t.push_back({"embedder.up_embedder.conv.bias", D});
  for (int i = 0; i < c.n_layers; ++i) {
    const std::string p = "embedder.layers." + std::to_string(i) + ".";
    t.push_back({p + "conv.weight", H * H * KS});
    t.push_back({p + "conv.bias", H});
    t.push_back({p + "sequence1.0.weight", D});
    t.push_back({p + "sequence1.0.bias", D});
    t.push_back({p + "sequence1.2.conv.weight", H * D});
    t.push_back({p + "sequence1.2.conv.bias", H});
    t.push_back({p + "sequence1.3.weight", H});
    t.push_back({p + "sequence1.3.bias", H});
    t.push_back({p + "sequence2.0.weight", H});
    t.push_back({p + "sequence2.0.bias", H});
    t.push_back({p + "sequence2.2.conv.weight", D * H});
    t.push_back({p + "sequence2.2.conv.bias", D});
  }
  t.push_back({"last_norm.weight", D});
  t.push_back({"last_norm.bias", D});
  t.push_back({"decoder.conv.weight", V * D});
  t.push_back({"decoder.conv.bias", V});
  return t;
}

inline int64_t align64(int64_t n) { return (n + 63) & ~(int64_t)63; }  // element offsets: keep 256-B alignment

// ------------------------------------------------------------------------------------------------
// probe helpers
// ------------------------------------------------------------------------------------------------
struct Probe {
  ed_model* m;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  Probe(ed_model* m_, hipStream_t s_) : m(m_), s(s_), on(m_->probing_now) {}
  hipError_t begin() {
    if (!on) return hipSuccess;
    if (!m->event_pool.empty()) {
      a = m->event_pool.back().first;
      b = m->event_pool.back().second;
      m->event_pool.pop_back();
    } else {
      hipError_t e = hipEventCreate(&a);
      if (e != hipSuccess) return e;
      e = hipEventCreate(&b);
      if (e != hipSuccess) return e;
    }
    return hipEventRecord(a, s);
  }
  hipError_t end(int kind, double flops, double bytes) {
    if (!on) return hipSuccess;
    hipError_t e = hipEventRecord(b, s);
    m->pending.push_back({a, b, kind, flops, bytes});
    return e;
  }
};

#define LAUNCH_TRY(expr)                                                                                 \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail(ED_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                                      __FILE__, __LINE__);                                               \
  } while (0)

// The tuned kernels work on 256-row (128-row for mid-size problems) x 256/320-column tiles, one per CU at a time: a
// problem with few such tiles leaves most of the chip idle, and the small-tile generic kernel is the faster one.
bool use_tuned_gemm(const ed_model* m, const GemmProblem& p) {
  if (!p.a_bf16 || !gemm_bf16_supported(p)) return false;
  const int rows = gemm_bf16_tile_rows(p);
  const long tiles = (long)p.B * ((p.L + rows - 1) / rows) * (p.N % 320 == 0 ? p.N / 320 : p.N / 256);
  return tiles >= m->min_tuned_tiles;
}

// `deferred`: the caller's next launch is the LayerNorm of this result.  If the K loop gets cut across workgroups the
// reduction launch is left out and *deferred receives the problem with its partial sums (ksplit > 1) for
// run_ln_after, which then adds them up inside the LayerNorm kernel.
// Mid-size problems (ED_OPT_TUNED_SPLIT_MIN_ROWS <= rows, fewer tuned tiles than ED_OPT_MIN_TUNED_TILES): how many ways
// the tuned kernel's K loop is cut so that about one workgroup per CU runs; 1 = not this path.
int tuned_split(const ed_model* m, const GemmProblem& p) {
  if (m->tuned_split_min_rows <= 0 || !p.a_bf16 || !gemm_bf16_supported(p) || !m->partials.p) return 1;
  const long rows = (long)p.B * p.L;
  if (rows < m->tuned_split_min_rows || gemm_bf16_tile_rows(p) != 128) return 1;
  const long tiles = (long)p.B * ((p.L + 127) / 128) * (p.N % 320 == 0 ? p.N / 320 : p.N / 256);
  if (tiles >= m->min_tuned_tiles) return 1;
  const long steps = (long)p.ntaps * (p.K / 64);
  long ks = std::min<long>({(256 + tiles - 1) / tiles, steps / 4, 16L});
  while (ks > 1 && (size_t)ks * rows * p.N * 4 > m->partials.bytes) --ks;
  return (int)std::max<long>(ks, 1);
}

int run_gemm(ed_model* m, const GemmProblem& p, int kind, hipStream_t s, GemmProblem* deferred = nullptr) {
  if (deferred) deferred->ksplit = 1;
  Probe pr(m, s);
  LAUNCH_TRY(pr.begin());
  const int tuned_ks = tuned_split(m, p);
  if (use_tuned_gemm(m, p)) {
    LAUNCH_TRY(launch_gemm_bf16(p, s));
    if (kind == ED_K_CONV && !conv_window_supported(p)) kind = ED_K_CONV_OTHER;
  } else if (tuned_ks > 1) {
    // mid-size problems: too few tiles for the tuned kernel as it is, too many rows for the small-tile kernel to be
    // efficient - the tuned kernel (128-row tiles) with its K loop cut across workgroups
    GemmProblem q = p;
    q.ksplit = tuned_ks;
    q.partial = m->partials.as<float>();
    LAUNCH_TRY(launch_gemm_bf16(q, s));
    if (deferred && m->fuse_reduce_ln && reduce_ln_supported(q)) *deferred = q;
    else LAUNCH_TRY(launch_reduce_partials(q, s));
    if (kind == ED_K_CONV) kind = ED_K_CONV_OTHER;
  } else {
    // Small problems in the fast path: a few workgroups streaming cold weights are latency-bound, so the K loop
    // is cut across workgroups (deterministic: partial sums to a workspace, added up in a fixed order).
    GemmProblem q = p;
    // problems with few 128 x 128 tiles take 64 x 64 tiles: four times the workgroups before any K loop is cut, so
    // fewer (or no) partial sums to write and add up
    long tiles = (long)p.B * ((p.L + 127) / 128) * ((p.N + 127) / 128);
    if (p.a_bf16 && tiles <= m->tile64_max) {
      q.tile = 64;
      tiles = (long)p.B * ((p.L + 63) / 64) * ((p.N + 63) / 64);
    }
    if (p.a_bf16 && m->split_target > 0 && m->partials.p) {
      const int iters = p.ntaps * ((p.K + 63) / 64);
      // few tiles: cutting the K loops beyond ~4 workgroups per tile costs more in partial sums than it saves (measured)
      const long target = std::min<long>(m->split_target, std::max<long>(512, 4 * tiles));
      long ks = std::min<long>({target / tiles, (long)iters / m->split_min_iters, 16L});
      while (ks > 1 && (size_t)ks * p.B * p.L * p.N * 4 > m->partials.bytes) --ks;
      if (ks > 1) {
        q.ksplit = (int)ks;
        q.partial = m->partials.as<float>();
      }
    }
    LAUNCH_TRY(launch_gemm_generic(q, s));
    if (q.ksplit > 1) {
      if (deferred && m->fuse_reduce_ln && reduce_ln_supported(q)) *deferred = q;
      else LAUNCH_TRY(launch_reduce_partials(q, s));
    }
    if (kind == ED_K_CONV) kind = ED_K_CONV_OTHER;
  }
  double taps = 0;
  for (int t = 0; t < p.ntaps; ++t) {
    const long off = std::labs((long)(t - p.center()) * p.dilation);
    taps += off < p.L ? (double)(p.L - off) : 0.0;
  }
  const double flops = 2.0 * p.B * taps * p.N * (double)p.K;
  const double esz = p.a_bf16 ? 2.0 : 4.0;
  const double rows = (double)p.B * p.L;
  const double bytes = rows * p.K * esz + (double)p.ntaps * p.N * p.K * esz +
                       rows * p.N * (p.out_bf16 ? 2.0 : 4.0) + (p.resid ? rows * p.N * 4.0 : 0.0);
  LAUNCH_TRY(pr.end(kind, flops, bytes));
  return ED_OK;
}

// The LayerNorm + GELU behind GEMM `p` (kind 0 PFF1, 1 convolution, 2 PFF2) applied in p's own epilogue, when p runs on
// the tuned kernels and they can (ln_epilogue.hpp): fills p.ln and returns true; the caller then skips that LayerNorm.
// Flat GEMM row R goes to row R of `out`, or with seq_len > 0 to row (R / seq_len) * pitch + row0 + R % seq_len.
bool fuse_ln_into(ed_model* m, GemmProblem& p, int kind, const float* gamma, const float* beta, void* out, int seq_len,
                  long pitch, long row0) {
  if (!m->fuse_ln || m->cfg.activation != 1 || !p.a_bf16 || !m->ln_stats.p) return false;
  if (!use_tuned_gemm(m, p) || !gemm_bf16_ln_supported(p)) return false;
  const long row_tiles = (long)p.B * ((p.L + 127) / 128);
  if (row_tiles > m->ln_counter_pitch) return false;
  {  // the exchange slab the launch indexes: (row tile, column tile, tile row) at the tile height the launcher picks
    const long bm = gemm_bf16_tile_rows(p), tn = p.N % 320 == 0 ? p.N / 320 : p.N / 256;
    const size_t slots = (size_t)p.B * ((p.L + bm - 1) / bm) * bm * tn;
    if (slots * 8 > m->ln_stats.bytes) return false;
  }
  p.ln.gamma = gamma;
  p.ln.beta = beta;
  p.ln.out = (bf16_t*)out;
  p.ln.seq_len = seq_len;
  p.ln.out_pitch = pitch;
  p.ln.out_row0 = row0;
  p.ln.stats = m->ln_stats.p;
  p.ln.counters = m->ln_counters.as<unsigned>() + (long)kind * m->ln_counter_pitch;
  p.ln.error = m->ln_err_dev + ED_AERR_LN_WAIT;
  return true;
}

// Errors the kernels could only report asynchronously: raised in host-visible words by an EARLIER launch (out-of-range
// tokens / timesteps / positions / targets, which the kernels clamp, or a bounded inter-workgroup wait that ran out).
int check_async_errors(ed_model* m) {
  if (!m->ln_err_host) return ED_OK;
  volatile unsigned* w = m->ln_err_host;
  static const char* const what[] = {
      "a fused-LayerNorm kernel gave up waiting for the other workgroups of its row tile (ED_OPT_FUSE_LN = 0 avoids the exchange)",
      "token ids outside [0, n_tokens) (D3PM: [0, K)) were clamped", "timesteps outside [0, timesteps) were clamped",
      "order positions outside [0, L) were clamped", "targets outside [0, n_classes) were clamped"};
  for (int code = 0; code < 5; ++code)
    if (w[code]) {
      for (int k = 0; k < ED_AERR_WORDS; ++k) w[k] = 0;
      return fail(code == ED_AERR_LN_WAIT ? ED_EHIP : ED_EINVAL,
                  "an earlier call on this model ran with invalid data: %s; its results are not valid", what[code]);
    }
  return ED_OK;
}

int run_ln(ed_model* m, const void* in, bool in_bf16, void* out, bool out_bf16, const float* g, const float* b,
           int rows, int C, const LnRowMap& map, int act, hipStream_t s) {
  Probe pr(m, s);
  LAUNCH_TRY(pr.begin());
  LAUNCH_TRY(launch_ln_act(in, in_bf16, out, out_bf16, g, b, rows, C, map, act, s));
  LAUNCH_TRY(pr.end(ED_K_LNACT, 0.0, (double)rows * C * ((in_bf16 ? 2.0 : 4.0) + (out_bf16 ? 2.0 : 4.0))));
  return ED_OK;
}

// The LayerNorm behind a GEMM launched with `deferred` (see run_gemm): in / in_bf16 describe the GEMM's stored result,
// which exists only if its reduction ran.
int run_ln_after(ed_model* m, const GemmProblem& deferred, const void* in, bool in_bf16, void* out, bool out_bf16,
                 const float* g, const float* b, int rows, int C, const LnRowMap& map, int act, hipStream_t s) {
  if (deferred.ksplit <= 1) return run_ln(m, in, in_bf16, out, out_bf16, g, b, rows, C, map, act, s);
  if (!out_bf16 || C != deferred.N) return fail(ED_ESTATE, "deferred split-K reduction needs a bf16 LayerNorm of the GEMM's width");
  Probe pr(m, s);
  LAUNCH_TRY(pr.begin());
  LAUNCH_TRY(launch_reduce_ln(deferred, (bf16_t*)out, g, b, rows, map, act, s));
  LAUNCH_TRY(pr.end(ED_K_LNACT, 0.0, (double)rows * C * (4.0 * deferred.ksplit + 2.0)));
  return ED_OK;
}

// Largest window problem (in 256-row tiles) an incremental leading layer may have: beyond it recomputing the
// window on the small-tile kernel is no cheaper than the whole layer on the tuned one (measured at B = 64).
constexpr long INC_MAX_TILES = 70;

int ensure_workspace(ed_model* m, int B, int L, int prec, hipStream_t s) {
  const ed_config& c = m->cfg;
  const size_t rows = (size_t)B * L + 256;  // slack rows: tile loads may run past the last sequence
  const size_t Cmax = (size_t)std::max(c.d_model, c.d_hidden);
  bool grew = false;
  auto need = [&](DevBuf& b, size_t n) -> hipError_t {
    if (n > b.bytes) grew = true;
    return b.ensure(n);
  };
  HIP_TRY(need(m->X, rows * c.d_model * 4));
  HIP_TRY(need(m->Abuf, rows * Cmax * 4));
  HIP_TRY(need(m->Ybuf, rows * (size_t)c.d_hidden * 4));
  HIP_TRY(need(m->tok, (size_t)B * L * 8));
  HIP_TRY(need(m->tvec, (size_t)B * 8));
  HIP_TRY(need(m->state, 256));
  // Convolution input: per sequence [halo zeros | L rows (+ round-up to the 256-row tile) | halo zeros]; the
  // LayerNorm that produces it only ever writes the L valid rows, so the zeros survive across steps.
  int max_dil = 1;
  for (const auto& lw : m->layers) max_dil = std::max(max_dil, lw.dilation);
  const long halo = (long)(c.causal ? c.kernel_size - 1 : c.kernel_size / 2) * max_dil;  // causal: all taps look back
  const long pitch = halo + ((L + 255) / 256) * 256L + halo;
  const size_t cbytes = (size_t)B * pitch * c.d_hidden * 4;
  // the element size (and with it every row's byte position) differs between the two precisions
  const bool relayout = B != m->ws_B || L != m->ws_L || prec != m->ws_prec || pitch != m->conv_pitch ||
                        cbytes > m->Cpad.bytes;
  HIP_TRY(need(m->Cpad, cbytes));
  if (relayout) HIP_TRY(hipMemsetAsync(m->Cpad.p, 0, m->Cpad.bytes, s));
  // pruning plan for this L (see LayerPlan) and the compact convolution inputs it needs, one zero-initialised
  // region per pruned layer: [gap zeros | B x (L/s_in rows | gap zeros) | tile slack]
  m->plan.assign((size_t)c.n_layers, LayerPlan{});
  m->plan_s_final = 1;
  long prune_rows = 0;
  if (m->prune && !c.causal) {
    int sset = 1;
    while (sset < (1 << 20) && L % (2 * sset) == 0) sset *= 2;  // largest power of two dividing L
    m->plan_s_final = sset;
    for (int i = c.n_layers - 1; i >= 0 && sset > 1; --i) {
      const int d = m->layers[i].dilation;
      int s_in = std::gcd(sset, d);
      long gap = (long)(c.kernel_size / 2) * (d / s_in);
      if (gap > L / s_in) s_in = 1;  // more padding than data: read every position instead
      LayerPlan& pl = m->plan[i];
      pl.s_out = sset;
      pl.s_in = s_in;
      if (s_in > 1) {
        pl.cp_row = prune_rows;
        const long body = (long)B * (L / s_in + gap);
        prune_rows += gap + ((body + 255) / 256) * 256 + gap + 256;
      }
      sset = s_in;
    }
  }
  const size_t pbytes = (size_t)prune_rows * c.d_hidden * 4;
  const bool prelayout = relayout || pbytes > m->Cprune.bytes;
  HIP_TRY(need(m->Cprune, pbytes));
  if (prelayout && m->Cprune.p) HIP_TRY(hipMemsetAsync(m->Cprune.p, 0, m->Cprune.bytes, s));
  if (m->plan_s_final > 1) HIP_TRY(need(m->X2, ((size_t)B * L / 2 + 256) * c.d_model * 4));
  HIP_TRY(need(m->partials, (size_t)m->partials_mb << 20));
  {
    HIP_TRY(need(m->Abuf1, rows * Cmax * 4));
    // fused LayerNorm exchange: a row tile is 128 or 256 rows of ONE sequence (every row of a ragged last tile has a
    // slot, so a sequence takes ceil(L / 256) * 256 slots per column tile at the larger tile height), a column tile at
    // least 256 channels; fuse_ln_into checks every problem against this size
    const size_t tile_rows = (size_t)B * ((L + 255) / 256) * 256 + 256;
    const size_t max_tn = std::max<size_t>(1, Cmax / 256);
    HIP_TRY(need(m->ln_stats, tile_rows * max_tn * 8));
    const size_t n_ctr = (((size_t)B * ((L + 127) / 128) * 128 + 256) / 128 + 3) & ~(size_t)3;  // counters per GEMM kind (16-byte multiple)
    if (3 * n_ctr * 4 > m->ln_counters.bytes) {
      HIP_TRY(need(m->ln_counters, 3 * n_ctr * 4));
      HIP_TRY(hipMemsetAsync(m->ln_counters.p, 0, m->ln_counters.bytes, s));  // zero once; launches keep them consistent
    }
    m->ln_counter_pitch = (long)(m->ln_counters.bytes / 12);
  }
  // incremental leading layers: as many as keep their window problems on the small-tile kernel, none of them pruned
  m->n_inc = 0;
  if (m->incremental && c.timesteps == 0 && !c.causal) {
    int r = 0;
    for (int k = 0; k < ed_model::MAX_INC && k < c.n_layers - 1; ++k) {
      r += (c.kernel_size / 2) * m->layers[k].dilation;
      const long M = 2L * r + 1;
      const long tiles = ((long)B * M + 255) / 256 * ((c.d_hidden + 319) / 320);
      if (M * 2 > L || tiles >= INC_MAX_TILES || m->plan[k].s_out > 1 || m->plan[k + 1].s_out > 1) break;
      m->inc_r[k + 1] = r;
      m->n_inc = k + 1;
    }
  }
  for (int k = 0; k <= m->n_inc && m->n_inc > 0; ++k) HIP_TRY(need(m->Xc[k], rows * c.d_model * 4));
  if (m->n_inc > 0) HIP_TRY(need(m->A1c, rows * Cmax * 4));
  for (int k = 0; k < m->n_inc; ++k) {
    const bool fresh = relayout || cbytes > m->Cc[k].bytes;
    HIP_TRY(need(m->Cc[k], cbytes));
    if (fresh) HIP_TRY(hipMemsetAsync(m->Cc[k].p, 0, m->Cc[k].bytes, s));
  }
  if (pitch != m->conv_pitch) grew = true;  // the graphs bake the pitch in
  m->ws_prec = prec;
  m->conv_pitch = pitch;
  m->conv_row0 = halo;
  if (grew) m->drop_graphs();  // captured graphs hold the old pointers
  m->ws_B = B;
  m->ws_L = L;
  return ED_OK;
}

// Variants of a step: every position with no state kept (forward, D3PM), every position while filling the caches of
// the incremental layers (first OA-DM step of a call), and the incremental step proper.
enum { STEP_PLAIN = 0, STEP_FILL = 1, STEP_INCR = 2, STEP_AR = 3, STEP_AR_SCAFFOLD = 4 };

// One incremental leading layer k (STEP_INCR): the rows within r_{k-1} of the changed position go through
// LN1 / PFF1 / LN2 into the cached convolution input, the rows within r_k through conv / LN3 / PFF2 into the cached
// layer output.  Windows have a fixed size and are clamped into [0, L), so the launch shapes do not depend on the
// position (hipGraph) - rows they cover beyond the changed ones are recomputed to the same values.
int run_incremental_layer(ed_model* m, int k, int B, int L, int prec, hipStream_t s) {
  const ed_config& c = m->cfg;
  const int D = c.d_model, H = c.d_hidden;
  const bool h = prec == ED_PREC_BF16;
  const int act = c.activation == 1 ? 1 : 2;
  const LayerW& w = m->layers[k];
  const int M_in = 2 * m->inc_r[k] + 1, M_out = 2 * m->inc_r[k + 1] + 1;
  const int rows_in = B * M_in, rows_out = B * M_out;
  const int* order = m->order.as<int>();
  const StepState* st = m->state.as<StepState>();
  const PosBit win_in = PosBit::window(order, st, m->inc_r[k], L - M_in);
  const PosBit win_out = PosBit::window(order, st, m->inc_r[k + 1], L - M_out);
  const float* x_in = m->Xc[k].as<float>();
  float* x_out = m->Xc[k + 1].as<float>();
  void* A = m->Abuf.p;
  void* Y = m->Ybuf.p;
  const bool y_bf16 = h;
  int rc;
  const LnRowMap gather_in{M_in, L, 0, 1, win_in, M_in, 0, PosBit{}};
  if ((rc = run_ln(m, x_in, false, A, h, w.ln1_g, w.ln1_b, rows_in, D, gather_in, act, s))) return rc;
  GemmProblem g1{A, rows_in, 0, h ? (const void*)w.w1_h : (const void*)w.w1_f, w.b1, Y, nullptr, 1, rows_in, H, D, 1, 1, h, y_bf16};
  GemmProblem dq{};
  if ((rc = run_gemm(m, g1, ED_K_GEMM1X1, s, &dq))) return rc;
  const LnRowMap scatter_pad{M_in, M_in, 0, 1, PosBit{}, m->conv_pitch, m->conv_row0, win_in};
  if ((rc = run_ln_after(m, dq, Y, y_bf16, m->Cc[k].p, h, w.ln2_g, w.ln2_b, rows_in, H, scatter_pad, act, s))) return rc;
  GemmProblem gc{m->Cc[k].p, m->conv_pitch, m->conv_row0, h ? (const void*)w.wc_h : (const void*)w.wc_f, w.bc, Y, nullptr,
                 1, rows_out, H, H, c.kernel_size, w.dilation, h, y_bf16};
  gc.tap_center = c.causal ? c.kernel_size - 1 : -1;
  gc.seq_rows = M_out;
  gc.a_unbounded = true;
  gc.a_sel = win_out;
  gc.out_pitch = M_out;
  if ((rc = run_gemm(m, gc, ED_K_CONV, s, &dq))) return rc;
  if ((rc = run_ln_after(m, dq, Y, y_bf16, A, h, w.ln3_g, w.ln3_b, rows_out, H, LnRowMap::plain(rows_out), act, s))) return rc;
  GemmProblem g2{A, 0, 0, h ? (const void*)w.w2_h : (const void*)w.w2_f, w.b2, x_out, x_in, 1, rows_out, D, H, 1, 1, h, false};
  g2.seq_rows = M_out;
  g2.a_seq_pitch = M_out;
  g2.out_pitch = L;
  g2.out_sel = win_out;
  g2.resid_pitch = L;
  g2.resid_sel = win_out;
  return run_gemm(m, g2, ED_K_GEMM1X1, s);
}

// The block stack (model.py:142-147 + ByteNetBlock).  The fp32 residual stream starts in X (STEP_PLAIN: layers update
// it in place) or in Xc[0] (STEP_FILL / STEP_INCR: layer k < n_inc reads Xc[k] and writes Xc[k+1], layer n_inc reads
// Xc[n_inc] and writes X, the rest update X in place).  With `pruned` (OA-DM steps only) the trailing layers follow
// m->plan: they work on compact position sets and a layer that narrows the set writes its output to the other of the
// two residual buffers.  *x_out receives the buffer that holds the last block's output, L / m->plan_s_final rows per
// sequence when pruned.
int run_blocks(ed_model* m, int B, int L, int prec, float* acts, bool pruned, int variant, const float** x_out,
               hipStream_t s) {
  const ed_config& c = m->cfg;
  const int D = c.d_model, H = c.d_hidden, half = c.kernel_size / 2;
  const bool h = prec == ED_PREC_BF16;
  const size_t esz = h ? 2 : 4;
  const int act = c.activation == 1 ? 1 : 2;
  const int n_inc = variant == STEP_PLAIN ? 0 : m->n_inc;
  float* Xc = m->X.as<float>();
  float* Xo = m->X2.as<float>();
  void* A = m->Abuf.p;    // LayerNorm 3 -> PFF2
  void* A1 = m->Abuf1.p;  // LayerNorm 1 -> PFF1 (its own buffer: a PFF2 that applies the next block's LayerNorm 1 in its
                          // epilogue writes it while other workgroups of the launch still read A)
  void* Y = m->Ybuf.p;
  // GEMM1 / conv results feed a LayerNorm straight away; in the fast path they are stored as bf16 (the
  // LayerNorm output is rounded to bf16 right after anyway), which cuts a third of the LayerNorm traffic
  const bool y_bf16 = h;
  bool ln1_done = false;  // the previous block's PFF2 reduction already ran this block's first LayerNorm
  auto pruned_here = [](const LayerPlan& pl) { return pl.s_in != 1 || pl.s_out != 1; };
  for (int i = 0; i < c.n_layers; ++i) {
    int rc;
    if (variant == STEP_INCR && i < n_inc) {
      if ((rc = run_incremental_layer(m, i, B, L, prec, s))) return rc;
      continue;
    }
    const LayerW& w = m->layers[i];
    const LayerPlan pl = pruned ? m->plan[i] : LayerPlan{};
    const int M_in = L / pl.s_in, M_out = L / pl.s_out, in_mul = pl.s_out / pl.s_in;
    const int rows_in = B * M_in, rows_out = B * M_out;
    // residual stream of this layer: where it is read and where the layer's output goes
    const float* x_src = Xc;
    float* x_dst = Xc;
    void* cpad = m->Cpad.p;
    if (i < n_inc) {
      x_src = m->Xc[i].as<float>();
      x_dst = m->Xc[i + 1].as<float>();
      cpad = m->Cc[i].p;
    } else if (i == n_inc && n_inc > 0) {
      x_src = m->Xc[i].as<float>();
    }
    // The first layer behind the incremental ones: its input only changed inside the last incremental layer's output
    // window, so its LayerNorm 1 is kept across steps (A1c) and only that window is redone
    const bool cached_ln1 = n_inc > 0 && i == n_inc && !pruned_here(pl);        // this block reads A1c
    const bool next_cached = n_inc > 0 && i + 1 == n_inc;                         // the next block does: its LN1 goes there
    void* const A1_in = cached_ln1 ? m->A1c.p : A1;     // this block's LayerNorm 1 result, the input of its PFF1
    void* const A1_next = next_cached ? m->A1c.p : A1;  // where a PFF2 that applies the next block's LayerNorm 1 puts it
    if (cached_ln1 && !ln1_done) {
      if (variant == STEP_INCR) {
        const int M_w = 2 * m->inc_r[n_inc] + 1;
        const PosBit win = PosBit::window(m->order.as<int>(), m->state.as<StepState>(), m->inc_r[n_inc], L - M_w);
        const LnRowMap window{M_w, L, 0, 1, win, L, 0, win};
        if ((rc = run_ln(m, x_src, false, A1_in, h, w.ln1_g, w.ln1_b, B * M_w, D, window, act, s))) return rc;
      } else {
        if ((rc = run_ln(m, x_src, false, A1_in, h, w.ln1_g, w.ln1_b, rows_in, D, LnRowMap::plain(rows_in), act, s))) return rc;
      }
      ln1_done = true;
    }
    // the 1x1 projections are row-independent: tile them over all rows as one long sequence
    if (!ln1_done && (rc = run_ln(m, x_src, false, A1_in, h, w.ln1_g, w.ln1_b, rows_in, D, LnRowMap::plain(rows_in), act, s))) return rc;
    ln1_done = false;
    GemmProblem g1{A1_in, rows_in, 0, h ? (const void*)w.w1_h : (const void*)w.w1_f, w.b1, Y, nullptr, 1, rows_in, H, D, 1, 1, h, y_bf16};
    GemmProblem dq{};
    long y_pitch;  // rows per sequence of the convolution result in Y
    bool ln3_fused = false;
    if (pl.s_in == 1) {
      // every position: per-sequence zero halo in Cpad.  Where the tuned kernels run, each GEMM applies the LayerNorm
      // that follows it in its own epilogue (ln_epilogue.hpp) and the stand-alone launch is skipped.
      const LnRowMap to_pad{L, L, 0, 1, PosBit{}, m->conv_pitch, m->conv_row0, PosBit{}};
      const bool ln2_fused = fuse_ln_into(m, g1, 0, w.ln2_g, w.ln2_b, cpad, L, m->conv_pitch, m->conv_row0);
      if ((rc = run_gemm(m, g1, ED_K_GEMM1X1, s, ln2_fused ? nullptr : &dq))) return rc;
      if (!ln2_fused && (rc = run_ln_after(m, dq, Y, y_bf16, cpad, h, w.ln2_g, w.ln2_b, rows_in, H, to_pad, act, s))) return rc;
      GemmProblem gc{cpad, m->conv_pitch, m->conv_row0, h ? (const void*)w.wc_h : (const void*)w.wc_f, w.bc, Y,
                     nullptr, B, L, H, H, c.kernel_size, w.dilation, h, y_bf16};
      gc.tap_center = c.causal ? c.kernel_size - 1 : -1;
      ln3_fused = in_mul == 1 && fuse_ln_into(m, gc, 1, w.ln3_g, w.ln3_b, A, 0, 0, 0);
      if ((rc = run_gemm(m, gc, ED_K_CONV, s, ln3_fused ? nullptr : &dq))) return rc;
      y_pitch = L;
    } else {
      // compact position set: neighbours at +-d are dc = d / s_in rows apart; the sequences sit back to back with
      // `gap` zero rows between them and are convolved as ONE long sequence (the rows in the gaps are never read)
      const int dc = w.dilation / pl.s_in;
      const long gap = (long)half * dc, pitch = M_in + gap;
      char* Cc = (char*)m->Cprune.p + (size_t)pl.cp_row * H * esz;
      const LnRowMap to_compact{M_in, M_in, 0, 1, PosBit{}, pitch, gap, PosBit{}};
      const bool ln2_fused = fuse_ln_into(m, g1, 0, w.ln2_g, w.ln2_b, Cc, M_in, pitch, gap);
      if ((rc = run_gemm(m, g1, ED_K_GEMM1X1, s, ln2_fused ? nullptr : &dq))) return rc;
      if (!ln2_fused && (rc = run_ln_after(m, dq, Y, y_bf16, Cc, h, w.ln2_g, w.ln2_b, rows_in, H, to_compact, act, s))) return rc;
      GemmProblem gc{Cc, 0, gap, h ? (const void*)w.wc_h : (const void*)w.wc_f, w.bc, Y, nullptr,
                     1, (int)(B * pitch - gap), H, H, c.kernel_size, dc, h, y_bf16};
      if ((rc = run_gemm(m, gc, ED_K_CONV, s, &dq))) return rc;
      y_pitch = pitch;
    }
    // positions kept for the next layer: every in_mul-th row, starting at the row the fill position selects
    PosBit sel;
    if (in_mul > 1) {
      sel.order = m->order.as<int>();
      sel.state = m->state.as<StepState>();
      sel.shift = __builtin_ctz(pl.s_in);
      sel.mask = in_mul - 1;
    }
    const LnRowMap keep{M_out, y_pitch, 0, in_mul, sel, M_out, 0, PosBit{}};
    if (!ln3_fused && (rc = run_ln_after(m, dq, Y, y_bf16, A, h, w.ln3_g, w.ln3_b, rows_out, H, keep, act, s))) return rc;
    GemmProblem g2{A, rows_out, 0, h ? (const void*)w.w2_h : (const void*)w.w2_f, w.b2, x_dst, x_src, 1, rows_out, D, H, 1, 1, h, false};
    if (in_mul > 1) {
      g2.out = Xo;
      g2.resid_mul = in_mul;
      g2.resid_sel = sel;
      std::swap(Xc, Xo);
    }
    // PFF2 is followed by the first LayerNorm of the next block, which reads exactly the rows this block produced:
    // fused into PFF2's epilogue on the tuned kernels; a split-K PFF2 leaves its reduction to that LayerNorm (which
    // also stores the new residual stream)
    GemmProblem dq2{};
    const bool has_next = i + 1 < c.n_layers;
    const bool ln1_fused = has_next && fuse_ln_into(m, g2, 2, m->layers[i + 1].ln1_g, m->layers[i + 1].ln1_b, A1_next, 0, 0, 0);
    if ((rc = run_gemm(m, g2, ED_K_GEMM1X1, s, h && has_next && !ln1_fused ? &dq2 : nullptr))) return rc;
    if (ln1_fused) ln1_done = true;
    if (dq2.ksplit > 1) {
      const LayerW& wn = m->layers[i + 1];
      if ((rc = run_ln_after(m, dq2, nullptr, false, A1_next, h, wn.ln1_g, wn.ln1_b, rows_out, D, LnRowMap::plain(rows_out), act, s))) return rc;
      ln1_done = true;
    }
    if (acts)
      HIP_TRY(hipMemcpyAsync(acts + (size_t)(i + 1) * rows_out * D, Xc, (size_t)rows_out * D * 4, hipMemcpyDeviceToDevice, s));
  }
  if (x_out) *x_out = Xc;
  return ED_OK;
}

int run_embed(ed_model* m, const int64_t* tokens, const int64_t* tvec, int B, int L, int variant, hipStream_t s) {
  const ed_config& c = m->cfg;
  Probe pr(m, s);
  LAUNCH_TRY(pr.begin());
  float* x0 = variant != STEP_PLAIN && m->n_inc > 0 ? m->Xc[0].as<float>() : m->X.as<float>();
  if (variant == STEP_INCR && m->n_inc > 0) {
    // only the row the previous step filled differs from the cached embedding
    LAUNCH_TRY(launch_embed_gather_rows(tokens, m->emb_table.as<float>(), x0, B, L, c.d_model, c.n_tokens,
                                        PosBit::window(m->order.as<int>(), m->state.as<StepState>(), 0, L - 1), s));
    LAUNCH_TRY(pr.end(ED_K_EMBED, 0.0, (double)B * (8 + c.d_model * 4.0)));
    return ED_OK;
  }
  if (c.timesteps == 0)
    LAUNCH_TRY(launch_embed_gather(tokens, m->emb_table.as<float>(), x0, B * L, c.d_model, c.n_tokens, s));
  else
    LAUNCH_TRY(launch_embed(tokens, tvec, m->state.as<StepState>(), m->emb, m->pe, c.timesteps, m->up_w, m->up_b,
                            m->X.as<float>(), B * L, B, c.d_model, c.d_embed, c.n_tokens, s));
  LAUNCH_TRY(pr.end(ED_K_EMBED, 2.0 * B * L * c.d_embed * c.d_model, (double)B * L * (8 + c.d_model * 4.0)));
  return ED_OK;
}

int check_shape(const ed_model* m, int B, int L, int prec) {
  if (!m) return fail(ED_EINVAL, "null model handle");
  if (!m->committed) return fail(ED_ESTATE, "ed_model_commit has not been called");
  if (B <= 0 || L <= 0) return fail(ED_EINVAL, "B and L must be positive (got B=%d L=%d)", B, L);
  if ((int64_t)B * L > (1 << 30)) return fail(ED_EINVAL, "B*L too large");
  if (prec != ED_PREC_FP32 && prec != ED_PREC_BF16) return fail(ED_EINVAL, "unknown precision %d", prec);
  return ED_OK;
}

enum { MODE_OADM = 0, MODE_D3PM = 1, MODE_AR = 2 };

// One autoregressive step of a causal model (generate.py:238-244 without re-running the prefix): the token at position
// p = order[step] is new; every layer computes row p only - its 1x1 projections on the B new rows, its convolution
// from the cached rows p - tau*d of its own input - and the draw from row p's logits becomes token p + 1.
int run_ar_step(ed_model* m, int B, int L, int n, int prec, float* probs, hipStream_t s) {
  const ed_config& c = m->cfg;
  const int D = c.d_model, H = c.d_hidden;
  const bool h = prec == ED_PREC_BF16;
  const int act = c.activation == 1 ? 1 : 2;
  const PosBit pos = PosBit::window(m->order.as<int>(), m->state.as<StepState>(), 0, L - 1, /*lag=*/0);
  float* x = m->X.as<float>();   // residual stream: one row per sequence
  void* A = m->Abuf.p;
  void* Y = m->Ybuf.p;
  int rc;
  bool ln1_done = false;
  // few sequences, bf16: three weight-streaming launches per block (ar_gemv.hip), each applying the LayerNorm in front of
  // its GEMM itself; the raw GEMM results stay fp32 between them
  const bool gemv = h && B <= m->ar_gemv && c.kernel_size == 5;
  auto run_gemv = [&](const ArGemv& g, int kind) -> int {
    Probe pr(m, s);
    LAUNCH_TRY(pr.begin());
    LAUNCH_TRY(launch_ar_gemv(g, m->ar_gemv_wgs > 0 ? m->ar_gemv_wgs : m->n_cus, s));
    LAUNCH_TRY(pr.end(kind, 2.0 * g.B * g.N * (double)g.C * g.ntaps, (double)g.ntaps * g.N * g.C * 2.0));
    return ED_OK;
  };
  {
    Probe pr(m, s);
    LAUNCH_TRY(pr.begin());
    LAUNCH_TRY(launch_embed_gather_rows(m->tok.as<int64_t>(), m->emb_table.as<float>(), x, B, L, D, c.n_tokens, pos, s,
                                        /*compact=*/true));
    LAUNCH_TRY(pr.end(ED_K_EMBED, 0.0, (double)B * (8 + D * 4.0)));
  }
  for (int k = 0; k < c.n_layers; ++k) {
    const LayerW& w = m->layers[k];
    if (gemv) {
      float* y1 = (float*)Y;
      float* y2 = (float*)A;
      ArGemv g1{x, D, w.ln1_g, w.ln1_b, D, act, w.w1_h, w.b1, nullptr, y1, B, H, 1, 1, nullptr, 0, 0, PosBit{}};
      ArGemv gc{y1, H, w.ln2_g, w.ln2_b, H, act, w.wc_h, w.bc, nullptr, y2, B, H, c.kernel_size, w.dilation,
                m->Car[k].as<bf16_t>(), m->conv_pitch, m->conv_row0, pos};
      ArGemv g2{y2, H, w.ln3_g, w.ln3_b, H, act, w.w2_h, w.b2, x, x, B, D, 1, 1, nullptr, 0, 0, PosBit{}};
      if (!ar_gemv_supported(g1) || !ar_gemv_supported(gc) || !ar_gemv_supported(g2))
        return fail(ED_ESTATE, "autoregressive weight-streaming path: unsupported geometry (set ED_OPT_AR_GEMV = 0)");
      if ((rc = run_gemv(g1, ED_K_GEMM1X1)) || (rc = run_gemv(gc, ED_K_CONV_OTHER)) || (rc = run_gemv(g2, ED_K_GEMM1X1))) return rc;
      continue;
    }
    if (!ln1_done && (rc = run_ln(m, x, false, A, h, w.ln1_g, w.ln1_b, B, D, LnRowMap::plain(B), act, s))) return rc;
    ln1_done = false;
    GemmProblem g1{A, B, 0, h ? (const void*)w.w1_h : (const void*)w.w1_f, w.b1, Y, nullptr, 1, B, H, D, 1, 1, h, h};
    GemmProblem dq{};
    if ((rc = run_gemm(m, g1, ED_K_GEMM1X1, s, &dq))) return rc;
    const LnRowMap to_cache{1, 1, 0, 1, PosBit{}, m->conv_pitch, m->conv_row0, pos};
    if ((rc = run_ln_after(m, dq, Y, h, m->Car[k].p, h, w.ln2_g, w.ln2_b, B, H, to_cache, act, s))) return rc;
    GemmProblem gc{m->Car[k].p, m->conv_pitch, m->conv_row0, h ? (const void*)w.wc_h : (const void*)w.wc_f, w.bc, Y,
                   nullptr, 1, B, H, H, c.kernel_size, w.dilation, h, h};
    gc.tap_center = c.kernel_size - 1;
    gc.seq_rows = 1;
    gc.a_unbounded = true;   // rows p - tau*d >= -halo: real zero padding in front of every sequence
    gc.a_sel = pos;
    gc.out_pitch = 1;
    if ((rc = run_gemm(m, gc, ED_K_CONV, s, &dq))) return rc;
    if ((rc = run_ln_after(m, dq, Y, h, A, h, w.ln3_g, w.ln3_b, B, H, LnRowMap::plain(B), act, s))) return rc;
    GemmProblem g2{A, B, 0, h ? (const void*)w.w2_h : (const void*)w.w2_f, w.b2, x, x, 1, B, D, H, 1, 1, h, false};
    GemmProblem dq2{};
    if ((rc = run_gemm(m, g2, ED_K_GEMM1X1, s, h && k + 1 < c.n_layers ? &dq2 : nullptr))) return rc;
    if (dq2.ksplit > 1) {
      const LayerW& wn = m->layers[k + 1];
      if ((rc = run_ln_after(m, dq2, nullptr, false, A, h, wn.ln1_g, wn.ln1_b, B, D, LnRowMap::plain(B), act, s))) return rc;
      ln1_done = true;
    }
  }
  Probe pr(m, s);
  LAUNCH_TRY(pr.begin());
  LAUNCH_TRY(launch_final_oadm(x, 1, 30, m->last_g, m->last_b, m->dec_w, m->dec_b, m->tok.as<int64_t>(),
                               m->order.as<int>(), m->uni.as<float>(), m->state.as<StepState>(), probs, B, L, D, n, s,
                               /*write_shift=*/1, /*advance=*/B <= ED_ADVANCE_IN_TAIL_MAX_B,
                               m->ar_forced ? m->ar_ndraw.as<int>() : nullptr,
                               m->ar_forced ? m->ar_keep.as<uint8_t>() : nullptr));
  LAUNCH_TRY(pr.end(ED_K_FINAL, 2.0 * B * n * D, (double)B * D * 4.0));
  if (B > ED_ADVANCE_IN_TAIL_MAX_B) LAUNCH_TRY(launch_advance_step(m->state.as<StepState>(), s));
  return ED_OK;
}

// One sampling step on the model-owned buffers (tok/order/uni/qbuf/state).
int run_step(ed_model* m, int mode, int variant, int B, int L, int n, int prec, float* probs, double* post,
             hipStream_t s) {
  const ed_config& c = m->cfg;
  int rc;
  if (mode == MODE_AR) return run_ar_step(m, B, L, n, prec, probs, s);
  if (m->n_inc == 0) variant = STEP_PLAIN;
  if ((rc = run_embed(m, m->tok.as<int64_t>(), nullptr, B, L, variant, s))) return rc;
  const bool pruned = mode == MODE_OADM && m->plan_s_final > 1;
  const float* xf = nullptr;
  if ((rc = run_blocks(m, B, L, prec, nullptr, pruned, variant, &xf, s))) return rc;
  Probe pr(m, s);
  LAUNCH_TRY(pr.begin());
  if (mode == MODE_OADM) {
    const int sf = pruned ? m->plan_s_final : 1;
    LAUNCH_TRY(launch_final_oadm(xf, L / sf, __builtin_ctz(sf), m->last_g, m->last_b, m->dec_w, m->dec_b,
                                 m->tok.as<int64_t>(), m->order.as<int>(), m->uni.as<float>(),
                                 m->state.as<StepState>(), probs, B, L, c.d_model, n, s, 0,
                                 /*advance=*/B <= ED_ADVANCE_IN_TAIL_MAX_B));
    LAUNCH_TRY(pr.end(ED_K_FINAL, 2.0 * B * n * c.d_model, (double)B * c.d_model * 4.0));
    if (B <= ED_ADVANCE_IN_TAIL_MAX_B) return ED_OK;
  } else {
    const size_t kk = (size_t)c.timesteps * n * n;
    LAUNCH_TRY(launch_final_d3pm(m->X.as<float>(), m->last_g, m->last_b, m->dec_w, m->dec_b, m->tok.as<int64_t>(),
                                 m->qbuf.as<double>(), m->qbuf.as<double>() + kk, m->uni.as<float>(),
                                 m->state.as<StepState>(), post, B, L, c.d_model, n, c.timesteps, s));
    LAUNCH_TRY(pr.end(ED_K_FINAL, 2.0 * B * L * n * (c.d_model + 2.0 * n), (double)B * L * c.d_model * 4.0));
  }
  LAUNCH_TRY(launch_advance_step(m->state.as<StepState>(), s));
  return ED_OK;
}

int get_graph(ed_model* m, int mode, int variant, int B, int L, int n, int prec, float* probs, double* post,
              hipGraphExec_t* out) {
  // trace buffers are caller-owned: graphs are only cached for the trace-free form
  const GraphKey key{mode, B, L, prec, n, variant};
  auto it = m->graphs.find(key);
  if (it != m->graphs.end()) {
    *out = it->second;
    return ED_OK;
  }
  if (!m->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
  const bool was = m->probing_now;
  m->probing_now = false;
  int rc = run_step(m, mode, variant, B, L, n, prec, probs, post, m->cap_stream);
  m->probing_now = was;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(m->cap_stream, &graph);
  if (rc) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess) return fail(ED_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(ED_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  m->graphs[key] = exec;
  *out = exec;
  return ED_OK;
}

int run_loop(ed_model* m, int mode, int B, int L, int n, int n_steps, int prec, int use_graph, float* probs,
             double* post, hipStream_t s) {
  const bool traced = probs != nullptr || post != nullptr;
  // OA-DM with incremental leading layers: the first step of a call fills their caches, the others update them
  const bool inc = mode == MODE_OADM && m->n_inc > 0;
  hipGraphExec_t exec_first = nullptr, exec_rest = nullptr;
  if (use_graph && !traced) {
    int rc = get_graph(m, mode, mode == MODE_AR ? (m->ar_forced ? STEP_AR_SCAFFOLD : STEP_AR) : inc ? STEP_FILL : STEP_PLAIN,
                       B, L, n, prec, nullptr, nullptr, &exec_first);
    if (rc) return rc;
    exec_rest = exec_first;
    if (inc && n_steps > 1 && (rc = get_graph(m, mode, STEP_INCR, B, L, n, prec, nullptr, nullptr, &exec_rest))) return rc;
  }
  for (int step = 0; step < n_steps; ++step) {
    const bool probe = m->probe_every > 0 && (step % m->probe_every) == 0;
    hipGraphExec_t exec = step == 0 ? exec_first : exec_rest;
    if (exec && !probe) {
      HIP_TRY(hipGraphLaunch(exec, s));
    } else {
      m->probing_now = probe;
      int rc = run_step(m, mode, !inc ? STEP_PLAIN : (step == 0 ? STEP_FILL : STEP_INCR), B, L, n, prec, probs, post, s);
      m->probing_now = false;
      if (rc) return rc;
    }
  }
  return ED_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int ed_abi_version(void) { return ED_ABI_VERSION; }
const char* ed_last_error(void) { return g_err.c_str(); }

int ed_model_create(const ed_config* cfg, int device, ed_model** out) {
  if (!cfg || !out) return fail(ED_EINVAL, "null argument");
  const ed_config& c = *cfg;
  if (c.n_tokens < 2 || c.n_tokens > 64) return fail(ED_EINVAL, "n_tokens must be in [2, 64]");
  if (c.d_embed < 1 || c.d_embed > 16) return fail(ED_EINVAL, "d_embed must be in [1, 16]");
  if (c.d_model < 32 || c.d_model > 2048 || c.d_model % 32) return fail(ED_EINVAL, "d_model must be a multiple of 32 in [32, 2048]");
  if (c.d_hidden < 32 || c.d_hidden > 2048 || c.d_hidden % 32) return fail(ED_EINVAL, "d_hidden must be a multiple of 32 in [32, 2048]");
  if (c.n_layers < 1 || c.n_layers > 1024) return fail(ED_EINVAL, "n_layers out of range");
  if (c.kernel_size < 1 || c.kernel_size > 9 || !(c.kernel_size & 1)) return fail(ED_EINVAL, "kernel_size must be odd, <= 9");
  if (c.r < 1 || c.r > 4096) return fail(ED_EINVAL, "r out of range");
  if (c.timesteps < 0 || c.timesteps > 100000) return fail(ED_EINVAL, "timesteps out of range");
  if (c.activation != 0 && c.activation != 1) return fail(ED_EINVAL, "activation must be 0 (relu) or 1 (gelu)");
  if (c.causal != 0 && c.causal != 1) return fail(ED_EINVAL, "causal must be 0 or 1");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(ED_EINVAL, "device %d not present (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  ed_model* m = new (std::nothrow) ed_model();
  if (!m) return fail(ED_ENOMEM, "out of host memory");
  m->cfg = c;
  m->device = device;
  (void)hipDeviceGetAttribute(&m->n_cus, hipDeviceAttributeMultiprocessorCount, device);
  hipError_t e = hipHostMalloc((void**)&m->ln_err_host, ED_AERR_WORDS * sizeof(unsigned), hipHostMallocMapped);
  if (e == hipSuccess) {
    std::memset(m->ln_err_host, 0, ED_AERR_WORDS * sizeof(unsigned));
    e = hipHostGetDevicePointer((void**)&m->ln_err_dev, m->ln_err_host, 0);
  }
  if (e != hipSuccess) {
    delete m;
    return fail(ED_EHIP, "host-visible error words: %s", hipGetErrorString(e));
  }
  *out = m;
  return ED_OK;
}

void ed_model_destroy(ed_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  delete m;
}

int ed_model_load_tensor(ed_model* m, const char* key, const float* host, int64_t numel) {
  if (!m || !key || !host) return fail(ED_EINVAL, "null argument");
  const std::string k(key);
  int64_t want = -1;
  if (k == "time_encoding.table") want = (int64_t)m->cfg.timesteps * m->cfg.d_embed;
  else
    for (const auto& t : expected_tensors(m->cfg))
      if (t.key == k) { want = t.numel; break; }
  if (want < 0) return fail(ED_EINVAL, "unexpected tensor key '%s'", key);
  if (want != numel) return fail(ED_EINVAL, "tensor '%s' has %lld elements, expected %lld", key, (long long)numel, (long long)want);
  m->staged[k].assign(host, host + numel);
  m->committed = false;
  return ED_OK;
}

int ed_model_commit(ed_model* m) {
  if (!m) return fail(ED_EINVAL, "null model handle");
  const ed_config& c = m->cfg;
  HIP_TRY(hipSetDevice(m->device));
  const auto specs = expected_tensors(c);
  for (const auto& t : specs)
    if (!m->staged.count(t.key)) return fail(ED_ESTATE, "missing tensor '%s'", t.key.c_str());
  if (c.timesteps > 0 && !m->staged.count("time_encoding.table")) return fail(ED_ESTATE, "missing tensor 'time_encoding.table'");

  const int64_t D = c.d_model, H = c.d_hidden, KS = c.kernel_size;
  // layout pass: fp32 image holds everything; bf16 image holds the three matrices per layer
  int64_t n32 = 0, n16 = 0;
  auto take32 = [&](int64_t n) { int64_t o = n32; n32 += align64(n); return o; };
  auto take16 = [&](int64_t n) { int64_t o = n16; n16 += align64(n); return o; };
  std::vector<float> img32;
  std::vector<bf16_t> img16;
  struct Place { int64_t off; const std::vector<float>* src; };
  std::vector<Place> vec_places;
  auto stage_vec = [&](const std::string& key) {
    const auto& v = m->staged[key];
    int64_t o = take32((int64_t)v.size());
    vec_places.push_back({o, &v});
    return o;
  };
  m->n_params = 0;
  for (const auto& t : specs) m->n_params += t.numel;

  const int64_t o_emb = stage_vec("embedder.embedder.weight");
  const int64_t o_upw = stage_vec("embedder.up_embedder.conv.weight");
  const int64_t o_upb = stage_vec("embedder.up_embedder.conv.bias");
  const int64_t o_lg = stage_vec("last_norm.weight");
  const int64_t o_lb = stage_vec("last_norm.bias");
  const int64_t o_dw = stage_vec("decoder.conv.weight");
  const int64_t o_db = stage_vec("decoder.conv.bias");
  const int64_t o_pe = c.timesteps > 0 ? stage_vec("time_encoding.table") : -1;
  struct LOff { int64_t ln1g, ln1b, b1, ln2g, ln2b, bc, ln3g, ln3b, b2, w1, wc, w2, w1h, wch, w2h; };
  std::vector<LOff> lo(c.n_layers);
  for (int i = 0; i < c.n_layers; ++i) {
    const std::string p = "embedder.layers." + std::to_string(i) + ".";
    LOff& o = lo[i];
    o.ln1g = stage_vec(p + "sequence1.0.weight");
    o.ln1b = stage_vec(p + "sequence1.0.bias");
    o.b1 = stage_vec(p + "sequence1.2.conv.bias");
    o.ln2g = stage_vec(p + "sequence1.3.weight");
    o.ln2b = stage_vec(p + "sequence1.3.bias");
    o.bc = stage_vec(p + "conv.bias");
    o.ln3g = stage_vec(p + "sequence2.0.weight");
    o.ln3b = stage_vec(p + "sequence2.0.bias");
    o.b2 = stage_vec(p + "sequence2.2.conv.bias");
    o.w1 = stage_vec(p + "sequence1.2.conv.weight");   // (H, D, 1) == [N=H][K=D]
    o.w2 = stage_vec(p + "sequence2.2.conv.weight");   // (D, H, 1) == [N=D][K=H]
    o.wc = take32(H * H * KS);                          // repacked below: (H, H, KS) -> [tap][N][K]
    o.w1h = take16(H * D);
    o.w2h = take16(D * H);
    o.wch = take16(H * H * KS);
  }
  img32.assign((size_t)n32, 0.f);
  img16.assign((size_t)n16, 0);
  for (const auto& pl : vec_places) std::memcpy(img32.data() + pl.off, pl.src->data(), pl.src->size() * 4);
  for (int i = 0; i < c.n_layers; ++i) {
    const std::string p = "embedder.layers." + std::to_string(i) + ".";
    const auto& wc = m->staged[p + "conv.weight"];
    float* dst = img32.data() + lo[i].wc;
    for (int64_t o = 0; o < H; ++o)
      for (int64_t ch = 0; ch < H; ++ch)
        for (int64_t t = 0; t < KS; ++t) dst[(t * H + o) * H + ch] = wc[(o * H + ch) * KS + t];
    const float* s1 = img32.data() + lo[i].w1;
    const float* s2 = img32.data() + lo[i].w2;
    for (int64_t j = 0; j < H * D; ++j) img16[lo[i].w1h + j] = f32_to_bf16_bits(s1[j]);
    for (int64_t j = 0; j < D * H; ++j) img16[lo[i].w2h + j] = f32_to_bf16_bits(s2[j]);
    for (int64_t j = 0; j < H * H * KS; ++j) img16[lo[i].wch + j] = f32_to_bf16_bits(dst[j]);
  }
  HIP_TRY(m->w32.ensure((size_t)n32 * 4));
  HIP_TRY(m->w16.ensure((size_t)n16 * 2 + 256));
  HIP_TRY(hipMemcpy(m->w32.p, img32.data(), (size_t)n32 * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m->w16.p, img16.data(), (size_t)n16 * 2, hipMemcpyHostToDevice));
  const float* b32 = m->w32.as<float>();
  const bf16_t* b16 = m->w16.as<bf16_t>();
  m->emb = b32 + o_emb; m->up_w = b32 + o_upw; m->up_b = b32 + o_upb;
  m->last_g = b32 + o_lg; m->last_b = b32 + o_lb; m->dec_w = b32 + o_dw; m->dec_b = b32 + o_db;
  m->pe = o_pe >= 0 ? b32 + o_pe : nullptr;
  const auto dil = dilations(c);
  m->layers.resize(c.n_layers);
  for (int i = 0; i < c.n_layers; ++i) {
    const LOff& o = lo[i];
    m->layers[i] = LayerW{dil[i], b32 + o.ln1g, b32 + o.ln1b, b32 + o.b1, b32 + o.ln2g, b32 + o.ln2b, b32 + o.bc,
                          b32 + o.ln3g, b32 + o.ln3b, b32 + o.b2, b32 + o.w1, b32 + o.wc, b32 + o.w2,
                          b16 + o.w1h, b16 + o.wch, b16 + o.w2h};
  }
  if (c.timesteps == 0) {
    // the embedded row of each token, computed by the embedding kernel itself (same arithmetic as a per-row call)
    std::vector<int64_t> ids((size_t)c.n_tokens);
    for (int i = 0; i < c.n_tokens; ++i) ids[i] = i;
    DevBuf ids_dev;
    HIP_TRY(ids_dev.ensure(ids.size() * 8));
    HIP_TRY(m->emb_table.ensure((size_t)c.n_tokens * c.d_model * 4));
    HIP_TRY(hipMemcpy(ids_dev.p, ids.data(), ids.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(launch_embed(ids_dev.as<int64_t>(), nullptr, nullptr, m->emb, nullptr, 0, m->up_w, m->up_b,
                         m->emb_table.as<float>(), c.n_tokens, 1, c.d_model, c.d_embed, c.n_tokens, nullptr));
    HIP_TRY(hipDeviceSynchronize());
  }
  m->staged.clear();
  m->drop_graphs();
  m->committed = true;
  return ED_OK;
}

int64_t ed_model_num_params(const ed_model* m) { return m ? m->n_params : 0; }

int ed_model_get_config(const ed_model* m, ed_config* out) {
  if (!m || !out) return fail(ED_EINVAL, "null argument");
  *out = m->cfg;
  return ED_OK;
}

int ed_forward_collect(ed_model* m, const int64_t* tokens_dev, const int64_t* timesteps_dev, int B, int L,
                       float* logits_dev, float* acts_dev, int precision, void* stream) {
  int rc = check_shape(m, B, L, precision);
  if (rc) return rc;
  if (!tokens_dev || !logits_dev) return fail(ED_EINVAL, "null tokens/logits pointer");
  if (m->cfg.timesteps > 0 && !timesteps_dev) return fail(ED_EINVAL, "timesteps_dev is required for a model with timestep encoding");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(m->device));
  if ((rc = check_async_errors(m))) return rc;
  if ((rc = ensure_workspace(m, B, L, precision, s))) return rc;
  // range checks on the device (clamp + asynchronous report): the kernels read the model's own, clamped copies
  LAUNCH_TRY(launch_prepare_ids(tokens_dev, m->tok.as<int64_t>(), (long)B * L, m->cfg.n_tokens, m->ln_err_dev, ED_AERR_TOKEN, s));
  if (m->cfg.timesteps > 0)
    LAUNCH_TRY(launch_prepare_ids(timesteps_dev, m->tvec.as<int64_t>(), B, m->cfg.timesteps, m->ln_err_dev, ED_AERR_TIMESTEP, s));
  m->probing_now = m->probe_every > 0;
  rc = run_embed(m, m->tok.as<int64_t>(), m->cfg.timesteps > 0 ? m->tvec.as<int64_t>() : nullptr, B, L, STEP_PLAIN, s);
  if (!rc && acts_dev)
    HIP_TRY(hipMemcpyAsync(acts_dev, m->X.p, (size_t)B * L * m->cfg.d_model * 4, hipMemcpyDeviceToDevice, s));
  if (!rc) rc = run_blocks(m, B, L, precision, acts_dev, false, STEP_PLAIN, nullptr, s);
  if (!rc) {
    Probe pr(m, s);
    LAUNCH_TRY(pr.begin());
    LAUNCH_TRY(launch_final_logits(m->X.as<float>(), m->last_g, m->last_b, m->dec_w, m->dec_b, logits_dev, B * L,
                                   m->cfg.d_model, m->cfg.n_tokens, s));
    LAUNCH_TRY(pr.end(ED_K_FINAL, 2.0 * B * L * m->cfg.n_tokens * m->cfg.d_model, (double)B * L * m->cfg.d_model * 4.0));
  }
  m->probing_now = false;
  return rc;
}

int ed_forward(ed_model* m, const int64_t* tokens_dev, const int64_t* timesteps_dev, int B, int L,
               float* logits_dev, int precision, void* stream) {
  return ed_forward_collect(m, tokens_dev, timesteps_dev, B, L, logits_dev, nullptr, precision, stream);
}

int ed_score_nll(ed_model* m, const int64_t* tokens_dev, const int64_t* timesteps_dev, const int64_t* targets_dev,
                 const uint8_t* select_dev, int B, int L, int n_classes, float* nll_dev, int precision, void* stream) {
  int rc = check_shape(m, B, L, precision);
  if (rc) return rc;
  if (!tokens_dev || !targets_dev || !nll_dev) return fail(ED_EINVAL, "null tokens/targets/nll pointer");
  if (m->cfg.timesteps > 0 && !timesteps_dev) return fail(ED_EINVAL, "timesteps_dev is required for a model with timestep encoding");
  if (n_classes < 1 || n_classes > m->cfg.n_tokens) return fail(ED_EINVAL, "n_classes must be in [1, n_tokens]");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(m->device));
  if ((rc = check_async_errors(m))) return rc;
  if ((rc = ensure_workspace(m, B, L, precision, s))) return rc;
  // range checks on the device (clamp + asynchronous report; the scoring tail clamps the targets itself)
  LAUNCH_TRY(launch_prepare_ids(tokens_dev, m->tok.as<int64_t>(), (long)B * L, m->cfg.n_tokens, m->ln_err_dev, ED_AERR_TOKEN, s));
  if (m->cfg.timesteps > 0)
    LAUNCH_TRY(launch_prepare_ids(timesteps_dev, m->tvec.as<int64_t>(), B, m->cfg.timesteps, m->ln_err_dev, ED_AERR_TIMESTEP, s));
  LAUNCH_TRY(launch_check_targets(targets_dev, select_dev, (long)B * L, n_classes, m->ln_err_dev, s));
  m->probing_now = m->probe_every > 0;
  rc = run_embed(m, m->tok.as<int64_t>(), m->cfg.timesteps > 0 ? m->tvec.as<int64_t>() : nullptr, B, L, STEP_PLAIN, s);
  if (!rc) rc = run_blocks(m, B, L, precision, nullptr, false, STEP_PLAIN, nullptr, s);
  if (!rc) {
    Probe pr(m, s);
    LAUNCH_TRY(pr.begin());
    LAUNCH_TRY(launch_final_nll(m->X.as<float>(), m->last_g, m->last_b, m->dec_w, m->dec_b, targets_dev, select_dev,
                                nll_dev, B * L, m->cfg.d_model, n_classes, s));
    LAUNCH_TRY(pr.end(ED_K_FINAL, 2.0 * B * L * n_classes * m->cfg.d_model, (double)B * L * m->cfg.d_model * 4.0));
  }
  m->probing_now = false;
  return rc;
}

int ed_oadm_sample(ed_model* m, int64_t* tokens_dev, const int32_t* order_dev, const float* uniforms_dev, int B,
                   int L, int n_steps, int n_draw, float* probs_dev, int precision, int use_graph, void* stream) {
  int rc = check_shape(m, B, L, precision);
  if (rc) return rc;
  if (!tokens_dev || !order_dev || !uniforms_dev) return fail(ED_EINVAL, "null tokens/order/uniforms pointer");
  if (n_steps < 0 || n_steps > (1 << 24)) return fail(ED_EINVAL, "n_steps out of range");
  if (n_draw < 1 || n_draw > m->cfg.n_tokens) return fail(ED_EINVAL, "n_draw must be in [1, n_tokens]");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(m->device));
  if ((rc = check_async_errors(m))) return rc;
  if ((rc = ensure_workspace(m, B, L, precision, s))) return rc;
  if (n_steps == 0) return ED_OK;
  {
    const size_t ob = m->order.bytes, ub = m->uni.bytes;
    HIP_TRY(m->order.ensure((size_t)n_steps * 4));
    HIP_TRY(m->uni.ensure((size_t)n_steps * B * 4));
    if (m->order.bytes != ob || m->uni.bytes != ub) m->drop_graphs();
  }
  // positions and tokens are range-checked on the device while they are copied into the model's buffers (clamp +
  // asynchronous report): no copy to the host, no synchronisation
  LAUNCH_TRY(launch_prepare_order(order_dev, m->order.as<int>(), n_steps, L, m->ln_err_dev, s));
  HIP_TRY(hipMemcpyAsync(m->uni.p, uniforms_dev, (size_t)n_steps * B * 4, hipMemcpyDeviceToDevice, s));
  LAUNCH_TRY(launch_prepare_ids(tokens_dev, m->tok.as<int64_t>(), (long)B * L, m->cfg.n_tokens, m->ln_err_dev, ED_AERR_TOKEN, s));
  LAUNCH_TRY(launch_init_state(m->state.as<StepState>(), 0, 0, s));
  if ((rc = run_loop(m, MODE_OADM, B, L, n_draw, n_steps, precision, use_graph, probs_dev, nullptr, s))) return rc;
  HIP_TRY(hipMemcpyAsync(tokens_dev, m->tok.p, (size_t)B * L * 8, hipMemcpyDeviceToDevice, s));
  return ED_OK;
}

static int ar_sample_impl(ed_model* m, int64_t* tokens_dev, const float* uniforms_dev, const uint8_t* keep_dev,
                          const int32_t* n_draw_step_dev, int B, int L, int n_steps, int n_draw, float* probs_dev,
                          int precision, int use_graph, void* stream) {
  int rc = check_shape(m, B, L, precision);
  if (rc) return rc;
  const ed_config& c = m->cfg;
  if (!c.causal) return fail(ED_ESTATE, "autoregressive sampling needs a causal model (cfg.causal = 1)");
  if (c.timesteps > 0) return fail(ED_ESTATE, "autoregressive sampling is for models without timestep encoding");
  if (!tokens_dev || !uniforms_dev) return fail(ED_EINVAL, "null tokens/uniforms pointer");
  if (n_steps < 0 || n_steps > L - 1) return fail(ED_EINVAL, "n_steps must be in [0, L-1]: step s writes position s+1");
  if (n_draw < 1 || n_draw > c.n_tokens) return fail(ED_EINVAL, "n_draw must be in [1, n_tokens]");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(m->device));
  // the shared workspace only has to hold one row per sequence here, but its convolution layout (pitch, halo) is the
  // one the caches use: size it for (B, L) like any other call
  if ((rc = check_async_errors(m))) return rc;
  if ((rc = ensure_workspace(m, B, L, precision, s))) return rc;
  if (n_steps == 0) return ED_OK;
  {
    const size_t cbytes = (size_t)B * m->conv_pitch * c.d_hidden * (precision == ED_PREC_BF16 ? 2 : 4);
    bool fresh = m->ar_B != B || m->ar_L != L || m->ar_prec != precision || (int)m->Car.size() != c.n_layers;
    if ((int)m->Car.size() != c.n_layers) m->Car.resize((size_t)c.n_layers);
    for (auto& buf : m->Car) {
      if (cbytes > buf.bytes) {
        fresh = true;
        HIP_TRY(buf.ensure(cbytes));
      }
    }
    if (fresh) {  // zero halo in front of every sequence; rows >= 0 are written before they are read
      for (auto& buf : m->Car) HIP_TRY(hipMemsetAsync(buf.p, 0, buf.bytes, s));
      m->drop_graphs();
      m->ar_B = B; m->ar_L = L; m->ar_prec = precision;
    }
    const size_t ob = m->order.bytes, ub = m->uni.bytes;
    HIP_TRY(m->order.ensure((size_t)n_steps * 4));
    HIP_TRY(m->uni.ensure((size_t)n_steps * B * 4));
    if (m->order.bytes != ob || m->uni.bytes != ub) m->drop_graphs();
  }
  m->ar_forced = keep_dev != nullptr;
  if (m->ar_forced) {  // model-owned copies: the captured graph sees stable pointers
    const size_t kb = m->ar_keep.bytes, nb = m->ar_ndraw.bytes;
    HIP_TRY(m->ar_keep.ensure((size_t)L));
    HIP_TRY(m->ar_ndraw.ensure((size_t)n_steps * 4));
    if (m->ar_keep.bytes != kb || m->ar_ndraw.bytes != nb) m->drop_graphs();
    HIP_TRY(hipMemcpyAsync(m->ar_keep.p, keep_dev, (size_t)L, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(m->ar_ndraw.p, n_draw_step_dev, (size_t)n_steps * 4, hipMemcpyDeviceToDevice, s));
  }
  LAUNCH_TRY(launch_prepare_order(nullptr, m->order.as<int>(), n_steps, L, m->ln_err_dev, s));  // step i extends position i
  HIP_TRY(hipMemcpyAsync(m->uni.p, uniforms_dev, (size_t)n_steps * B * 4, hipMemcpyDeviceToDevice, s));
  LAUNCH_TRY(launch_prepare_ids(tokens_dev, m->tok.as<int64_t>(), (long)B * L, c.n_tokens, m->ln_err_dev, ED_AERR_TOKEN, s));
  LAUNCH_TRY(launch_init_state(m->state.as<StepState>(), 0, 0, s));
  rc = run_loop(m, MODE_AR, B, L, n_draw, n_steps, precision, use_graph, probs_dev, nullptr, s);
  m->ar_forced = false;
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(tokens_dev, m->tok.p, (size_t)B * L * 8, hipMemcpyDeviceToDevice, s));
  return ED_OK;
}

int ed_ar_sample(ed_model* m, int64_t* tokens_dev, const float* uniforms_dev, int B, int L, int n_steps, int n_draw,
                 float* probs_dev, int precision, int use_graph, void* stream) {
  return ar_sample_impl(m, tokens_dev, uniforms_dev, nullptr, nullptr, B, L, n_steps, n_draw, probs_dev, precision,
                        use_graph, stream);
}

int ed_ar_sample_scaffold(ed_model* m, int64_t* tokens_dev, const float* uniforms_dev, const uint8_t* given_dev,
                          const int32_t* n_draw_step_dev, int B, int L, int n_steps, int n_draw_max, float* probs_dev,
                          int precision, int use_graph, void* stream) {
  if (!given_dev || !n_draw_step_dev) return fail(ED_EINVAL, "null given/n_draw_step pointer");
  return ar_sample_impl(m, tokens_dev, uniforms_dev, given_dev, n_draw_step_dev, B, L, n_steps, n_draw_max, probs_dev,
                        precision, use_graph, stream);
}

int ed_d3pm_sample(ed_model* m, int64_t* tokens_dev, const double* Q_dev, const double* Qbar_dev,
                   const float* uniforms_dev, int B, int L, int K, int t_start, int n_steps, double* post_dev,
                   int precision, int use_graph, void* stream) {
  int rc = check_shape(m, B, L, precision);
  if (rc) return rc;
  const int T = m->cfg.timesteps;
  if (T <= 1) return fail(ED_ESTATE, "model was created without timestep encoding (cfg.timesteps = %d)", T);
  if (!tokens_dev || !Q_dev || !Qbar_dev || !uniforms_dev) return fail(ED_EINVAL, "null tokens/Q/Qbar/uniforms pointer");
  if (K < 7 || K > m->cfg.n_tokens || K > 62) return fail(ED_EINVAL, "K must be in [7, n_tokens]");
  if (t_start < 1 || t_start > T - 1) return fail(ED_EINVAL, "t_start must be in [1, timesteps-1]");
  if (n_steps < 0 || n_steps > t_start) return fail(ED_EINVAL, "n_steps must be in [0, t_start] (t runs down to 1)");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(m->device));
  if ((rc = check_async_errors(m))) return rc;
  if ((rc = ensure_workspace(m, B, L, precision, s))) return rc;
  if (n_steps == 0) return ED_OK;
  const size_t kk = (size_t)T * K * K;
  {
    const size_t qb = m->qbuf.bytes, ub = m->uni.bytes;
    HIP_TRY(m->qbuf.ensure(2 * kk * 8));
    HIP_TRY(m->uni.ensure((size_t)T * B * L * 4));
    if (m->qbuf.bytes != qb || m->uni.bytes != ub) m->drop_graphs();
  }
  HIP_TRY(hipMemcpyAsync(m->qbuf.p, Q_dev, kk * 8, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(m->qbuf.as<double>() + kk, Qbar_dev, kk * 8, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(m->uni.p, uniforms_dev, (size_t)T * B * L * 4, hipMemcpyDeviceToDevice, s));
  // x_t indexes Q's columns: range-checked on the device while it is copied (clamp + asynchronous report)
  LAUNCH_TRY(launch_prepare_ids(tokens_dev, m->tok.as<int64_t>(), (long)B * L, K, m->ln_err_dev, ED_AERR_TOKEN, s));
  LAUNCH_TRY(launch_init_state(m->state.as<StepState>(), 0, t_start, s));
  if ((rc = run_loop(m, MODE_D3PM, B, L, K, n_steps, precision, use_graph, nullptr, post_dev, s))) return rc;
  HIP_TRY(hipMemcpyAsync(tokens_dev, m->tok.p, (size_t)B * L * 8, hipMemcpyDeviceToDevice, s));
  return ED_OK;
}

static long long* g_stamp_buf = nullptr;
static int64_t g_stamp_cap = 0;

int ed_debug_set_stamp_buffer(int64_t* stamps_dev, int64_t n) {
  g_stamp_buf = (long long*)stamps_dev;
  g_stamp_cap = stamps_dev ? n : 0;
  return ED_OK;
}

int ed_debug_gemm(const void* A_dev, int64_t a_seq_pitch, int64_t a_row0, const void* W_dev, const float* bias_dev,
                  void* out_dev, const float* resid_dev, int B, int L, int N, int K, int ntaps, int dilation,
                  int a_bf16, int out_bf16, int kernel, void* stream) {
  if (!A_dev || !W_dev || !bias_dev || !out_dev) return fail(ED_EINVAL, "null pointer");
  if (B <= 0 || L <= 0 || N <= 0 || K <= 0 || ntaps < 1 || !(ntaps & 1) || dilation < 1)
    return fail(ED_EINVAL, "bad GEMM shape");
  if (resid_dev && out_bf16) return fail(ED_EINVAL, "a residual add requires fp32 output");
  if (K % (a_bf16 ? 32 : 16)) return fail(ED_EINVAL, "K must be a multiple of %d", a_bf16 ? 32 : 16);
  GemmProblem p{A_dev, (long)a_seq_pitch, (long)a_row0, W_dev, bias_dev, out_dev, resid_dev, B, L, N, K, ntaps,
                dilation, a_bf16 != 0, out_bf16 != 0};
  hipStream_t s = (hipStream_t)stream;
  if (kernel >= 3 && kernel < 100) {  // diagnostic ablations of the tuned kernel (3: no MFMAs, 4: no DMA); results are not valid
    p.ablate = kernel - 2;
    kernel = 2;
    // the no-MFMA / no-copy forms exist with fp32 stores and no residual only: a bf16 output buffer is half as large
    if ((p.ablate == 1 || p.ablate == 2) && (out_bf16 || resid_dev))
      return fail(ED_EINVAL, "kernels 3 and 4 (ablated forms) store fp32 and take no residual: pass out_bf16 = 0, resid = NULL");
    if (p.ablate == 7) {  // in-kernel clock stamps: eight values per 256-row tile
      const int64_t tiles = (int64_t)B * ((L + 255) / 256) * ((N + 255) / 256);
      if (!g_stamp_buf || g_stamp_cap < 8 * tiles) return fail(ED_EINVAL, "stamp buffer missing or too small");
      p.stamps = g_stamp_buf;
    }
  }
  if (kernel >= 500) {  // 500 + diagnostic bits: the fused-LayerNorm kernel (as 400 + bits) with in-kernel clock stamps
    const int64_t tiles = (int64_t)B * ((L + 255) / 256) * ((N + 255) / 256);
    if (!g_stamp_buf || g_stamp_cap < 8 * tiles) return fail(ED_EINVAL, "stamp buffer missing or too small");
    p.stamps = g_stamp_buf;
    p.ablate = 7;
    kernel -= 100;
  }
  if (kernel >= 400) {  // tuned kernel with a LayerNorm + GELU (gamma = 1, beta = 0) in its epilogue; 400 + diagnostic bits
    static DevBuf dbg_ln_out, dbg_ln_stats, dbg_ln_ctr, dbg_ln_vec;
    static unsigned* dbg_err_host = nullptr;
    static unsigned* dbg_err_dev = nullptr;
    if (!gemm_bf16_ln_supported(p)) return fail(ED_EINVAL, "kernel 400+: shape outside the fused-LayerNorm contract");
    const size_t rows = (size_t)B * ((L + 255) / 256) * 256 + 256;   // every row of a ragged 256-row tile has a slot
    const size_t dbg_tn = (size_t)(N % 320 == 0 ? N / 320 : N / 256);
    if (dbg_tn > 8) return fail(ED_EINVAL, "kernel 400+: at most 8 column tiles per row tile");
    HIP_TRY(dbg_ln_out.ensure(rows * N * 2));
    HIP_TRY(dbg_ln_stats.ensure(3 * rows * 8 * 8));
    const int kind = ntaps > 1 ? 1 : resid_dev ? 2 : 0;
    // one set of counters (and statistics) per kind of GEMM.  A counter must always see the same number of column
    // tiles and the sets' offsets depend on the shape: any change of geometry starts from zeroed counters again.
    static size_t dbg_rows = 0, dbg_last_tn = 0;
    if (dbg_ln_ctr.bytes < 3 * (rows / 128) * 4 || rows != dbg_rows || dbg_tn != dbg_last_tn) {
      HIP_TRY(hipStreamSynchronize(s));
      HIP_TRY(dbg_ln_ctr.ensure(3 * (rows / 128) * 4));
      HIP_TRY(hipMemsetAsync(dbg_ln_ctr.p, 0, dbg_ln_ctr.bytes, s));
      HIP_TRY(hipStreamSynchronize(s));
      dbg_rows = rows;
      dbg_last_tn = dbg_tn;
    }
    if (dbg_ln_vec.bytes < (size_t)N * 8) {
      HIP_TRY(dbg_ln_vec.ensure((size_t)N * 8));
      std::vector<float> gb((size_t)2 * N, 0.f);
      std::fill(gb.begin(), gb.begin() + N, 1.f);
      HIP_TRY(hipMemcpy(dbg_ln_vec.p, gb.data(), gb.size() * 4, hipMemcpyHostToDevice));
    }
    if (!dbg_err_host) {
      HIP_TRY(hipHostMalloc((void**)&dbg_err_host, 64, hipHostMallocMapped));
      *dbg_err_host = 0;
      HIP_TRY(hipHostGetDevicePointer((void**)&dbg_err_dev, dbg_err_host, 0));
    }
    p.ln.gamma = dbg_ln_vec.as<float>();
    p.ln.beta = dbg_ln_vec.as<float>() + N;
    p.ln.out = dbg_ln_out.as<bf16_t>();
    p.ln.stats = (char*)dbg_ln_stats.p + (size_t)kind * rows * 8 * 8;
    p.ln.counters = dbg_ln_ctr.as<unsigned>() + (size_t)kind * (rows / 128);
    p.ln.error = dbg_err_dev;
    p.ln.dbg = kernel - 400;
    LAUNCH_TRY(launch_gemm_bf16(p, s));
  } else if (kernel >= 300) {  // tuned kernel with its K loop cut kernel - 300 ways + the reduction
    static DevBuf dbg_partials2;
    p.ksplit = kernel - 300;
    if (!gemm_bf16_supported(p) || p.ksplit < 2 || p.ksplit > 16) return fail(ED_EINVAL, "kernel 300 + ksplit: tuned shape, ksplit in [2, 16]");
    HIP_TRY(dbg_partials2.ensure((size_t)p.ksplit * B * L * N * 4));
    p.partial = dbg_partials2.as<float>();
    LAUNCH_TRY(launch_gemm_bf16(p, s));
    LAUNCH_TRY(launch_reduce_partials(p, s));
  } else if (kernel >= 100) {  // small-tile kernel with its K loop cut kernel - 100 ways + the reduction (bf16 only); stamps if set
    static DevBuf dbg_partials;
    if (kernel >= 200) {  // 200 + ksplit: the same on 64 x 64 tiles
      p.tile = 64;
      kernel -= 100;
    }
    p.ksplit = kernel - 100;
    if (!p.a_bf16 || p.ksplit < 1 || p.ksplit > 16) return fail(ED_EINVAL, "kernel 100 + ksplit: bf16, ksplit in [1, 16]");
    HIP_TRY(dbg_partials.ensure((size_t)p.ksplit * B * L * N * 4));
    p.partial = dbg_partials.as<float>();
    const int64_t wgs = (int64_t)((N + p.tile - 1) / p.tile) * B * ((L + p.tile - 1) / p.tile) * p.ksplit;
    if (g_stamp_buf && g_stamp_cap >= 4 * wgs) p.stamps = g_stamp_buf;
    LAUNCH_TRY(launch_gemm_generic(p, s));
    if (p.ksplit > 1) LAUNCH_TRY(launch_reduce_partials(p, s));
  } else if (kernel == 2 || (kernel == 0 && gemm_bf16_supported(p))) {
    if (!gemm_bf16_supported(p)) return fail(ED_EINVAL, "shape outside the tuned bf16 kernel's contract");
    LAUNCH_TRY(launch_gemm_bf16(p, s));
  } else {
    LAUNCH_TRY(launch_gemm_generic(p, s));
  }
  return ED_OK;
}

// ---- allocation-free entry point (no counterpart in the reference) --------------------------------------------------
// The split-K form of ed_debug_gemm (kernel = 100 / 200 + ksplit) keeps its partial sums in a library-allocated buffer.
// Callers that cannot call hipMalloc (under hipGraph capture, tight serving loops) get the same computation with the
// partial sums in a file-scope device array instead: ED_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends.
// `scratch` is a host-side stand-in for the array, not a device address: host code only ever uses the per-device
// pointer hipGetSymbolAddress returns (each GPU has its own copy of the array at its own address).
static __device__ __align__(256) unsigned char scratch[ED_NOALLOC_SCRATCH_BYTES];
constexpr int ED_MAX_DEVICES = 64;
static float* g_scratch_ptr[ED_MAX_DEVICES] = {nullptr};  // device address of `scratch`, per device (set once)
static std::mutex g_scratch_mu;

// Device address of this device's copy of `scratch`; looked up once per device, with that device current.
static int scratch_for_current_device(float** out) {
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= ED_MAX_DEVICES) return fail(ED_EINVAL, "device %d is outside the table of %d devices", dev, ED_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p)
      return fail(ED_EHIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
    g_scratch_ptr[dev] = (float*)p;
  }
  *out = g_scratch_ptr[dev];
  return ED_OK;
}

int ed_debug_gemm_splitk_noalloc(const void* A_dev, int64_t a_seq_pitch, int64_t a_row0, const void* W_dev,
                                 const float* bias_dev, void* out_dev, const float* resid_dev, int B, int L, int N, int K,
                                 int ntaps, int dilation, int out_bf16, int tile, int ksplit, void* stream) {
  if (!A_dev || !W_dev || !bias_dev || !out_dev) return fail(ED_EINVAL, "null pointer");
  if (B <= 0 || L <= 0 || N <= 0 || K <= 0 || ntaps < 1 || !(ntaps & 1) || dilation < 1)
    return fail(ED_EINVAL, "bad GEMM shape");
  if (resid_dev && out_bf16) return fail(ED_EINVAL, "a residual add requires fp32 output");
  if (K % 32) return fail(ED_EINVAL, "K must be a multiple of 32 (bf16 operands)");
  if (N % 4) return fail(ED_EINVAL, "N must be a multiple of 4");
  if (tile != 64 && tile != 128) return fail(ED_EINVAL, "tile must be 64 or 128");
  if (ksplit < 2 || ksplit > 16) return fail(ED_EINVAL, "ksplit must be in [2, 16]");
  const unsigned long long need = 4ull * (unsigned long long)ksplit * (unsigned long long)B * (unsigned long long)L * (unsigned long long)N;
  if (need > ED_NOALLOC_SCRATCH_BYTES)
    return fail(ED_EINVAL, "ksplit * B * L * N * 4 = %llu bytes of partial sums exceed the %llu-byte static scratch "
                           "(ED_NOALLOC_SCRATCH_BYTES); use ed_debug_gemm, which allocates", need,
                (unsigned long long)ED_NOALLOC_SCRATCH_BYTES);
  float* ptr = nullptr;
  int rc = scratch_for_current_device(&ptr);
  if (rc) return rc;
  GemmProblem p{A_dev, (long)a_seq_pitch, (long)a_row0, W_dev, bias_dev, out_dev, resid_dev, B, L, N, K, ntaps,
                dilation, true, out_bf16 != 0};
  p.tile = tile;
  p.ksplit = ksplit;
  p.partial = ptr;  // every range z stores every element of its slab (zeros for an empty range): nothing to clear
  hipStream_t s = (hipStream_t)stream;
  LAUNCH_TRY(launch_gemm_generic(p, s));
  LAUNCH_TRY(launch_reduce_partials(p, s));
  return ED_OK;
}

int ed_debug_ln_act(const void* in_dev, int in_bf16, void* out_dev, int out_bf16, const float* gamma_dev,
                    const float* beta_dev, int rows, int C, int act, void* stream) {
  if (!in_dev || !out_dev || !gamma_dev || !beta_dev) return fail(ED_EINVAL, "null pointer");
  if (rows <= 0 || C <= 0 || C % 4 || C > 2048 || act < 0 || act > 2) return fail(ED_EINVAL, "bad LayerNorm shape");
  LAUNCH_TRY(launch_ln_act(in_dev, in_bf16 != 0, out_dev, out_bf16 != 0, gamma_dev, beta_dev, rows, C,
                           LnRowMap::plain(rows), act, (hipStream_t)stream));
  return ED_OK;
}

int ed_model_set_option(ed_model* m, int option, int value) {
  if (!m) return fail(ED_EINVAL, "null model handle");
  switch (option) {
    case ED_OPT_PRUNE:
      if (value != 0 && value != 1) return fail(ED_EINVAL, "ED_OPT_PRUNE takes 0 or 1");
      m->prune = value != 0;
      break;
    case ED_OPT_SPLIT_TARGET:
      if (value < 0 || value > 4096) return fail(ED_EINVAL, "ED_OPT_SPLIT_TARGET must be in [0, 4096]");
      m->split_target = value;
      break;
    case ED_OPT_INCREMENTAL:
      if (value != 0 && value != 1) return fail(ED_EINVAL, "ED_OPT_INCREMENTAL takes 0 or 1");
      m->incremental = value != 0;
      break;
    case ED_OPT_MIN_TUNED_TILES:
      if (value < 0) return fail(ED_EINVAL, "ED_OPT_MIN_TUNED_TILES must be >= 0");
      m->min_tuned_tiles = value;
      break;
    case ED_OPT_FUSE_REDUCE_LN:
      if (value != 0 && value != 1) return fail(ED_EINVAL, "ED_OPT_FUSE_REDUCE_LN takes 0 or 1");
      m->fuse_reduce_ln = value != 0;
      break;
    case ED_OPT_TUNED_SPLIT_MIN_ROWS:
      if (value < 0) return fail(ED_EINVAL, "ED_OPT_TUNED_SPLIT_MIN_ROWS must be >= 0");
      m->tuned_split_min_rows = value;
      break;
    case ED_OPT_TILE64_MAX:
      if (value < 0) return fail(ED_EINVAL, "ED_OPT_TILE64_MAX must be >= 0");
      m->tile64_max = value;
      break;
    case ED_OPT_PARTIALS_MB:
      if (value < 1 || value > 4096) return fail(ED_EINVAL, "ED_OPT_PARTIALS_MB must be in [1, 4096]");
      m->partials_mb = value;
      m->ws_B = m->ws_L = 0;
      break;
    case ED_OPT_FUSE_LN:
      if (value != 0 && value != 1) return fail(ED_EINVAL, "ED_OPT_FUSE_LN takes 0 or 1");
      m->fuse_ln = value != 0;
      break;
    case ED_OPT_AR_GEMV:
      if (value < 0 || value > AR_GEMV_MAX_B) return fail(ED_EINVAL, "ED_OPT_AR_GEMV must be in [0, 8]");
      m->ar_gemv = value;
      break;
    case ED_OPT_AR_GEMV_WGS:
      if (value < 0 || value > 4096) return fail(ED_EINVAL, "ED_OPT_AR_GEMV_WGS must be in [0, 4096]");
      m->ar_gemv_wgs = value;
      break;
    case ED_OPT_SPLIT_MIN_ITERS:
      if (value < 1 || value > 64) return fail(ED_EINVAL, "ED_OPT_SPLIT_MIN_ITERS must be in [1, 64]");
      m->split_min_iters = value;
      break;
    default:
      return fail(ED_EINVAL, "unknown option %d", option);
  }
  HIP_TRY(hipSetDevice(m->device));
  m->drop_graphs();  // the captured launch sequences depend on both options
  return ED_OK;
}

int ed_model_check_errors(ed_model* m, void* stream, int synchronize) {
  if (!m) return fail(ED_EINVAL, "null model handle");
  if (synchronize) {
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  }
  return check_async_errors(m);
}

int ed_probe_config(ed_model* m, int every_n_steps) {
  if (!m) return fail(ED_EINVAL, "null model handle");
  if (every_n_steps < 0) return fail(ED_EINVAL, "every_n_steps must be >= 0");
  m->probe_every = every_n_steps;
  return ED_OK;
}

int ed_probe_read(ed_model* m, ed_probe_stat* stats, void* stream) {
  if (!m || !stats) return fail(ED_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  if (int rc = check_async_errors(m)) return rc;
  for (int k = 0; k < ED_K_COUNT; ++k) stats[k] = ed_probe_stat{0.0, 0.0, 0.0, 0};
  for (auto& p : m->pending) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
    stats[p.kind].ms += ms;
    stats[p.kind].flops += p.flops;
    stats[p.kind].bytes += p.bytes;
    stats[p.kind].launches += 1;
    m->event_pool.push_back({p.a, p.b});
  }
  m->pending.clear();
  return ED_OK;
}

}  // extern "C"
