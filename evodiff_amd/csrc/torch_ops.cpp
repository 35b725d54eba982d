// Thin PyTorch-ROCm C++ extension over the C ABI (include/evodiff_hip.h): registers the hot-path entry points as
// torch operators (namespace evodiff_amd, CUDA = ROCm dispatch key) that take torch tensors, launch on PyTorch's current
// HIP stream and raise RuntimeError with ed_last_error() on failure.  It holds no kernels and no state: `handle` is the
// ed_model* (as an integer) that evodiff_amd.model.ByteNetLMTime.hip_handle() created.  Built with g++ against libtorch
// and linked to libevodiff_hip.so (__graft_entry__.build()); the ctypes binding in evodiff_amd/_lib.py is the other,
// torch-free way in.
#include "../../include/evodiff_hip.h"

#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

namespace {

ed_model* as_model(int64_t handle) {
  TORCH_CHECK(handle != 0, "evodiff_amd: null model handle");
  return reinterpret_cast<ed_model*>(static_cast<intptr_t>(handle));
}

void* current_stream(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check(int rc) { TORCH_CHECK(rc == 0, "libevodiff_hip error ", rc, ": ", ed_last_error()); }

void want(const at::Tensor& t, at::ScalarType dtype, const char* name) {
  TORCH_CHECK(t.is_cuda(), "evodiff_amd: ", name, " must live on the GPU (there is no CPU path)");
  TORCH_CHECK(t.scalar_type() == dtype && t.is_contiguous(), "evodiff_amd: ", name, " must be contiguous ", dtype);
}

ed_config config_of(int64_t handle) {
  ed_config c;
  check(ed_model_get_config(as_model(handle), &c));
  return c;
}

// (B, L) of a token tensor; every other tensor of a call is checked against them and against the model's own
// hyper-parameters, so that a caller's mistake is a RuntimeError here and never an out-of-bounds access on the device
void batch_shape(const at::Tensor& tokens, int64_t* B, int64_t* L) {
  TORCH_CHECK(tokens.dim() == 2 && tokens.size(0) > 0 && tokens.size(1) > 0, "evodiff_amd: tokens must be (batch, length)");
  *B = tokens.size(0);
  *L = tokens.size(1);
}

void want_timesteps(const ed_config& c, const at::Tensor& timesteps, int64_t B) {
  want(timesteps, at::kLong, "timesteps");
  TORCH_CHECK(c.timesteps == 0 || timesteps.numel() == B, "evodiff_amd: timesteps must hold one entry per sequence (", B,
              "), got ", timesteps.numel());
}

// ByteNetLMTime.forward(x, y) (evodiff/model.py:170-173): (B, L) int64 tokens, (B,) int64 timesteps -> (B, L, n_tokens) fp32
at::Tensor forward(int64_t handle, const at::Tensor& tokens, const at::Tensor& timesteps, int64_t n_tokens,
                   int64_t precision) {
  want(tokens, at::kLong, "tokens");
  int64_t B, L;
  batch_shape(tokens, &B, &L);
  const ed_config c = config_of(handle);
  want_timesteps(c, timesteps, B);
  TORCH_CHECK(n_tokens == c.n_tokens, "evodiff_amd: n_tokens = ", n_tokens, " but the model has ", c.n_tokens, " (it sizes the logits)");
  at::Tensor logits = at::empty({B, L, n_tokens}, tokens.options().dtype(at::kFloat));
  check(ed_forward(as_model(handle), tokens.data_ptr<int64_t>(), timesteps.data_ptr<int64_t>(), (int)B, (int)L,
                   logits.data_ptr<float>(), (int)precision, current_stream(tokens)));
  return logits;
}

// the loop of generate_oaardm (evodiff/generate.py:193-211); tokens (B, L) updated in place
at::Tensor oadm_sample_(int64_t handle, at::Tensor tokens, const at::Tensor& order, const at::Tensor& uniforms,
                        int64_t n_draw, int64_t precision, bool use_graph) {
  want(tokens, at::kLong, "tokens");
  want(order, at::kInt, "order");
  want(uniforms, at::kFloat, "uniforms");
  int64_t B, L;
  batch_shape(tokens, &B, &L);
  const int64_t n_steps = order.numel();
  TORCH_CHECK(uniforms.numel() == n_steps * B, "evodiff_amd: uniforms must be (n_steps, B)");
  check(ed_oadm_sample(as_model(handle), tokens.data_ptr<int64_t>(), order.data_ptr<int32_t>(), uniforms.data_ptr<float>(),
                       (int)B, (int)L, (int)n_steps, (int)n_draw, nullptr, (int)precision, use_graph ? 1 : 0,
                       current_stream(tokens)));
  return tokens;
}

// the loop of generate_d3pm (evodiff/generate.py:274-298); tokens (B, L) = x_t on entry, updated in place
at::Tensor d3pm_sample_(int64_t handle, at::Tensor tokens, const at::Tensor& Q, const at::Tensor& Qbar,
                        const at::Tensor& uniforms, int64_t K, int64_t t_start, int64_t n_steps, int64_t precision,
                        bool use_graph) {
  want(tokens, at::kLong, "tokens");
  want(Q, at::kDouble, "Q");
  want(Qbar, at::kDouble, "Qbar");
  want(uniforms, at::kFloat, "uniforms");
  TORCH_CHECK(Q.dim() == 3 && Q.size(1) == K && Q.size(2) == K && Qbar.sizes() == Q.sizes(), "evodiff_amd: Q / Qbar must be (timesteps, K, K)");
  int64_t B, L;
  batch_shape(tokens, &B, &L);
  TORCH_CHECK(Q.size(0) == config_of(handle).timesteps, "evodiff_amd: Q / Qbar must hold one matrix per timestep of the model");
  TORCH_CHECK(uniforms.numel() == Q.size(0) * B * L, "evodiff_amd: uniforms must be (timesteps, B, L)");
  check(ed_d3pm_sample(as_model(handle), tokens.data_ptr<int64_t>(), Q.data_ptr<double>(), Qbar.data_ptr<double>(),
                       uniforms.data_ptr<float>(), (int)B, (int)L, (int)K, (int)t_start, (int)n_steps, nullptr,
                       (int)precision, use_graph ? 1 : 0, current_stream(tokens)));
  return tokens;
}

// forward + cross entropy at the targets (analysis/sequence_perp.py:125-154): (B, L) fp32, zero where select == 0
at::Tensor score_nll(int64_t handle, const at::Tensor& tokens, const at::Tensor& timesteps, const at::Tensor& targets,
                     const c10::optional<at::Tensor>& select, int64_t n_classes, int64_t precision) {
  want(tokens, at::kLong, "tokens");
  want(targets, at::kLong, "targets");
  int64_t B, L;
  batch_shape(tokens, &B, &L);
  want_timesteps(config_of(handle), timesteps, B);
  TORCH_CHECK(targets.sizes() == tokens.sizes(), "evodiff_amd: targets must have the shape of tokens");
  if (select.has_value()) {
    want(*select, at::kByte, "select");
    TORCH_CHECK(select->numel() == B * L, "evodiff_amd: select must hold one byte per position (", B * L, ")");
  }
  at::Tensor nll = at::empty({B, L}, tokens.options().dtype(at::kFloat));
  check(ed_score_nll(as_model(handle), tokens.data_ptr<int64_t>(), timesteps.data_ptr<int64_t>(), targets.data_ptr<int64_t>(),
                     select.has_value() ? select->data_ptr<uint8_t>() : nullptr, (int)B, (int)L, (int)n_classes,
                     nll.data_ptr<float>(), (int)precision, current_stream(tokens)));
  return nll;
}

}  // namespace

TORCH_LIBRARY(evodiff_amd, m) {
  m.def("forward(int handle, Tensor tokens, Tensor timesteps, int n_tokens, int precision) -> Tensor");
  m.def("oadm_sample_(int handle, Tensor(a!) tokens, Tensor order, Tensor uniforms, int n_draw, int precision, bool use_graph) -> Tensor(a!)");
  m.def("d3pm_sample_(int handle, Tensor(a!) tokens, Tensor Q, Tensor Qbar, Tensor uniforms, int K, int t_start, int n_steps, int precision, bool use_graph) -> Tensor(a!)");
  m.def("score_nll(int handle, Tensor tokens, Tensor timesteps, Tensor targets, Tensor? select, int n_classes, int precision) -> Tensor");
}

TORCH_LIBRARY_IMPL(evodiff_amd, CUDA, m) {
  m.impl("forward", &forward);
  m.impl("oadm_sample_", &oadm_sample_);
  m.impl("d3pm_sample_", &d3pm_sample_);
  m.impl("score_nll", &score_nll);
}
