// Tuned bf16 implicit-GEMM kernel for gfx950 (placeholder until the first parity run of the generic kernel).
#include "kernels.hpp"

namespace ed {

bool gemm_bf16_supported(const GemmProblem&) { return false; }
hipError_t launch_gemm_bf16(const GemmProblem&, hipStream_t) { return hipErrorNotSupported; }

}  // namespace ed
