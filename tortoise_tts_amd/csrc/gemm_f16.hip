// fp16 instantiations of the MFMA GEMM family (gemm_impl.h); split per operand type so the two halves compile in parallel.
#include "gemm_impl.h"
namespace tt {
template int gemm_launch_typed<f16>(int, const GemmArgs&, const GemmPlan&, hipStream_t);
template int gemm_init_typed<f16>();
template int gemm_gna_launch_typed<f16>(const GemmArgs&, const GemmPlan&, const GnaArgs&, hipStream_t);
}  // namespace tt
