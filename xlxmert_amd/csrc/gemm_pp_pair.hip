// Two-problem launches of the 256x256 ping-pong kernel (gemm_pp_kernel.h PairParams): the instances the paired visual / language
// sub-blocks use -- forward layout (y = x W^T + b: plain, dropout + residual, GELU with saved derivative) and dX layout (dx = dy W:
// plain, + residual, x saved GELU derivative with fused column sums).
#include "gemm_pp_kernel.h"

namespace xl {

bool pp_pair_has_instance(int b_kmajor, int epik) {
    if (epik == XL_EPI_NONE || epik == XL_EPI_RESIDUAL) return true;
    return b_kmajor ? epik == XL_EPI_GELU_DG : epik == XL_EPI_MULAUX;
}

hipError_t launch_pp_pair(const PairParams& pp, int b_kmajor, int epik, int nblk, hipStream_t st) {
    if (b_kmajor) {
        switch (epik) {
            case XL_EPI_NONE: return launch_pp_pair_one<true, true, XL_EPI_NONE>(pp, nblk, st);
            case XL_EPI_RESIDUAL: return launch_pp_pair_one<true, true, XL_EPI_RESIDUAL>(pp, nblk, st);
            case XL_EPI_GELU_DG: return launch_pp_pair_one<true, true, XL_EPI_GELU_DG>(pp, nblk, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (epik) {
        case XL_EPI_NONE: return launch_pp_pair_one<true, false, XL_EPI_NONE>(pp, nblk, st);
        case XL_EPI_RESIDUAL: return launch_pp_pair_one<true, false, XL_EPI_RESIDUAL>(pp, nblk, st);
        case XL_EPI_MULAUX: return launch_pp_pair_one<true, false, XL_EPI_MULAUX>(pp, nblk, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace xl
