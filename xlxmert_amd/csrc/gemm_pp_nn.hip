// 256x256 ping-pong kernel: dX layout (A K-major, B M-major) and the layouts with an M-major A (generic epilogue only).
#include "gemm_pp_kernel.h"

namespace xl {

hipError_t launch_pp_other256(const GemmParams& p, int a_kmajor, int b_kmajor, int epik, int nblk, hipStream_t st) {
    if (a_kmajor && !b_kmajor) return launch_pp_layout<true, false, 256>(p, epik, nblk, st);
    if (!a_kmajor && b_kmajor) return launch_pp_layout<false, true, 256>(p, epik, nblk, st);
    return launch_pp_layout<false, false, 256>(p, epik, nblk, st);
}

}  // namespace xl
