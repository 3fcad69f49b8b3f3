// 256x192 tiles of the ping-pong kernel (N = 768 / 2304: forward and dX layouts, fast epilogues only).
#include "gemm_pp_kernel.h"

namespace xl {

hipError_t launch_pp_192(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st) {
    return b_kmajor ? launch_pp_layout<true, true, 192>(p, epik, nblk, st) : launch_pp_layout<true, false, 192>(p, epik, nblk, st);
}

}  // namespace xl
