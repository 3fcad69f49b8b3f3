// 128x192 "duo" tiles of the ping-pong kernel: four waves, 80 KiB of LDS, two workgroups per CU (see PPGeo in gemm_pp_kernel.h).
#include "gemm_pp_kernel.h"

namespace xl {

hipError_t launch_pp_duo(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st) {
    return b_kmajor ? launch_pp_layout<true, true, 192, 128>(p, epik, nblk, st) : launch_pp_layout<true, false, 192, 128>(p, epik, nblk, st);
}

}  // namespace xl
