// Dense contractions, large shapes: the 256x256 ping-pong kernel -- forward layout (A and B K-major) instances, the grouped
// weight-gradient kernel and the dispatcher (kernel templates: gemm_pp_kernel.h; other instances: gemm_pp_nn.hip, gemm_pp_192.hip).
#include "gemm_pp_kernel.h"

namespace xl {

// Grouped weight gradients: the output tiles of up to 8 problems C_i[M_i,N_i] += A_i^T B_i (M-major operands, fp32
// atomics) dealt to the CUs of ONE launch.  Linear block order: XCD-contiguous chunks of [split z][problem][tile].
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp_group_kernel(GroupParams g) {
    const int nblk = gridDim.x, b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, pos = b >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    const int total = g.tile_start[g.count];
    const int z = L / total;
    const int t = L - z * total;
    int i = 0, otm = -1, otn = 0;
    if (g.order_n > 0) {
        const int code = g.order[t];
        i = code >> 12; otm = (code >> 6) & 63; otn = code & 63;
    } else {
#pragma unroll 1
        while (i + 1 < g.count && t >= g.tile_start[i + 1]) ++i;
    }
    const GroupProblem& pr = g.prob[i];
    GemmParams p;
    p.A = pr.A; p.B = pr.B; p.C = pr.C; p.bias = nullptr; p.residual = nullptr; p.aux = nullptr;
    p.M = pr.M; p.N = pr.N; p.K = pr.K; p.lda = pr.lda; p.ldb = pr.ldb; p.ldc = pr.ldc; p.ldr = 0; p.ldx = 0;
    p.epilogue = XL_EPI_NONE; p.out_f32 = 1; p.atomic_out = 1; p.splitk = g.splitk; p.kper = pr.kper; p.vec_epi = 0;
    p.alpha = 1.0f; p.p_drop = 0.f; p.inv_keep = 1.f; p.seed = 0; p.step_seed = nullptr;
    p.tiles_m = pr.tiles_m; p.tiles_n = pr.tiles_n; p.ablate = 0; p.trace = nullptr; p.colsum_ws = nullptr;
    p.slab = g.slab; p.tickets = g.tickets; p.vec_epi = pr.vec; p.tail_tiles = 0; p.tail_kper = 0; p.overwrite = pr.overwrite; p.slab_det = g.slab != nullptr ? 1 : 0;
    if (z * pr.kper >= pr.K) return;                 // this problem's contraction is shorter than the group's split
    const int tl = t - g.tile_start[i];
    const int kbeg = z * pr.kper;
    pp_tile<false, false, -1>(p, otm >= 0 ? otm : tl % pr.tiles_m, otm >= 0 ? otn : tl / pr.tiles_m, kbeg, min(pr.K, kbeg + pr.kper), z == 0,
                              (g.slab != nullptr || g.rmw) ? t : -1, z, (pr.K + pr.kper - 1) / pr.kper);
}

hipError_t launch_pp_group(const GroupParams& g, int nblk, hipStream_t st) {
    constexpr int lds = 131072;
    hipError_t e = hipSuccess;
    static bool attr = false;
    if (!attr) { e = hipFuncSetAttribute((const void*)gemm_bf16_pp_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL(gemm_bf16_pp_group_kernel, dim3(nblk), dim3(512), lds, st, g);
    return e;
}

hipError_t launch_pp_nt256(const GemmParams& p, int epik, int nblk, hipStream_t st) {
    return launch_pp_layout<true, true, 256>(p, epik, nblk, st);
}

hipError_t launch_pp(const GemmParams& p, int a_kmajor, int b_kmajor, int epik, int bn, int nblk, hipStream_t st, int bm) {
    if (bm == 128) return (a_kmajor && bn == 192) ? launch_pp_duo(p, b_kmajor, epik, nblk, st) : hipErrorInvalidValue;
#ifdef XL_EXPERIMENTAL
    if (bn == 192) return a_kmajor ? launch_pp_192(p, b_kmajor, epik, nblk, st) : hipErrorInvalidValue;
#else
    if (bn == 192) return hipErrorInvalidValue;          // 256x192 tiles: experimental build only (XL_EXPERIMENTAL=1)
#endif
    if (a_kmajor && b_kmajor) return launch_pp_nt256(p, epik, nblk, st);
    return launch_pp_other256(p, a_kmajor, b_kmajor, epik, nblk, st);
}

}  // namespace xl
