// Dense contractions of the X-LXMERT path (nn.Linear forward, dX, dW) for gfx950.
//
//   C[M,N] = alpha * sum_k A(m,k) B(n,k) (+bias) -> epilogue     (see include/xlxmert_hip.h: xl_gemm)
//
// Kernels here (the 256x256 ping-pong kernel for the large shapes lives in gemm_pp.hip):
//   gemm_bf16_mfma_kernel<AK, BKM, TR, EPIK, BM, BN, WM, WN>
//       bf16 operands, fp32 accumulate on v_mfma_f32_32x32x16_bf16.  Block tile 128 x 128 x 64 with 2 x 2 waves, two
//       workgroups per CU: the kernel for shapes with too few 256x256 tiles to fill the chip (language stream) and for
//       operands the ping-pong kernel (gemm_pp.hip) does not take.  Operands are staged through registers (16 B per
//       lane) into a double-buffered LDS ring, one barrier per K tile.  K-major operands ([rows][k]) sit row-major in LDS with a 16-byte-chunk XOR swizzle
//       (chunk ^= (row>>1)&7) so that ds_read_b128 fragment reads are conflict free; M-major operands ([k][rows];
//       the dX / dW contractions) sit as stored and become MFMA fragments through ds_read_b64_tr_b16 (LDS
//       transpose read) with a 64-byte XOR swizzle on k&3.  Because LDS-DMA writes lane-linear, both swizzles are
//       applied to the per-lane SOURCE address.  The MFMA k-slot <-> k mapping is applied identically to A and B,
//       which is all the contraction needs.
//   gemm_generic_kernel    any dtype / any alignment, fp32 FMA, 64x64x16 tile.  It is the exact-fp32
//       path (XL_F32: parity configuration) and the fallback for operands the MFMA loader cannot
//       take (leading dimension not a multiple of 8 elements).
#include <algorithm>
#include <vector>
#include <mutex>
#include <unordered_map>
#include "gemm_common.h"

namespace xl {

// ================================================================== generic fp32-FMA kernel
template <typename TIn>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmParams p, int a_kmajor, int b_kmajor) {
    constexpr int TM = 64, TN = 64, TK = 16;
    __shared__ float As[TK][TM + 4];
    __shared__ float Bs[TK][TN + 4];
    int tm, tn, z;
    tile_coords(p, tm, tn, z);
    const int m0 = tm * TM, n0 = tn * TN;
    const int kbeg = z * p.kper, kend = min(p.K, kbeg + p.kper);
    const TIn* A = reinterpret_cast<const TIn*>(p.A);
    const TIn* B = reinterpret_cast<const TIn*>(p.B);
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;       // 16x16 threads, 4x4 outputs each
    float acc[4][4] = {};
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256;
            int mi, ki;
            if (a_kmajor) { ki = idx & 15; mi = idx >> 4; } else { mi = idx & 63; ki = idx >> 6; }
            const int gm = m0 + mi, gk = k0 + ki;
            float v = 0.f;
            if (gm < p.M && gk < kend)
                v = Elem<TIn>::ld(a_kmajor ? A + (size_t)gm * p.lda + gk : A + (size_t)gk * p.lda + gm);
            As[ki][mi] = v;
            int ni, kj;
            if (b_kmajor) { kj = idx & 15; ni = idx >> 4; } else { ni = idx & 63; kj = idx >> 6; }
            const int gn = n0 + ni, gk2 = k0 + kj;
            float w = 0.f;
            if (gn < p.N && gk2 < kend)
                w = Elem<TIn>::ld(b_kmajor ? B + (size_t)gn * p.ldb + gk2 : B + (size_t)gk2 * p.ldb + gn);
            Bs[kj][ni] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < p.M && n < p.N) epilogue_store<TIn>(p, m, n, acc[i][j], z == 0);
        }
}

// ================================================================== bf16 MFMA kernel (128x128)
template <bool AK, bool BKM, bool TR, int EPIK, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16_mfma_kernel(GemmParams p) {   // >= 2 waves / SIMD
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;       // wave tile
    constexpr int FM = WTM / 32, FN = WTN / 32;       // 32x32 MFMA fragments per wave
    using TA = OpTile<AK, BM>;
    using TB = OpTile<BKM, BN>;
    constexpr int STAGE = TA::BYTES + TB::BYTES;
    constexpr int PA = TA::BYTES / 1024 / NW, PB = TB::BYTES / 1024 / NW;     // 1 KiB pieces per wave
    static_assert(NW * 16384 <= 2 * STAGE, "epilogue needs 16 KiB of LDS per wave");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];            // [2 stages][A tile | B tile]
    int tm, tn, z;
    tile_coords(p, tm, tn, z);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * p.kper, kend = min(p.K, kbeg + p.kper);
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int wm = (wave / WN) * WTM, wn = (wave % WN) * WTN;

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nkt = (kend - kbeg + BK - 1) / BK;
    const int nfull = (kend - kbeg) / BK;          // k-tiles fully in range: unpredicated loads
    // Each lane moves 16 bytes per piece; a wave's piece is 1 KiB of the LDS tile image, so the swizzles are applied to the
    // per-lane SOURCE address.  Rows / row-chunks past the operand extent are clamped (they only feed outputs that are
    // never stored).
    size_t srca[PA], srcb[PB];                      // element offsets of this lane's pieces (k0 excluded)
    const int am0 = (p.ablate & 1) ? 0 : m0, bn0 = (p.ablate & 1) ? 0 : n0;
#pragma unroll
    for (int t = 0; t < PA; ++t) {
        int rs, c;
        TA::decode((wave_u * PA + t) * 1024 + lane * 16, rs, c);
        srca[t] = AK ? (size_t)min(am0 + rs, p.M - 1) * p.lda + c * 8 : (size_t)rs * p.lda + min(am0 + c * 8, p.lda - 8);
    }
#pragma unroll
    for (int t = 0; t < PB; ++t) {
        int rs, c;
        TB::decode((wave_u * PB + t) * 1024 + lane * 16, rs, c);
        srcb[t] = BKM ? (size_t)min(bn0 + rs, p.N - 1) * p.ldb + c * 8 : (size_t)rs * p.ldb + min(bn0 + c * 8, p.ldb - 8);
    }
    uint4 ra[PA], rb[PB];                           // staging registers
    auto stage_issue = [&](int kt) {
        const int k0 = kbeg + kt * BK;
        if (kt < nfull) {
            const bf16_t* ga = A + (AK ? (size_t)k0 : (size_t)k0 * p.lda);
            const bf16_t* gb = B + (BKM ? (size_t)k0 : (size_t)k0 * p.ldb);
#pragma unroll
            for (int t = 0; t < PA; ++t) ra[t] = *reinterpret_cast<const uint4*>(ga + srca[t]);
#pragma unroll
            for (int t = 0; t < PB; ++t) rb[t] = *reinterpret_cast<const uint4*>(gb + srcb[t]);
        } else {                                      // ragged last k-tile: predicated loads (zero fill past the extents)
#pragma unroll
            for (int t = 0; t < PA; ++t) ra[t] = gload16<AK, BM>(A, p.lda, m0, p.M, k0, kend, (wave * PA + t) * 1024 + lane * 16);
#pragma unroll
            for (int t = 0; t < PB; ++t) rb[t] = gload16<BKM, BN>(B, p.ldb, n0, p.N, k0, kend, (wave * PB + t) * 1024 + lane * 16);
        }
    };
    auto stage_commit = [&](uint8_t* buf) {
#pragma unroll
        for (int t = 0; t < PA; ++t) *reinterpret_cast<uint4*>(buf + (wave * PA + t) * 1024 + lane * 16) = ra[t];
#pragma unroll
        for (int t = 0; t < PB; ++t) *reinterpret_cast<uint4*>(buf + TA::BYTES + (wave * PB + t) * 1024 + lane * 16) = rb[t];
    };
    if (nkt > 0) { stage_issue(0); stage_commit(smem); }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const uint8_t* ta = smem + (kt & 1) * STAGE;
        const uint8_t* tb = ta + TA::BYTES;
        const bool more = kt + 1 < nkt && !(p.ablate & 2);
        uint8_t* nbuf = smem + ((kt + 1) & 1) * STAGE;
        // the other buffer was last read in iteration kt-1 (barrier passed): refill it while this one is consumed
        if (more) stage_issue(kt + 1);
        if (!(p.ablate & 8))
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = TA::template frag<TR>(ta, wm + i * 32, s, lane);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = TB::template frag<TR>(tb, wn + j * 32, s, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(v8bf16_t, fa[i]), __builtin_bit_cast(v8bf16_t, fb[j]), acc[i][j], 0, 0, 0);
        }
        if (more) stage_commit(nbuf);
        __syncthreads();                              // publishes the refill
    }
    if (p.ablate & 4) return;
    // ---- epilogue
    const bool first = (z == 0);
    if (p.atomic_out) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) epilogue_atomic_frag(p, lane, first, m0 + wm + i * 32, n0 + wn + j * 32, acc[i][j]);
        return;
    }
    float* wbuf = reinterpret_cast<float*>(smem + wave * 16384);
    static_assert(FM == 2 && FN == 2, "one 64x64 quad per wave");
    const int mq = m0 + wm, nq = n0 + wn;
    if constexpr (EPIK >= 0) {
        if (mq + 64 <= p.M && nq + 64 <= p.N) {
            QuadOperand op;
            quad_operand_load<EPIK>(p, lane, mq, nq, op);
            epilogue_quad_fast<EPIK>(p, wbuf, lane, first, mq, nq, op, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
            return;
        }
    }
    epilogue_quad(p, wbuf, lane, first, mq, nq, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
}

template <bool AK, bool BKM, bool TR, int EPIK>
static void launch_one(const GemmParams& p, int nblk, hipStream_t st) {
    constexpr int lds = 2 * (128 + 128) * BK * 2;
    hipLaunchKernelGGL((gemm_bf16_mfma_kernel<AK, BKM, TR, EPIK, 128, 128, 2, 2>), dim3(nblk), dim3(256), lds, st, p);
}

// fast-epilogue instantiations exist for the layouts that carry epilogues (forward NT, dX NN); weight gradients (TN) go
// through atomics and everything else through the generic epilogue
template <bool AK, bool BKM>
static void launch_mfma(const GemmParams& p, int epik, int nblk, hipStream_t st) {
    if (!ctx().use_tr_read) return launch_one<AK, BKM, false, -1>(p, nblk, st);
    if constexpr (AK) {
        switch (epik) {
            case XL_EPI_NONE: return launch_one<AK, BKM, true, XL_EPI_NONE>(p, nblk, st);
            case XL_EPI_GELU: return launch_one<AK, BKM, true, XL_EPI_GELU>(p, nblk, st);
            case XL_EPI_RESIDUAL: return launch_one<AK, BKM, true, XL_EPI_RESIDUAL>(p, nblk, st);
            case XL_EPI_DGELU: return launch_one<AK, BKM, true, XL_EPI_DGELU>(p, nblk, st);
            case XL_EPI_GELU_DG:
                if constexpr (BKM) return launch_one<AK, BKM, true, XL_EPI_GELU_DG>(p, nblk, st);
                break;
            case XL_EPI_MULAUX:
                if constexpr (!BKM) return launch_one<AK, BKM, true, XL_EPI_MULAUX>(p, nblk, st);
                break;
            default: break;
        }
    }
    return launch_one<AK, BKM, true, -1>(p, nblk, st);
}

static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

// The kernel-choice switches, the trace buffer and the split-K slab workspaces (one per stream, xl_gemm_set_workspace:
// caller-owned memory, [16 KiB of tickets | slabs]) live in the calling thread's context (common.h Ctx).
// -> true and the pointers when the stream has a workspace for `slabs` partial tiles
static bool slab_workspace(hipStream_t st, long tiles, long slabs, float** slab, int** tickets) {
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.slab_ws.find(st);
    if (it == c.slab_ws.end() || tiles * (long)sizeof(int) > (long)SLAB_TICKET_BYTES ||
        SLAB_TICKET_BYTES + (size_t)slabs * SLAB_FLOATS * sizeof(float) > it->second.bytes) return false;
    *tickets = reinterpret_cast<int*>(it->second.ptr);
    *slab = reinterpret_cast<float*>(it->second.ptr + SLAB_TICKET_BYTES);
    return true;
}

}  // namespace xl


using namespace xl;

extern "C" int xl_set_gemm_tail_split(int max_tail_tiles, int min_k) {
    XL_CHECK_ARG(max_tail_tiles >= 0 && max_tail_tiles < 256 && min_k >= 1024, XL_ERR_BAD_ARG,
                 "xl_set_gemm_tail_split: max_tail_tiles %d (0..255), min_k %d (>= 1024)", max_tail_tiles, min_k);
    ctx().tail_max = max_tail_tiles; ctx().tail_min_k = min_k;
    return XL_OK;
}

extern "C" int xl_set_gemm_wgrad_slabs(int on) {
    ctx().wgrad_slabs = on ? 1 : 0;
    return XL_OK;
}

extern "C" int64_t xl_gemm_workspace_bytes(int slabs) {
    return (int64_t)SLAB_TICKET_BYTES + (int64_t)std::max(slabs, 0) * (int64_t)(SLAB_FLOATS * sizeof(float));
}

extern "C" int xl_gemm_set_workspace(void* ws, int64_t bytes, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Ctx& c = ctx();
    std::lock_guard<std::mutex> lk(c.mu);
    if (ws == nullptr || bytes <= 0) { c.slab_ws.erase(st); return XL_OK; }
    XL_CHECK_ARG(aligned16(ws) && bytes >= (int64_t)SLAB_TICKET_BYTES, XL_ERR_BAD_ARG,
                 "xl_gemm_set_workspace: workspace must be 16-byte aligned and hold the 16 KiB ticket block");
    hipError_t e = hipMemsetAsync(ws, 0, SLAB_TICKET_BYTES, st);          // tickets are zero between launches
    XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm_set_workspace: memset failed: %s", hipGetErrorString(e));
    c.slab_ws[st] = SlabWs{reinterpret_cast<uint8_t*>(ws), (size_t)bytes};
    return XL_OK;
}

extern "C" int xl_gemm_trace(void* buffer) {
    ctx().gemm_trace = reinterpret_cast<unsigned long long*>(buffer);
    return XL_OK;
}

extern "C" int xl_set_gemm_pingpong(int mode) {
    XL_CHECK_ARG(mode >= 0 && mode <= 2, XL_ERR_BAD_ARG, "xl_set_gemm_pingpong: mode %d", mode);
    ctx().gemm_pp = mode;
    return XL_OK;
}

#ifdef XL_EXPERIMENTAL
extern "C" int xl_set_gemm_persistent(int on) {
    ctx().gemm_persist = on ? 1 : 0;
    return XL_OK;
}
#endif

#ifdef XL_EXPERIMENTAL
extern "C" int xl_set_gemm_split_epi(int on) {
    ctx().gemm_split_epi = on ? 1 : 0;
    return XL_OK;
}
#endif

extern "C" int xl_set_gemm_duo(int mode) {
    XL_CHECK_ARG(mode >= 0 && mode <= 2, XL_ERR_BAD_ARG, "xl_set_gemm_duo: mode %d", mode);
    ctx().gemm_duo = mode;
    return XL_OK;
}

#ifdef XL_EXPERIMENTAL
extern "C" int xl_set_gemm_tile192(int mode) {
    XL_CHECK_ARG(mode >= 0 && mode <= 2, XL_ERR_BAD_ARG, "xl_set_gemm_tile192: mode %d", mode);
    ctx().gemm_bn192 = mode;
    return XL_OK;
}
#endif

extern "C" int xl_gemm(const void* A, const void* B, void* C, const float* bias,
                       const void* residual, void* aux,
                       int M, int N, int K, int lda, int ldb, int ldc, int ldr, int ldx,
                       int a_kmajor, int b_kmajor, int in_dtype, int out_dtype,
                       int epilogue, float alpha, int accumulate,
                       float p_drop, uint64_t seed, float* colsum_out, float* colsum_ws, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Ctx& cx = ctx();
    XL_CHECK_ARG(M > 0 && N > 0 && K > 0, XL_ERR_BAD_SHAPE, "xl_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    XL_CHECK_ARG(in_dtype == XL_F32 || in_dtype == XL_BF16, XL_ERR_BAD_DTYPE, "xl_gemm: bad in_dtype %d", in_dtype);
    XL_CHECK_ARG(out_dtype == in_dtype || out_dtype == XL_F32, XL_ERR_BAD_DTYPE, "xl_gemm: bad out_dtype %d", out_dtype);
    XL_CHECK_ARG(A && B && (C || epilogue == XL_EPI_ROWMAX), XL_ERR_BAD_ARG, "xl_gemm: null operand");
    XL_CHECK_ARG(lda >= (a_kmajor ? K : M) && ldb >= (b_kmajor ? K : N) && ldc >= N, XL_ERR_BAD_SHAPE,
                 "xl_gemm: leading dimension too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    XL_CHECK_ARG(epilogue >= XL_EPI_NONE && epilogue <= XL_EPI_MULAUX, XL_ERR_BAD_ARG, "xl_gemm: bad epilogue %d", epilogue);
    if (epilogue == XL_EPI_ROWMAX)
        XL_CHECK_ARG(in_dtype == XL_BF16 && a_kmajor && b_kmajor && M % 256 == 0 && N % 256 == 0 && K % 8 == 0 && lda % 8 == 0 &&
                     ldb % 8 == 0 && aux && aligned16(aux) && aligned16(A) && aligned16(B) && (!bias || aligned16(bias)) &&
                     !accumulate && !colsum_out && cx.use_tr_read, XL_ERR_BAD_SHAPE,
                     "xl_gemm: XL_EPI_ROWMAX takes bf16 K-major operands with M, N multiples of 256 (M=%d N=%d) and a 16-byte aligned aux", M, N);
    if (epilogue == XL_EPI_RESIDUAL) XL_CHECK_ARG(residual && ldr >= N, XL_ERR_BAD_ARG, "xl_gemm: residual missing");
    if (epilogue == XL_EPI_GELU || epilogue == XL_EPI_DGELU || epilogue == XL_EPI_GELU_DG || epilogue == XL_EPI_MULAUX)
        XL_CHECK_ARG(aux && ldx >= N, XL_ERR_BAD_ARG, "xl_gemm: aux missing");
    XL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, XL_ERR_BAD_ARG, "xl_gemm: p_drop %f", p_drop);
    if (accumulate) XL_CHECK_ARG(out_dtype == XL_F32 && epilogue == XL_EPI_NONE, XL_ERR_BAD_ARG,
                                 "xl_gemm: accumulate needs fp32 output and no epilogue");

    static const int ablate = env_int("XL_GEMM_ABLATE", 0);

    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.residual = residual; p.aux = aux;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldx = ldx;
    p.epilogue = epilogue; p.out_f32 = (out_dtype == XL_F32); p.alpha = alpha;
    p.p_drop = p_drop; p.inv_keep = 1.0f / (1.0f - p_drop); p.seed = seed; p.step_seed = cx.step_seed; p.ablate = ablate;
    p.trace = cx.gemm_trace;
    p.colsum_ws = nullptr;
    p.slab = nullptr; p.tickets = nullptr; p.tail_tiles = 0; p.tail_kper = 0; p.overwrite = 0; p.slab_det = 0;
    if (colsum_out != nullptr)
        XL_CHECK_ARG(colsum_ws != nullptr && !accumulate && (long)((M + 63) / 64) * N <= xl_workspace_floats(N), XL_ERR_BAD_ARG,
                     "xl_gemm: colsum_out needs a workspace (xl_workspace_floats), accumulate = 0 and M <= 262144");

    const bool mfma_ok = in_dtype == XL_BF16 && (lda % 8 == 0) && (ldb % 8 == 0) && aligned16(A) && aligned16(B);
    const bool may_split = mfma_ok && out_dtype == XL_F32 && epilogue == XL_EPI_NONE;
    // kernel choice.  Ping-pong 256x256 kernel (one workgroup per CU): XL_GEMM_PP = 0 never, 1 by shape, 2 whenever eligible
    if (cx.gemm_pp < 0) cx.gemm_pp = env_int("XL_GEMM_PP", 1);
    const int pp_mode = cx.gemm_pp;
    static const int pp_min_tiles = env_int("XL_GEMM_PP_MIN_TILES", 48);
    const long t256n = (long)((M + 255) / 256) * ((N + 255) / 256);
    const bool pp_ok = mfma_ok && cx.use_tr_read && K % 8 == 0 && (double)(a_kmajor ? M : K) * lda < 1e9 &&
                       (double)(b_kmajor ? N : K) * ldb < 1e9;      // 32-bit byte offsets below 2^31 inside the kernel
    // by shape: >= 48 tiles of 256x256 (x K splits for weight gradients).  In isolation the 128x128 kernel (two workgroups
    // per CU, 4x the tiles) is faster below ~128 tiles (tools/gemm_bench.py), but the language stream's 60-tile contractions
    // run NEXT TO the visual stream's 192-tile ones (N = 768: 3 column tiles x 64 row tiles on 256 CUs): as 60 whole-CU
    // workgroups they drop into the 64 idle CUs instead of competing for all of them (-0.5 ms per step)
    long pp_blocks = t256n;
    if (may_split && t256n < 256 && K >= 1024) pp_blocks = t256n * std::max<long>(1, std::min<long>(256 / t256n, K / 512));
    const bool use_pp = pp_ok && (pp_mode == 2 || (pp_mode == 1 && pp_blocks >= pp_min_tiles) || epilogue == XL_EPI_ROWMAX);
    const int tile = use_pp ? 256 : mfma_ok ? 128 : 64;
    p.tiles_m = (M + tile - 1) / tile;
    // 256x192 tiles (forward / dX layouts with a fast epilogue, every tile interior): taken when they shorten the launch,
    // estimated as rounds over the 256 CUs x relative tile cost (a 256x192 tile does 3/4 of the MFMA work plus the same
    // fixed prologue / epilogue latency: ~0.8).  N = 768: 64 row tiles give 192 tiles of 256x256 (a quarter of the chip
    // idle) or 256 of 256x192; N = 2304: 576 (2.25 rounds) or 768 (3 rounds of 0.8); N = 3072 stays at 256x256.
    // Default 0 since round 4: alone, N = 768 as 256 tiles of 256x192 beats 192 tiles of 256x256 (one full round instead of 3/4 of the
    // chip) and the isolated GEMM sum is 2 % better with mode 1 -- but the step runs four streams, and a main-chain launch that takes
    // every CU makes the language stream's and the weight-gradient stream's workgroups wait for a whole tile: measured on three boxes,
    // A/B/A: 17.23 -> 16.96, 17.35 -> 17.07, 17.85 -> 17.48 ms per step with 256x256 tiles only (mode 2, always 192-wide: 17.52).
#ifdef XL_EXPERIMENTAL
    if (cx.gemm_bn192 < 0) cx.gemm_bn192 = env_int("XL_GEMM_BN192", 0);     // 0 never, 1 by cost, 2 whenever eligible
#else
    cx.gemm_bn192 = 0;          // (256x192 tiles, split-K with an epilogue, q tiles, relay, persistent, pairs: experimental build, XL_EXPERIMENTAL=1)
#endif
    const int bn192_mode = cx.gemm_bn192;
    int bn = 256;
    if (use_pp && bn192_mode && a_kmajor && M % 256 == 0 && N % 192 == 0 && out_dtype == in_dtype && !accumulate &&
        colsum_out == nullptr && epilogue != XL_EPI_TANH && epilogue != XL_EPI_ROWMAX) {
        const long t256 = (long)p.tiles_m * ((N + 255) / 256), t192 = (long)p.tiles_m * (N / 192);
        const double c256 = (double)((t256 + 255) / 256), c192 = 0.8 * (double)((t192 + 255) / 256);
        if (bn192_mode == 2 || c192 < c256) bn = 192;
    }
    p.tiles_n = bn == 192 ? N / 192 : (N + tile - 1) / tile;
    int tiles = p.tiles_m * p.tiles_n;
    // split-K only for the weight-gradient shape (fp32 out, plain epilogue): few output tiles, deep K
    int splitk = 1;
    const int want = use_pp ? 256 : 768;
    if (may_split && tiles < want && K >= 1024) {
        splitk = use_pp ? want / tiles : (want + tiles - 1) / tiles;
        const int max_split = K / 512;
        if (splitk > max_split) splitk = max_split;
        if (splitk < 1) splitk = 1;
    }
    const int kstep = mfma_ok ? 64 : 16;
    int kper = (K + splitk - 1) / splitk;
    kper = ((kper + kstep - 1) / kstep) * kstep;
    splitk = (K + kper - 1) / kper;
    p.splitk = splitk; p.kper = kper;
    p.atomic_out = (accumulate || splitk > 1) ? 1 : 0;
    // split-K of the ping-pong kernel meets in slabs when the stream has a workspace (xl_gemm_set_workspace): one
    // read-modify-write pass over C by the last arriver of every tile instead of a pass of fp32 atomics per split
    if (cx.wgrad_slabs < 0) cx.wgrad_slabs = env_int("XL_GEMM_WGRAD_SLABS", 0);
    if (use_pp && splitk > 1 && cx.wgrad_slabs && slab_workspace(st, tiles, (long)tiles * splitk, &p.slab, &p.tickets))
        p.slab_det = 1;                   // slices summed in slice order whoever arrives last: the same bits every run
    if (splitk > 1 && !accumulate) {
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st);
        XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm: memset failed: %s", hipGetErrorString(e));
    }
    // 16-byte epilogue accesses need 8-element (bf16) / 4-element (fp32) aligned rows of C / residual / aux
    p.vec_epi = aligned16(C) && (out_dtype == XL_F32 ? ldc % 4 == 0 : ldc % 8 == 0);
    if (epilogue == XL_EPI_ROWMAX) {
        XL_CHECK_ARG(use_pp, XL_ERR_BAD_SHAPE, "xl_gemm: XL_EPI_ROWMAX needs the ping-pong kernel's operand ranges");
        p.vec_epi = 1;
    }
    if (epilogue == XL_EPI_RESIDUAL) p.vec_epi = p.vec_epi && aligned16(residual) && ldr % 8 == 0;
    if (epilogue == XL_EPI_GELU || epilogue == XL_EPI_DGELU || epilogue == XL_EPI_GELU_DG || epilogue == XL_EPI_MULAUX)
        p.vec_epi = p.vec_epi && aligned16(aux) && ldx % 8 == 0;
    // fast (templated) epilogue: aligned rows, a kind that has one, plain stores
    int epik = -1;
    if (p.vec_epi && !p.atomic_out && epilogue != XL_EPI_TANH && (bias == nullptr || aligned16(bias))) epik = epilogue;
    // the derivative-saving GELU pair is instantiated for the layouts that use it (forward NT / dX NN); generic otherwise
    if ((epilogue == XL_EPI_GELU_DG && !b_kmajor) || (epilogue == XL_EPI_MULAUX && b_kmajor)) epik = -1;
    if (bn == 192 && epik < 0) {          // the 256x192 tile has the fast epilogue only: back to 256x256
        bn = 256;
        p.tiles_n = (N + tile - 1) / tile;
        tiles = p.tiles_m * p.tiles_n;    // (the tail split below and the grid size count THESE tiles; the K split above was sized
    }                                     //  for an fp32-output launch, which never takes the 192-wide tile)
    // K split of a launch WITH an epilogue (forward / dX layouts, bf16 out): a launch of few output tiles and a deep contraction
    // -- the language stream's 3328 packed rows against d x dff / d x 3d weights: 52-56 tiles of 256x192, K = 3072 / 2304 -- runs every
    // tile as `splitk` K slices on whole-CU ping-pong workgroups whose partial accumulators meet in the stream's slab workspace
    // (slab_exchange, summed in slice order: deterministic); the last arriver runs the ordinary fast epilogue.  The same FLOPs on
    // 4x the CUs for a quarter of the time: as 112 half-CU "duo" workgroups, each alone on a CU with one wave per SIMD, these
    // launches ran at 0.14 of the MFMA peak (47 us for 16.9 GFLOP).  XL_GEMM_SPLIT_EPI=0 disables; thresholds below.
#ifdef XL_EXPERIMENTAL
    if (cx.gemm_split_epi < 0) cx.gemm_split_epi = env_int("XL_GEMM_SPLIT_EPI", 0);
#else
    cx.gemm_split_epi = 0;
#endif
    const int split_epi = cx.gemm_split_epi;
    static const int split_epi_min_k = env_int("XL_GEMM_SPLIT_EPI_MIN_K", 1536);
    static const int split_epi_max_tiles = env_int("XL_GEMM_SPLIT_EPI_MAX_TILES", 80);
    bool epi_split = false;
    if (split_epi && pp_ok && pp_mode && a_kmajor && epik >= 0 && epik != XL_EPI_ROWMAX && out_dtype == in_dtype && !accumulate &&
        colsum_out == nullptr && splitk == 1 && !p.atomic_out && K >= split_epi_min_k && K % 64 == 0 && M % 256 == 0 &&
        (double)M * lda < 1e9) {
        const int e_bn = (N % 192 == 0 && (N % 256 != 0 || (M / 256) * (N / 192) <= 256)) ? 192 : (N % 256 == 0 ? 256 : 0);
        if (e_bn) {
            const int e_tiles = (M / 256) * (N / e_bn);
            static const int split_epi_max_s = env_int("XL_GEMM_SPLIT_EPI_MAX_S", 4);
            int S = std::min(std::min(256 / std::max(e_tiles, 1), K / 768), split_epi_max_s);     // >= 12 K tiles per slice
            if (e_tiles <= split_epi_max_tiles && S >= 2) {
                int kper_e = ((K + S - 1) / S + 63) / 64 * 64;
                S = (K + kper_e - 1) / kper_e;
                if (S >= 2 && slab_workspace(st, e_tiles, (long)e_tiles * S, &p.slab, &p.tickets)) {
                    epi_split = true;
                    bn = e_bn;
                    p.tiles_m = M / 256; p.tiles_n = N / e_bn;
                    tiles = e_tiles;
                    p.splitk = S; p.kper = kper_e; p.slab_det = 1;
                }
            }
        }
    }
#ifdef XL_EXPERIMENTAL
    // "relay" kernel (gemm_relay.hip): one persistent workgroup per CU whose two wave groups trade roles every 256 x 128 tile -- one runs
    // the K loop as a self-pipelined MFMA stream, the other issues its LDS-DMA and runs the previous tile's epilogue under it.  Mode 0
    // never, 1 launches of more than XL_GEMM_RELAY_MIN_TILES 256x256 tiles with K <= XL_GEMM_RELAY_MAX_K, 2 every eligible launch.
    if (cx.gemm_relay < 0) cx.gemm_relay = env_int("XL_GEMM_RELAY", 0);
    static const int relay_min_tiles = env_int("XL_GEMM_RELAY_MIN_TILES", 257);
    static const int relay_max_k = env_int("XL_GEMM_RELAY_MAX_K", 1024);
    if (cx.gemm_relay_wgs < 0) cx.gemm_relay_wgs = env_int("XL_GEMM_RELAY_WGS", 256);
    const int relay_wgs = cx.gemm_relay_wgs;
    if (!epi_split && pp_ok && pp_mode && cx.gemm_relay && a_kmajor && M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && K >= 768 &&
        out_dtype == in_dtype && !accumulate && epik >= 0 && colsum_out == nullptr && splitk == 1 && !p.atomic_out &&
        relay_has_instance(b_kmajor, epik) &&
        (cx.gemm_relay == 2 || (t256n >= relay_min_tiles && K <= relay_max_k))) {
        p.tiles_m = M / 256; p.tiles_n = N / 256;
        p.splitk = 1; p.kper = K; p.tail_tiles = 0;
        const int nt = p.tiles_m * p.tiles_n * 2;
        const int wgs = std::max(1, std::min(relay_wgs, nt / 2));
        hipError_t e = launch_relay(p, b_kmajor, epik, wgs, st);
        XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        XL_CHECK_LAUNCH();
        return XL_OK;
    }
    // 128x192 tiles by eight waves of 32 x 96 at 128 registers, two workgroups per CU (gemm_q.hip): every workgroup keeps two waves per
    // SIMD in its K loop, so one's prologue / epilogue / hand-over runs under the other's K loop.  Mode 0 never, 1 contractions of at
    // most XL_GEMM_Q_MAX_K (the K = 768 launches, a third of whose tile time the whole-CU tile spends outside the K loop), 2 every
    // eligible launch.  Eligible: forward / dX layouts, bf16 in / out, fast epilogue kind with an instance, no fused column sums,
    // M % 128 == N % 192 == K % 64 == 0, enough tiles to fill the chip twice over.
    if (cx.gemm_q < 0) cx.gemm_q = env_int("XL_GEMM_Q", 0);
    static const int q_max_k = env_int("XL_GEMM_Q_MAX_K", 1024);
    static const int q_min_tiles = env_int("XL_GEMM_Q_MIN_TILES", 64);
    static const int q_max_n = env_int("XL_GEMM_Q_MAX_N", 2304);
    static const int q_max_tiles = env_int("XL_GEMM_Q_MAX_TILES", 1 << 30);
    if (!epi_split && pp_ok && pp_mode && cx.gemm_q && a_kmajor && M % 128 == 0 && N % 192 == 0 && K % 64 == 0 && out_dtype == in_dtype &&
        !accumulate && epik >= 0 && colsum_out == nullptr && splitk == 1 && !p.atomic_out && q_has_instance(b_kmajor, epik) &&
        (double)M * lda < 1e9 && (cx.gemm_q == 2 || (K <= q_max_k && N <= q_max_n && (long)(M / 128) * (N / 192) >= q_min_tiles && (long)(M / 128) * (N / 192) <= q_max_tiles))) {
        p.tiles_m = M / 128; p.tiles_n = N / 192;
        p.splitk = 1; p.kper = K; p.tail_tiles = 0;
        hipError_t e = launch_q(p, b_kmajor, epik, p.tiles_m * p.tiles_n, st);
        XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        XL_CHECK_LAUNCH();
        return XL_OK;
    }
#endif
    // 128x192 "duo" tiles, two four-wave workgroups per CU (gemm_pp_kernel.h PPGeo<192, 128>): same eligibility as the 256x192 tile
    // (forward / dX layouts, N a multiple of 192, fast epilogue, plain stores) with M a multiple of 128
    // mode 1: the launches of fewer than XL_GEMM_DUO_MAX_TILES 256x256 tiles (the language stream's: 3328 packed rows = 39 tiles,
    // which otherwise go to the 128x128 kernel at 0.10 MFMA-busy); mode 2: every eligible launch
    if (cx.gemm_duo < 0) cx.gemm_duo = env_int("XL_GEMM_DUO", 1);
    static const int duo_max_tiles = env_int("XL_GEMM_DUO_MAX_TILES", 64);
    int bm = 256;
    if (!epi_split && pp_ok && pp_mode && cx.gemm_duo && a_kmajor && M % 128 == 0 && N % 192 == 0 && out_dtype == in_dtype && !accumulate && epik >= 0 &&
        colsum_out == nullptr && epilogue != XL_EPI_TANH && epilogue != XL_EPI_ROWMAX && splitk == 1 && !p.atomic_out &&
        (double)M * lda < 1e9 && (cx.gemm_duo == 2 || t256n <= duo_max_tiles)) {
        bm = 128; bn = 192;
        p.tiles_m = M / 128; p.tiles_n = N / 192;
        tiles = p.tiles_m * p.tiles_n;
    }
    // column sums of C ride in the fast epilogue when every tile takes it; otherwise a separate pass over C follows
    const bool colsum_fused = colsum_out != nullptr && mfma_ok && epik >= 0 && splitk == 1 && a_kmajor && M % tile == 0 &&
                              N % tile == 0;
    if (colsum_fused) p.colsum_ws = colsum_ws;
    int nblk = p.tiles_m * p.tiles_n * (epi_split ? p.splitk : splitk);
    // tail split (gemm_pp.hip): more than one round of tiles, a nearly empty last round, a deep contraction, and a slab
    // workspace on this stream (xl_gemm_set_workspace)
    // (measured at the masked-row head, 8448 rows: d(feat) = d(logits) C, K = 10000, 264 tiles: 466 -> 332 us; the logits
    // contraction, K = 2048, 1320 tiles, does not gain -- 338 -> 368 us -- hence the depth threshold)
    if (cx.tail_max < 0) { cx.tail_max = env_int("XL_GEMM_TAIL_MAX", 64); cx.tail_min_k = env_int("XL_GEMM_TAIL_MIN_K", 4096); }
    const int tail_max = cx.tail_max, tail_min_k = cx.tail_min_k;
    if (use_pp && !epi_split && bm == 256 && splitk == 1 && !p.atomic_out && tiles > 256 && tiles % 256 <= tail_max && tiles % 256 > 0 && K >= tail_min_k) {
        const int rem = tiles % 256;
        int S = std::min(std::min(256 / rem, K / 512), 8);          // >= 8 K tiles per slice, <= 7 slabs for the last arriver to add
        if (S >= 2) {
            int kper_t = (K + S - 1) / S;
            kper_t = (kper_t + 63) / 64 * 64;
            S = (K + kper_t - 1) / kper_t;
            if (S >= 2 && slab_workspace(st, rem, (long)rem * S, &p.slab, &p.tickets)) {
                p.tail_tiles = rem; p.tail_kper = kper_t; p.splitk = S;
                nblk = tiles - rem + rem * S;
            }
        }
    }
    static const int n_cu = [] { int dev = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; return n; }();
#ifdef XL_EXPERIMENTAL
    // several rounds of 256x256 tiles with a short contraction: the persistent variant (next tile's first K tile requested under
    // the epilogue, no workgroup hand-over between tiles) -- OPT-IN: faster alone, slower inside the four-stream step
    if (cx.gemm_persist < 0) cx.gemm_persist = env_int("XL_GEMM_PERSIST", 0);      // opt-in: see gemm_pp_persist.hip
    static const int persist_max_k = env_int("XL_GEMM_PERSIST_MAX_K", 1536);
    if (use_pp && cx.gemm_persist && bn == 256 && bm == 256 && a_kmajor && epik >= 0 && epik != XL_EPI_TANH && epik != XL_EPI_ROWMAX &&
        epik != XL_EPI_RESIDUAL && out_dtype == XL_BF16 && !p.atomic_out && p.splitk == 1 && p.tail_tiles == 0 && M % 256 == 0 &&
        N % 256 == 0 && K % 64 == 0 && K >= 128 && K <= persist_max_k && tiles > n_cu && (colsum_out == nullptr || colsum_fused) &&
        (double)M * lda < 1e9 && cx.gemm_trace == nullptr) {
        hipError_t e = launch_pp_persist(p, b_kmajor, epik, std::min(tiles, n_cu), st);
        if (e == hipSuccess) {
            XL_CHECK_LAUNCH();
            if (colsum_fused) { launch_colsum_reduce(colsum_ws, M / 128, N, colsum_out, st); XL_CHECK_LAUNCH(); }
            return XL_OK;
        }
        XL_CHECK_ARG(e == hipErrorInvalidValue, XL_ERR_HIP, "xl_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
#endif
    if (use_pp || bm == 128 || epi_split) {
        hipError_t e = launch_pp(p, a_kmajor, b_kmajor, epik, bn, nblk, st, bm);
        XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    } else if (mfma_ok) {
        if (a_kmajor && b_kmajor) launch_mfma<true, true>(p, epik, nblk, st);
        else if (a_kmajor && !b_kmajor) launch_mfma<true, false>(p, epik, nblk, st);
        else if (!a_kmajor && b_kmajor) launch_mfma<false, true>(p, epik, nblk, st);
        else launch_mfma<false, false>(p, epik, nblk, st);
    } else if (in_dtype == XL_BF16) {
        hipLaunchKernelGGL((gemm_generic_kernel<bf16_t>), dim3(nblk), dim3(256), 0, st, p, a_kmajor, b_kmajor);
    } else {
        hipLaunchKernelGGL((gemm_generic_kernel<float>), dim3(nblk), dim3(256), 0, st, p, a_kmajor, b_kmajor);
    }
    XL_CHECK_LAUNCH();
    if (colsum_fused) {
        launch_colsum_reduce(colsum_ws, M / (use_pp ? 128 : 64), N, colsum_out, st);
        XL_CHECK_LAUNCH();
    } else if (colsum_out != nullptr) {
        return xl_colsum(C, colsum_out, M, N, ldc, colsum_ws, out_dtype, stream);
    }
    return XL_OK;
}

#ifdef XL_EXPERIMENTAL
// Two contractions of the same shape class in one launch (gemm_pp_kernel.h PairParams); see include/xlxmert_hip.h.  Whatever the
// launch strategy, the results are those of xl_gemm(problem 0) followed by xl_gemm(problem 1): the same tile kernel runs both.
extern "C" int xl_gemm_pair(const void* A0, const void* B0, void* C0, const float* bias0, const void* residual0, void* aux0, int M0,
                            uint64_t seed0, float* colsum_out0, float* colsum_ws0,
                            const void* A1, const void* B1, void* C1, const float* bias1, const void* residual1, void* aux1, int M1,
                            uint64_t seed1, float* colsum_out1, float* colsum_ws1,
                            int N, int K, int lda, int ldb, int ldc, int ldr, int ldx, int a_kmajor, int b_kmajor, int in_dtype,
                            int out_dtype, int epilogue, float alpha, float p_drop, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Ctx& cx = ctx();
    const void* A[2] = {A0, A1}; const void* B[2] = {B0, B1}; void* C[2] = {C0, C1};
    const float* bias[2] = {bias0, bias1}; const void* res[2] = {residual0, residual1}; void* aux[2] = {aux0, aux1};
    const int M[2] = {M0, M1}; const uint64_t seed[2] = {seed0, seed1};
    float* cs_out[2] = {colsum_out0, colsum_out1}; float* cs_ws[2] = {colsum_ws0, colsum_ws1};
    if (cx.gemm_pp < 0) cx.gemm_pp = env_int("XL_GEMM_PP", 1);
    if (cx.gemm_pair < 0) cx.gemm_pair = env_int("XL_GEMM_PAIR", 1);
    static const int pp_min_tiles = env_int("XL_GEMM_PP_MIN_TILES", 48);
    bool one = cx.gemm_pair != 0 && cx.gemm_pp != 0 && cx.use_tr_read && in_dtype == XL_BF16 && out_dtype == XL_BF16 && a_kmajor &&
               M0 > 0 && M1 > 0 && N > 0 && K > 0 && N % 256 == 0 && K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
               lda >= K && ldb >= (b_kmajor ? K : N) && ldc >= N && p_drop >= 0.f && p_drop < 1.f &&
               epilogue != XL_EPI_TANH && epilogue != XL_EPI_ROWMAX && pp_pair_has_instance(b_kmajor, epilogue) &&
               (double)(b_kmajor ? N : K) * ldb < 1e9;
    long tiles[2] = {0, 0};
    for (int i = 0; i < 2 && one; ++i) {
        one = A[i] && B[i] && C[i] && M[i] % 256 == 0 && aligned16(A[i]) && aligned16(B[i]) && aligned16(C[i]) && (double)M[i] * lda < 1e9 &&
              (bias[i] == nullptr || aligned16(bias[i]));
        if (epilogue == XL_EPI_RESIDUAL) one = one && res[i] && aligned16(res[i]) && ldr % 8 == 0 && ldr >= N;
        if (epilogue == XL_EPI_GELU_DG || epilogue == XL_EPI_MULAUX) one = one && aux[i] && aligned16(aux[i]) && ldx % 8 == 0 && ldx >= N;
        if (cs_out[i] != nullptr) one = one && cs_ws[i] != nullptr && (long)((M[i] + 63) / 64) * N <= xl_workspace_floats(N);
        tiles[i] = (long)(M[i] / 256) * (N / 256);
    }
    one = one && (cx.gemm_pp == 2 || tiles[0] + tiles[1] >= pp_min_tiles) && tiles[0] < (1 << 20) && tiles[1] < (1 << 20);
    if (!one) {          // fp32 parity path, small or ragged problems, kinds without a paired instance: one launch each
        for (int i = 0; i < 2; ++i) {
            const int rc = xl_gemm(A[i], B[i], C[i], bias[i], res[i], aux[i], M[i], N, K, lda, ldb, ldc, ldr, ldx, a_kmajor, b_kmajor,
                                   in_dtype, out_dtype, epilogue, alpha, 0, p_drop, seed[i], cs_out[i], cs_ws[i], stream);
            if (rc != XL_OK) return rc;
        }
        return XL_OK;
    }
    static const int ablate = env_int("XL_GEMM_ABLATE", 0);
    PairParams pp;
    for (int i = 0; i < 2; ++i) {
        GemmParams& p = pp.p[i];
        p.A = A[i]; p.B = B[i]; p.C = C[i]; p.bias = bias[i]; p.residual = res[i]; p.aux = aux[i];
        p.M = M[i]; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldx = ldx;
        p.epilogue = epilogue; p.out_f32 = 0; p.atomic_out = 0; p.splitk = 1; p.kper = (K + 63) / 64 * 64; p.vec_epi = 1;
        p.alpha = alpha; p.p_drop = p_drop; p.inv_keep = 1.0f / (1.0f - p_drop); p.seed = seed[i]; p.step_seed = cx.step_seed;
        p.tiles_m = M[i] / 256; p.tiles_n = N / 256; p.ablate = ablate; p.trace = cx.gemm_trace;
        p.colsum_ws = cs_out[i] != nullptr ? cs_ws[i] : nullptr;
        p.slab = nullptr; p.tickets = nullptr; p.tail_tiles = 0; p.tail_kper = 0; p.overwrite = 0; p.slab_det = 0;
    }
    pp.tiles0 = (int)tiles[0];
    hipError_t e = launch_pp_pair(pp, b_kmajor, epilogue, (int)(tiles[0] + tiles[1]), st);
    XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm_pair: launch failed: %s", hipGetErrorString(e));
    XL_CHECK_LAUNCH();
    for (int i = 0; i < 2; ++i)
        if (cs_out[i] != nullptr) { launch_colsum_reduce(cs_ws[i], M[i] / 128, N, cs_out[i], st); XL_CHECK_LAUNCH(); }
    return XL_OK;
}
#endif

#ifdef XL_EXPERIMENTAL
extern "C" int xl_set_gemm_relay(int mode) {
    XL_CHECK_ARG(mode >= 0 && mode <= 2, XL_ERR_BAD_ARG, "xl_set_gemm_relay: mode %d", mode);
    ctx().gemm_relay = mode;
    return XL_OK;
}
#endif

#ifdef XL_EXPERIMENTAL
extern "C" int xl_set_gemm_relay_wgs(int wgs) {
    XL_CHECK_ARG(wgs >= 1 && wgs <= 4096, XL_ERR_BAD_ARG, "xl_set_gemm_relay_wgs: %d", wgs);
    ctx().gemm_relay_wgs = wgs;
    return XL_OK;
}
#endif

#ifdef XL_EXPERIMENTAL
extern "C" int xl_set_gemm_q(int mode) {
    XL_CHECK_ARG(mode >= 0 && mode <= 2, XL_ERR_BAD_ARG, "xl_set_gemm_q: mode %d", mode);
    ctx().gemm_q = mode;
    return XL_OK;
}
#endif

#ifdef XL_EXPERIMENTAL
extern "C" int xl_set_gemm_pair(int on) {
    ctx().gemm_pair = on ? 1 : 0;
    return XL_OK;
}
#endif

extern "C" int xl_gemm_wgrad_group_splitk(const int* M, const int* N, const int* K, int count) {
    if (M == nullptr || N == nullptr || K == nullptr || count < 1 || count > 8) return 0;
    static const int group_max_wgs = env_int("XL_GEMM_GROUP_MAX_WGS", 256);
    long total = 0;
    int max_split = 1 << 20;
    for (int i = 0; i < count; ++i) {
        total += (long)((M[i] + 255) / 256) * ((N[i] + 255) / 256);
        max_split = std::min(max_split, std::max(1, K[i] / 512));
    }
    const int splitk = total >= group_max_wgs ? 1 : (int)std::min<long>(group_max_wgs / total, max_split);
    return std::max(splitk, 1);
}

extern "C" int xl_gemm_wgrad_group(const void* const* A, const void* const* B, void* const* C,
                                   const int* M, const int* N, const int* K, const int* lda, const int* ldb, const int* ldc,
                                   int count, int overwrite_mask, int dtype, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    XL_CHECK_ARG(count >= 1 && count <= 8, XL_ERR_BAD_ARG, "xl_gemm_wgrad_group: count %d (1..8)", count);
    XL_CHECK_ARG(A && B && C && M && N && K && lda && ldb && ldc, XL_ERR_BAD_ARG, "xl_gemm_wgrad_group: null argument");
    XL_CHECK_ARG(dtype == XL_F32 || dtype == XL_BF16, XL_ERR_BAD_DTYPE, "xl_gemm_wgrad_group: bad dtype %d", dtype);
    Ctx& cx = ctx();
    if (cx.gemm_pp < 0) cx.gemm_pp = env_int("XL_GEMM_PP", 1);
    bool grouped = dtype == XL_BF16 && cx.gemm_pp != 0 && cx.use_tr_read && count > 1;
    long total = 0;
    int max_split = 1 << 20;
    for (int i = 0; i < count; ++i) {
        XL_CHECK_ARG(A[i] && B[i] && C[i] && M[i] > 0 && N[i] > 0 && K[i] > 0 && lda[i] >= M[i] && ldb[i] >= N[i] && ldc[i] >= N[i],
                     XL_ERR_BAD_SHAPE, "xl_gemm_wgrad_group: problem %d: bad operands / shape", i);
        grouped = grouped && lda[i] % 8 == 0 && ldb[i] % 8 == 0 && aligned16(A[i]) && aligned16(B[i]) && K[i] % 8 == 0 &&
                  (double)K[i] * lda[i] < 1e9 && (double)K[i] * ldb[i] < 1e9;
        total += (long)((M[i] + 255) / 256) * ((N[i] + 255) / 256);
        max_split = std::min(max_split, std::max(1, K[i] / 512));
    }
    static const int group_min_blocks = env_int("XL_GEMM_GROUP_MIN_BLOCKS", 96);
    static const int group_max_wgs = env_int("XL_GEMM_GROUP_MAX_WGS", 256);       // workgroups a K-split group launch may put up
    int splitk = total >= group_max_wgs ? 1 : (int)std::min<long>(group_max_wgs / total, max_split);
    if (splitk < 1) splitk = 1;
    if (grouped && total * splitk < group_min_blocks) grouped = false;
    if (!grouped) {          // one launch per problem (fp32 parity path, operands the ping-pong kernel does not take, tiny groups)
        for (int i = 0; i < count; ++i) {
            const int acc_i = (overwrite_mask >> i) & 1 ? 0 : 1;          // overwrite: xl_gemm stores (or clears + atomics when it splits K)
            int rc = xl_gemm(A[i], B[i], C[i], nullptr, nullptr, nullptr, M[i], N[i], K[i], lda[i], ldb[i], ldc[i], 0, 0,
                             0, 0, dtype, XL_F32, XL_EPI_NONE, 1.0f, acc_i, 0.f, 0, nullptr, nullptr, stream);
            if (rc != XL_OK) return rc;
        }
        return XL_OK;
    }
    GroupParams g;
    g.count = count; g.splitk = splitk;
    g.slab = nullptr; g.tickets = nullptr;
    // slabs + vector read-modify-write of C need one writer per output element: no two problems may share any of C
    bool disjoint = true;
    for (int i = 0; i < count && disjoint; ++i)
        for (int j = i + 1; j < count; ++j) {
            const char* a0 = reinterpret_cast<const char*>(C[i]);
            const char* b0 = reinterpret_cast<const char*>(C[j]);
            const char* a1 = a0 + ((size_t)(M[i] - 1) * ldc[i] + N[i]) * sizeof(float);
            const char* b1 = b0 + ((size_t)(M[j] - 1) * ldc[j] + N[j]) * sizeof(float);
            if (a0 < b1 && b0 < a1) { disjoint = false; break; }
        }
    if (cx.wgrad_slabs < 0) cx.wgrad_slabs = env_int("XL_GEMM_WGRAD_SLABS", 0);
    if (disjoint && cx.wgrad_slabs) slab_workspace(st, total, total * splitk, &g.slab, &g.tickets);
    g.rmw = (disjoint && splitk == 1) ? 1 : 0;      // one workgroup per output tile: it owns the tile (two layers' weight gradients in
                                                    // one launch: 216 tiles, no K split, no atomics)
    int acc = 0;
    for (int i = 0; i < count; ++i) {
        GroupProblem& pr = g.prob[i];
        pr.A = A[i]; pr.B = B[i]; pr.C = C[i]; pr.M = M[i]; pr.N = N[i]; pr.K = K[i];
        pr.lda = lda[i]; pr.ldb = ldb[i]; pr.ldc = ldc[i];
        pr.tiles_m = (M[i] + 255) / 256; pr.tiles_n = (N[i] + 255) / 256;
        pr.vec = aligned16(C[i]) && ldc[i] % 4 == 0;
        // overwrite: plain stores when every tile of the problem has ONE writer that takes the vector epilogue (no K split or the
        // slab path, interior tiles, aligned rows); otherwise C is cleared here and the launch accumulates as usual
        pr.overwrite = 0;
        if ((overwrite_mask >> i) & 1) {
            const bool one_writer = (g.rmw || g.slab != nullptr) && pr.vec && M[i] % 256 == 0 && N[i] % 256 == 0;
            if (one_writer) pr.overwrite = 1;
            else {
                hipError_t e = hipMemset2DAsync(C[i], (size_t)ldc[i] * sizeof(float), 0, (size_t)N[i] * sizeof(float), M[i], st);
                XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm_wgrad_group: memset failed: %s", hipGetErrorString(e));
            }
        }
        int kper = (K[i] + splitk - 1) / splitk;
        pr.kper = (kper + 63) / 64 * 64;
        g.tile_start[i] = acc;
        acc += pr.tiles_m * pr.tiles_n;
    }
    for (int i = count; i <= 8; ++i) g.tile_start[i] = acc;
    // Tile walk.  The launch's linear tile order is cut into 8 contiguous chunks, one per XCD (its own 4 MiB L2); all tiles of a chunk
    // run side by side through K, so an operand column panel (K x 256) that several of them need is fetched once per chunk.  In
    // [problem][tn][tm] order a chunk of 27 tiles cuts across problems and rows (two visual layers, 216 tiles: 64 panel fetches per
    // layer for 48 distinct panels = the 1.33x over-fetch measured by tools/gemm_overfetch.py).  Here every problem's grid is cut into
    // rectangles of at most one chunk that span its SHORT side completely, and the rectangles are dealt largest first: 54 fetches.
    g.order_n = 0;
    static const int group_order = env_int("XL_GEMM_GROUP_ORDER", 1);
    if (group_order && acc <= kGroupOrderMax && count <= 8) {
        struct Blk { int prob, m0, m1, n0, n1; };
        std::vector<Blk> blks;
        const int chunk = std::max(1, (acc * splitk + 7) / 8);
        bool ok = true;
        for (int i = 0; i < count; ++i) {
            const int tm = g.prob[i].tiles_m, tn = g.prob[i].tiles_n;
            if (tm > 63 || tn > 63) { ok = false; break; }
            if (tm >= tn) {
                const int bm = std::max(1, chunk / tn);
                for (int m0 = 0; m0 < tm; m0 += bm) blks.push_back({i, m0, std::min(tm, m0 + bm), 0, tn});
            } else {
                const int bn = std::max(1, chunk / tm);
                for (int n0 = 0; n0 < tn; n0 += bn) blks.push_back({i, 0, tm, n0, std::min(tn, n0 + bn)});
            }
        }
        if (ok) {
            std::stable_sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) {
                return (a.m1 - a.m0) * (a.n1 - a.n0) > (b.m1 - b.m0) * (b.n1 - b.n0); });
            int n = 0;
            for (const Blk& b : blks)
                for (int tn = b.n0; tn < b.n1; ++tn)
                    for (int tm = b.m0; tm < b.m1; ++tm) g.order[n++] = (uint16_t)(b.prob << 12 | tm << 6 | tn);
            g.order_n = n;          // == acc
        }
    }
    hipError_t e = launch_pp_group(g, acc * splitk, st);
    XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm_wgrad_group: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    XL_CHECK_LAUNCH();
    return XL_OK;
}
