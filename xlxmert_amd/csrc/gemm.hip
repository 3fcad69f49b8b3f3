// Dense contractions of the X-LXMERT path (nn.Linear forward, dX, dW) for gfx950.
//
//   C[M,N] = alpha * sum_k A(m,k) B(n,k) (+bias) -> epilogue     (see include/xlxmert_hip.h: xl_gemm)
//
// Two kernels:
//   gemm_bf16_mfma_kernel  bf16 operands, fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//       128x128x64 block tile, 4 waves (2x2), each wave a 64x64 tile = 2x2 MFMA fragments.
//       Operands are staged global -> VGPR -> LDS (double buffered, one barrier per K tile).
//       K-major operands ([rows][k]) are kept row-major in LDS with a 16-byte-chunk XOR swizzle
//       (chunk ^= (row>>1)&7) so that ds_read_b128 fragment reads are conflict free.
//       M-major operands ([k][rows]; the dX / dW contractions) are kept as stored and turned into
//       MFMA fragments by ds_read_b64_tr_b16 (LDS transpose read); 64-byte XOR swizzle on k&3.
//       The MFMA k-slot <-> k mapping is applied identically to A and B, which is all the
//       contraction needs.
//   gemm_generic_kernel    any dtype / any alignment, fp32 FMA, 64x64x16 tile.  It is the exact-fp32
//       path (XL_F32: parity configuration) and the fallback for operands the MFMA loader cannot
//       take (leading dimension not a multiple of 8 elements).
#include "common.h"

namespace xl {

struct GemmParams {
    const void* A; const void* B; void* C;
    const float* bias; const void* residual; void* aux;
    int M, N, K, lda, ldb, ldc, ldr, ldx;
    int epilogue, out_f32, atomic_out, splitk, kper;
    float alpha, p_drop, inv_keep;
    uint64_t seed;
    int tiles_m, tiles_n;
};

// ------------------------------------------------------------------ epilogue (shared by both kernels)
template <typename TIn>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int m, int n, float v, bool add_bias) {
    v *= p.alpha;
    if (p.bias != nullptr && add_bias) v += p.bias[n];
    switch (p.epilogue) {
        case XL_EPI_GELU: {
            TIn* aux = reinterpret_cast<TIn*>(p.aux);
            Elem<TIn>::st(aux + (size_t)m * p.ldx + n, v);
            v = gelu_erf(v);
            break;
        }
        case XL_EPI_RESIDUAL: {
            if (p.p_drop > 0.0f) v *= dropout_scale(p.seed, (uint64_t)m * (uint64_t)p.N + n, p.p_drop, p.inv_keep);
            const TIn* res = reinterpret_cast<const TIn*>(p.residual);
            v += Elem<TIn>::ld(res + (size_t)m * p.ldr + n);
            break;
        }
        case XL_EPI_DGELU: {
            const TIn* aux = reinterpret_cast<const TIn*>(p.aux);
            v *= gelu_erf_grad(Elem<TIn>::ld(aux + (size_t)m * p.ldx + n));
            break;
        }
        case XL_EPI_TANH: v = tanhf(v); break;
        default: break;
    }
    if (p.out_f32) {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
        if (p.atomic_out) atomicAdd(c, v); else *c = v;
    } else {
        Elem<TIn>::st(reinterpret_cast<TIn*>(p.C) + (size_t)m * p.ldc + n, v);
    }
}

// tile id -> (tile_m, tile_n, split) with an XCD-aware remap: block b runs on XCD b%8 (observed
// placement, speed only); give every XCD a contiguous chunk of a grouped (8 m-tiles x all n) order
// so that the 32 tiles co-resident on one XCD share A row panels and B column panels in its L2.
__device__ __forceinline__ void tile_coords(const GemmParams& p, int& tm, int& tn, int& z) {
    const int nblk = gridDim.x;
    const int b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, pos = b >> 3;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    const int tiles = p.tiles_m * p.tiles_n;
    z = L / tiles;
    const int t = L - z * tiles;
    constexpr int GM = 8;
    const int per_group = GM * p.tiles_n;
    const int g = t / per_group;
    const int in_g = t - g * per_group;
    const int gsize = min(GM, p.tiles_m - g * GM);
    tn = in_g / gsize;
    tm = g * GM + (in_g - tn * gsize);
}

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

// ================================================================== bf16 MFMA kernel
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;     // 16 KiB per operand tile

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf16_t;

__device__ __forceinline__ bf16x4_t lds_tr_read(const uint8_t* ptr) {
    auto p = (__attribute__((address_space(3))) v4bf16_t*)(ptr);
    v4bf16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
    return __builtin_bit_cast(bf16x4_t, r);
}

// global -> registers: 4 x 16 bytes per thread for one 128(rows) x 64(k) operand tile
template <bool KMAJ>
__device__ __forceinline__ void gload_tile(const bf16_t* __restrict__ P, int ld, int row0, int rows_ext,
                                           int k0, int kend, uint4 (&r)[4], int tid) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        int gr, gc, lim;              // gr: index along the strided dim, gc: start along the contiguous dim
        bool row_ok;
        if (KMAJ) { gr = row0 + ps * 32 + (tid >> 3); gc = k0 + (tid & 7) * 8; row_ok = gr < rows_ext; lim = kend; }
        else      { gr = k0 + ps * 16 + (tid >> 4);   gc = row0 + (tid & 15) * 8; row_ok = gr < kend;  lim = rows_ext; }
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row_ok) {
            const bf16_t* src = P + (size_t)gr * ld + gc;
            if (gc + 8 <= lim) {
                v = *reinterpret_cast<const uint4*>(src);
            } else if (gc < lim) {            // ragged tail of the contiguous dim
                bf16_t e[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = (gc + i < lim) ? src[i] : (bf16_t)0;
                v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
                v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
            }
        }
        r[ps] = v;
    }
}

template <bool KMAJ>
__device__ __forceinline__ void lds_store_tile(uint8_t* tile, const uint4 (&r)[4], int tid) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        int off;
        if (KMAJ) { const int rr = ps * 32 + (tid >> 3), c = tid & 7; off = rr * 128 + ((c ^ ((rr >> 1) & 7)) << 4); }
        else      { const int kr = ps * 16 + (tid >> 4), c = tid & 15; off = kr * 256 + ((c << 4) ^ ((kr & 3) << 6)); }
        *reinterpret_cast<uint4*>(tile + off) = r[ps];
    }
}

// MFMA operand fragment: rows [r0, r0+32) of the tile (lane -> row l&31), k-slots s*16 + (l>>5)*8 + 0..7
template <bool KMAJ, bool TR>
__device__ __forceinline__ bf16x8_t lds_load_frag(const uint8_t* tile, int r0, int s, int lane) {
    if (KMAJ) {
        const int row = r0 + (lane & 31);
        const int c = s * 2 + (lane >> 5);
        return *reinterpret_cast<const bf16x8_t*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
    } else if (TR) {
        // 16-lane group g reads a [4 k][16 rows] block; lane t supplies the address of k-row t>>2,
        // row-chunk (t&3)*4 and receives column t (4 consecutive k).
        const int t = lane & 15;
        const int mcol = r0 + ((lane >> 4) & 1) * 16 + (t & 3) * 4;
        const int kb = s * 16 + (lane >> 5) * 8 + (t >> 2);
        const int k0r = kb, k1r = kb + 4;
        bf16x4_t lo = lds_tr_read(tile + k0r * 256 + ((mcol * 2) ^ ((k0r & 3) << 6)));
        bf16x4_t hi = lds_tr_read(tile + k1r * 256 + ((mcol * 2) ^ ((k1r & 3) << 6)));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
        const int m = r0 + (lane & 31);
        bf16x8_t f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = s * 16 + (lane >> 5) * 8 + j;
            f[j] = *reinterpret_cast<const short*>(tile + k * 256 + ((m * 2) ^ ((k & 3) << 6)));
        }
        return f;
    }
}

template <bool AK, bool BKM, bool TR>
__global__ __launch_bounds__(256) void gemm_bf16_mfma_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [2 buffers][A tile | B tile]
    int tm, tn, z;
    tile_coords(p, tm, tn, z);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * p.kper, kend = min(p.K, kbeg + p.kper);
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[4], rb[4];
    const int nkt = (kend - kbeg + BK - 1) / BK;
    if (nkt > 0) {
        gload_tile<AK>(A, p.lda, m0, p.M, kbeg, kend, ra, tid);
        gload_tile<BKM>(B, p.ldb, n0, p.N, kbeg, kend, rb, tid);
        lds_store_tile<AK>(smem, ra, tid);
        lds_store_tile<BKM>(smem + TILE_BYTES, rb, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const uint8_t* ta = smem + (kt & 1) * 2 * TILE_BYTES;
        const uint8_t* tb = ta + TILE_BYTES;
        const bool more = (kt + 1) < nkt;
        if (more) {                                   // next tile's global loads fly under the MFMAs
            gload_tile<AK>(A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, ra, tid);
            gload_tile<BKM>(B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, rb, tid);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t fa[2], fb[2];
            fa[0] = lds_load_frag<AK, TR>(ta, wm, s, lane);
            fa[1] = lds_load_frag<AK, TR>(ta, wm + 32, s, lane);
            fb[0] = lds_load_frag<BKM, TR>(tb, wn, s, lane);
            fb[1] = lds_load_frag<BKM, TR>(tb, wn + 32, s, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(v8bf16_t, fa[i]), __builtin_bit_cast(v8bf16_t, fb[j]), acc[i][j], 0, 0, 0);
        }
        if (more) {
            uint8_t* na = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
            lds_store_tile<AK>(na, ra, tid);
            lds_store_tile<BKM>(na + TILE_BYTES, rb, tid);
        }
        __syncthreads();
    }
    // C/D layout of v_mfma_f32_32x32x16: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) epilogue_store<bf16_t>(p, m, n, acc[i][j][r], z == 0);
            }
        }
}

template <bool AK, bool BKM>
static void launch_mfma(const GemmParams& p, int nblk, hipStream_t st) {
    if (g_use_tr_read)
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<AK, BKM, true>), dim3(nblk), dim3(256), 4 * TILE_BYTES, st, p);
    else
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<AK, BKM, false>), dim3(nblk), dim3(256), 4 * TILE_BYTES, st, p);
}

}  // namespace xl

using namespace xl;

extern "C" int xl_gemm(const void* A, const void* B, void* C, const float* bias,
                       const void* residual, void* aux,
                       int M, int N, int K, int lda, int ldb, int ldc, int ldr, int ldx,
                       int a_kmajor, int b_kmajor, int in_dtype, int out_dtype,
                       int epilogue, float alpha, int accumulate,
                       float p_drop, uint64_t seed, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    XL_CHECK_ARG(M > 0 && N > 0 && K > 0, XL_ERR_BAD_SHAPE, "xl_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    XL_CHECK_ARG(in_dtype == XL_F32 || in_dtype == XL_BF16, XL_ERR_BAD_DTYPE, "xl_gemm: bad in_dtype %d", in_dtype);
    XL_CHECK_ARG(out_dtype == in_dtype || out_dtype == XL_F32, XL_ERR_BAD_DTYPE, "xl_gemm: bad out_dtype %d", out_dtype);
    XL_CHECK_ARG(A && B && C, XL_ERR_BAD_ARG, "xl_gemm: null operand");
    XL_CHECK_ARG(lda >= (a_kmajor ? K : M) && ldb >= (b_kmajor ? K : N) && ldc >= N, XL_ERR_BAD_SHAPE,
                 "xl_gemm: leading dimension too small (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    XL_CHECK_ARG(epilogue >= XL_EPI_NONE && epilogue <= XL_EPI_TANH, XL_ERR_BAD_ARG, "xl_gemm: bad epilogue %d", epilogue);
    if (epilogue == XL_EPI_RESIDUAL) XL_CHECK_ARG(residual && ldr >= N, XL_ERR_BAD_ARG, "xl_gemm: residual missing");
    if (epilogue == XL_EPI_GELU || epilogue == XL_EPI_DGELU) XL_CHECK_ARG(aux && ldx >= N, XL_ERR_BAD_ARG, "xl_gemm: aux missing");
    XL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, XL_ERR_BAD_ARG, "xl_gemm: p_drop %f", p_drop);
    if (accumulate) XL_CHECK_ARG(out_dtype == XL_F32 && epilogue == XL_EPI_NONE, XL_ERR_BAD_ARG,
                                 "xl_gemm: accumulate needs fp32 output and no epilogue");

    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.residual = residual; p.aux = aux;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldx = ldx;
    p.epilogue = epilogue; p.out_f32 = (out_dtype == XL_F32); p.alpha = alpha;
    p.p_drop = p_drop; p.inv_keep = 1.0f / (1.0f - p_drop); p.seed = seed;

    const bool mfma_ok = in_dtype == XL_BF16 && (lda % 8 == 0) && (ldb % 8 == 0) && aligned16(A) && aligned16(B);
    const int tile = mfma_ok ? 128 : 64;
    p.tiles_m = (M + tile - 1) / tile;
    p.tiles_n = (N + tile - 1) / tile;
    const int tiles = p.tiles_m * p.tiles_n;
    // split-K only for the weight-gradient shape (fp32 out, plain epilogue): few output tiles, deep K
    int splitk = 1;
    if (mfma_ok && out_dtype == XL_F32 && epilogue == XL_EPI_NONE && tiles < 512 && K >= 1024) {
        splitk = (768 + tiles - 1) / tiles;
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
    if (splitk > 1 && !accumulate) {
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), M, st);
        XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm: memset failed: %s", hipGetErrorString(e));
    }
    const int nblk = tiles * splitk;
    if (mfma_ok) {
        if (a_kmajor && b_kmajor) launch_mfma<true, true>(p, nblk, st);
        else if (a_kmajor && !b_kmajor) launch_mfma<true, false>(p, nblk, st);
        else if (!a_kmajor && b_kmajor) launch_mfma<false, true>(p, nblk, st);
        else launch_mfma<false, false>(p, nblk, st);
    } else if (in_dtype == XL_BF16) {
        hipLaunchKernelGGL((gemm_generic_kernel<bf16_t>), dim3(nblk), dim3(256), 0, st, p, a_kmajor, b_kmajor);
    } else {
        hipLaunchKernelGGL((gemm_generic_kernel<float>), dim3(nblk), dim3(256), 0, st, p, a_kmajor, b_kmajor);
    }
    XL_CHECK_LAUNCH();
    return XL_OK;
}
