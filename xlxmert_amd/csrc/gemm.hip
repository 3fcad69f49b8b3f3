// Dense contractions of the X-LXMERT path (nn.Linear forward, dX, dW) for gfx950.
//
//   C[M,N] = alpha * sum_k A(m,k) B(n,k) (+bias) -> epilogue     (see include/xlxmert_hip.h: xl_gemm)
//
// Two kernels:
//   gemm_bf16_mfma_kernel<AK, BKM, TR, BM, BN, WM, WN>
//       bf16 operands, fp32 accumulate on v_mfma_f32_32x32x16_bf16.  Block tile BM x BN x 64 with WM x WN waves
//       (128x128 / 2x2 waves for shapes with few tiles, 256x256 / 2x4 waves where the grid still fills the chip:
//       a 128x128 tile at full MFMA rate would need ~39 TB/s of L2->LDS traffic, more than the L2s deliver).
//       Operands are staged by LDS-DMA (global_load_lds, 16 B per lane) into a double-buffered LDS ring, one
//       barrier per K tile.  K-major operands ([rows][k]) sit row-major in LDS with a 16-byte-chunk XOR swizzle
//       (chunk ^= (row>>1)&7) so that ds_read_b128 fragment reads are conflict free; M-major operands ([k][rows];
//       the dX / dW contractions) sit as stored and become MFMA fragments through ds_read_b64_tr_b16 (LDS
//       transpose read) with a 64-byte XOR swizzle on k&3.  Because LDS-DMA writes lane-linear, both swizzles are
//       applied to the per-lane SOURCE address.  The MFMA k-slot <-> k mapping is applied identically to A and B,
//       which is all the contraction needs.
//   gemm_generic_kernel    any dtype / any alignment, fp32 FMA, 64x64x16 tile.  It is the exact-fp32
//       path (XL_F32: parity configuration) and the fallback for operands the MFMA loader cannot
//       take (leading dimension not a multiple of 8 elements).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace xl {

struct GemmParams {
    const void* A; const void* B; void* C;
    const float* bias; const void* residual; void* aux;
    int M, N, K, lda, ldb, ldc, ldr, ldx;
    int epilogue, out_f32, atomic_out, splitk, kper, vec_epi;
    float alpha, p_drop, inv_keep;
    uint64_t seed;
    int tiles_m, tiles_n, ablate;
};

// ------------------------------------------------------------------ scalar epilogue (generic kernel, ragged edges)
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
// so that the tiles co-resident on one XCD share A row panels and B column panels in its L2.
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
constexpr int BK = 64;

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf16_t;

__device__ __forceinline__ bf16x4_t lds_tr_read(const uint8_t* ptr) {
    auto p = (__attribute__((address_space(3))) v4bf16_t*)(ptr);
    v4bf16_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
    return __builtin_bit_cast(bf16x4_t, r);
}

// Operand tile of ROWS x 64(k) bf16 in LDS.
//   K-major: [row][k], row pitch 128 B, 16-byte chunk c of row r stored at chunk c ^ ((r>>1)&7).
//   M-major: [k][row], k-row pitch ROWS*2 B, byte b of k-row kr stored at b ^ ((kr&3)<<6).
template <bool KMAJ, int ROWS>
struct OpTile {
    static constexpr int BYTES = ROWS * BK * 2;
    static constexpr int RP = ROWS * 2;               // M-major k-row pitch (bytes)

    // (strided index, contiguous chunk) of the 16 bytes stored at LDS byte offset o of the tile
    __device__ static __forceinline__ void decode(int o, int& rs, int& c) {
        if (KMAJ) { rs = o >> 7; c = ((o >> 4) & 7) ^ ((rs >> 1) & 7); }
        else { rs = o / RP; c = ((o % RP) ^ ((rs & 3) << 6)) >> 4; }
    }
    __device__ static __forceinline__ int encode(int rs, int c) {
        if (KMAJ) return rs * 128 + ((c ^ ((rs >> 1) & 7)) << 4);
        return rs * RP + ((c << 4) ^ ((rs & 3) << 6));
    }
    // MFMA operand fragment: rows [r0, r0+32) (lane -> row l&31), k-slots s*16 + (l>>5)*8 + 0..7
    template <bool TR>
    __device__ static __forceinline__ bf16x8_t frag(const uint8_t* tile, int r0, int s, int lane) {
        if (KMAJ) {
            const int row = r0 + (lane & 31);
            return *reinterpret_cast<const bf16x8_t*>(tile + encode(row, s * 2 + (lane >> 5)));
        } else if (TR) {
            // 16-lane group reads a [4 k][16 rows] block; lane t supplies the address of k-row t>>2,
            // row-chunk (t&3)*4 and receives column t (4 consecutive k).
            const int t = lane & 15;
            const int mb = (r0 + ((lane >> 4) & 1) * 16 + (t & 3) * 4) * 2;
            const int k0r = s * 16 + (lane >> 5) * 8 + (t >> 2), k1r = k0r + 4;
            bf16x4_t lo = lds_tr_read(tile + k0r * RP + (mb ^ ((k0r & 3) << 6)));
            bf16x4_t hi = lds_tr_read(tile + k1r * RP + (mb ^ ((k1r & 3) << 6)));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            const int m = r0 + (lane & 31);
            bf16x8_t f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = s * 16 + (lane >> 5) * 8 + j;
                f[j] = *reinterpret_cast<const short*>(tile + k * RP + ((m * 2) ^ ((k & 3) << 6)));
            }
            return f;
        }
    }
};

// predicated (zero-filling) load of the 16 bytes that belong at LDS offset o: ragged last k-tile only
template <bool KMAJ, int ROWS>
__device__ __forceinline__ uint4 gload16(const bf16_t* __restrict__ P, int ld, int row0, int rows_ext, int k0, int kend, int o) {
    int rs, c;
    OpTile<KMAJ, ROWS>::decode(o, rs, c);
    int gr, gc, lim;
    bool ok;
    if (KMAJ) { gr = row0 + rs; gc = k0 + c * 8; ok = gr < rows_ext; lim = kend; }
    else { gr = k0 + rs; gc = row0 + c * 8; ok = gr < kend; lim = rows_ext; }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ok) {
        const bf16_t* src = P + (size_t)gr * ld + gc;
        if (gc + 8 <= lim) {
            v = *reinterpret_cast<const uint4*>(src);
        } else if (gc < lim) {
            bf16_t e[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = (gc + i < lim) ? src[i] : (bf16_t)0;
            v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
            v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
        }
    }
    return v;
}

template <bool AK, bool BKM, bool TR, bool DMA, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16_mfma_kernel(GemmParams p) {   // >= 2 waves / SIMD
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;       // wave tile
    constexpr int FM = WTM / 32, FN = WTN / 32;       // 32x32 MFMA fragments per wave
    using TA = OpTile<AK, BM>;
    using TB = OpTile<BKM, BN>;
    constexpr int STAGE = TA::BYTES + TB::BYTES;
    constexpr int PA = TA::BYTES / 1024 / NW, PB = TB::BYTES / 1024 / NW;     // 1 KiB DMA pieces per wave
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
    const int nfull = (kend - kbeg) / BK;          // k-tiles fully in range: staged by LDS-DMA
    // Direct-to-LDS staging: a wave instruction moves 64 x 16 B into 1 KiB of LDS at wave-uniform base + lane*16, so the
    // LDS image is lane-linear and the swizzles are applied to the per-lane SOURCE address.  Rows / row-chunks past
    // the operand extent are clamped (they only feed outputs that are never stored).
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
    uint4 ra[PA], rb[PB];                           // register staging (DMA == false, and the ragged last k-tile)
    auto stage_issue = [&](int kt, uint8_t* buf) {
        const int k0 = kbeg + kt * BK;
        if (DMA && kt < nfull) {
            const bf16_t* ga = A + (AK ? (size_t)k0 : (size_t)k0 * p.lda);
            const bf16_t* gb = B + (BKM ? (size_t)k0 : (size_t)k0 * p.ldb);
#pragma unroll
            for (int t = 0; t < PA; ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + srca[t]),
                                                 (__attribute__((address_space(3))) void*)(buf + (wave_u * PA + t) * 1024), 16, 0, 0);
#pragma unroll
            for (int t = 0; t < PB; ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + srcb[t]),
                                                 (__attribute__((address_space(3))) void*)(buf + TA::BYTES + (wave_u * PB + t) * 1024), 16, 0, 0);
        } else if (kt < nfull) {                      // register staging: same clamped sources, no predication
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
    auto stage_commit = [&](int kt, uint8_t* buf) {
        if (DMA && kt < nfull) return;
#pragma unroll
        for (int t = 0; t < PA; ++t) *reinterpret_cast<uint4*>(buf + (wave * PA + t) * 1024 + lane * 16) = ra[t];
#pragma unroll
        for (int t = 0; t < PB; ++t) *reinterpret_cast<uint4*>(buf + TA::BYTES + (wave * PB + t) * 1024 + lane * 16) = rb[t];
    };
    if (nkt > 0) { stage_issue(0, smem); stage_commit(0, smem); }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const uint8_t* ta = smem + (kt & 1) * STAGE;
        const uint8_t* tb = ta + TA::BYTES;
        const bool more = kt + 1 < nkt && !(p.ablate & 2);
        uint8_t* nbuf = smem + ((kt + 1) & 1) * STAGE;
        // the other buffer was last read in iteration kt-1 (barrier passed): refill it while this one is consumed
        if (more) stage_issue(kt + 1, nbuf);
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
        if (more) stage_commit(kt + 1, nbuf);
        __syncthreads();                              // drains this wave's DMA (vmcnt(0)) and publishes the refill
    }
    if (p.ablate & 4) return;
    // ---- epilogue.  C/D layout of v_mfma_f32_32x32x16: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    if (p.atomic_out) {
        // split-K / accumulate (weight gradients; epilogue NONE): fp32 atomics straight from the accumulators -- a wave
        // instruction covers 2 rows x 32 consecutive columns (2 cache lines), which is what the L2 atomic units want.
        float* Cf = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int nn = n0 + wn + j * 32 + (lane & 31);
                const float bb = (z == 0 && p.bias != nullptr && nn < p.N) ? p.bias[nn] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mm = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (mm < p.M && nn < p.N) atomicAdd(Cf + (size_t)mm * p.ldc + nn, acc[i][j][r] * p.alpha + bb);
                }
            }
        return;
    }
    // Each wave transposes 64x64 fp32 sub-tiles through its own 16 KiB of the (now idle) staging LDS so that a lane ends
    // up with 8 consecutive columns of one row: bias / residual / aux are read and C is written with 16-byte
    // accesses.  16-byte chunks are XOR-swizzled by (row & 15): conflict-free both ways.
    float* wbuf = reinterpret_cast<float*>(smem + wave * 16384);
    const int c8 = lane & 7, rr = lane >> 3;
    const bool add_bias = (z == 0) && p.bias != nullptr;
    auto quad = [&](auto HI, auto HJ) {
            constexpr int hi = decltype(HI)::value, hj = decltype(HJ)::value;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = j * 32 + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        wbuf[row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3))] = acc[hi * 2 + i][hj * 2 + j][r];
                    }
                }
            __builtin_amdgcn_wave_barrier();
            const int n = n0 + wn + hj * 64 + c8 * 8;
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = (add_bias && n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int row = ps * 8 + rr;
                const int m = m0 + wm + hi * 64 + row;
                const float4 lo = *reinterpret_cast<const float4*>(wbuf + row * 64 + (((2 * c8) ^ (row & 15)) << 2));
                const float4 hi4 = *reinterpret_cast<const float4*>(wbuf + row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2));
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
                if (m >= p.M || n >= p.N) continue;
                if (p.vec_epi && n + 8 <= p.N) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bv[e];
                    const size_t mn = (size_t)m;
                    if (p.epilogue == XL_EPI_GELU) {
                        stvec(reinterpret_cast<bf16_t*>(p.aux) + mn * p.ldx + n, v);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
                    } else if (p.epilogue == XL_EPI_RESIDUAL) {
                        float rv[8];
                        ldvec(reinterpret_cast<const bf16_t*>(p.residual) + mn * p.ldr + n, rv);
                        if (p.p_drop > 0.0f) {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                v[e] *= dropout_scale(p.seed, (uint64_t)m * (uint64_t)p.N + n + e, p.p_drop, p.inv_keep);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rv[e];
                    } else if (p.epilogue == XL_EPI_DGELU) {
                        float av[8];
                        ldvec(reinterpret_cast<const bf16_t*>(p.aux) + mn * p.ldx + n, av);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_fast(av[e]);
                    } else if (p.epilogue == XL_EPI_TANH) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
                    }
                    if (p.out_f32) {
                        float* c = reinterpret_cast<float*>(p.C) + mn * p.ldc + n;
                        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
                        stvec(reinterpret_cast<bf16_t*>(p.C) + mn * p.ldc + n, v);
                    }
                } else {
#pragma unroll 1
                    for (int e = 0; e < 8; ++e)
                        if (n + e < p.N) epilogue_store<bf16_t>(p, m, n + e, v[e], z == 0);
                }
            }
    };
    // 64x64 quads of the wave tile, compile-time indices (the accumulators must stay in registers)
    quad(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    if constexpr (FN / 2 > 1) quad(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    if constexpr (FM / 2 > 1) {
        quad(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
        if constexpr (FN / 2 > 1) quad(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    }
    static_assert(FM / 2 <= 2 && FN / 2 <= 2, "epilogue handles up to 2x2 quads per wave");
}

template <bool AK, bool BKM, bool TR, bool DMA, int BM, int BN, int WM, int WN>
static hipError_t launch_one(const GemmParams& p, int nblk, hipStream_t st) {
    constexpr int lds = 2 * (BM + BN) * BK * 2;
    hipError_t e = hipSuccess;
    auto k = gemm_bf16_mfma_kernel<AK, BKM, TR, DMA, BM, BN, WM, WN>;
    static bool attr = false;
    if (lds > 65536 && !attr) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL(k, dim3(nblk), dim3(WM * WN * 64), lds, st, p);
    return e;
}

int g_gemm_dma = -1;     // staging mode: 1 LDS-DMA (global_load_lds), 0 registers; -1 = read XL_GEMM_DMA (default 0)

template <bool AK, bool BKM, int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const GemmParams& p, int nblk, hipStream_t st) {
    if (g_use_tr_read) {
        if (g_gemm_dma) return launch_one<AK, BKM, true, true, BM, BN, WM, WN>(p, nblk, st);
        return launch_one<AK, BKM, true, false, BM, BN, WM, WN>(p, nblk, st);
    }
    if (g_gemm_dma) return launch_one<AK, BKM, false, true, BM, BN, WM, WN>(p, nblk, st);
    return launch_one<AK, BKM, false, false, BM, BN, WM, WN>(p, nblk, st);
}

template <bool AK, bool BKM>
static hipError_t launch_mfma(const GemmParams& p, int tile, int nblk, hipStream_t st) {
    if (tile == 256) return launch_cfg<AK, BKM, 256, 256, 2, 4>(p, nblk, st);
    return launch_cfg<AK, BKM, 128, 128, 2, 2>(p, nblk, st);
}

static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
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

    static const int force_tile = env_int("XL_GEMM_TILE", 0);          // tuning / debug overrides
    static const int ablate = env_int("XL_GEMM_ABLATE", 0);
    if (g_gemm_dma < 0) g_gemm_dma = env_int("XL_GEMM_DMA", 0);
    static const int big_min_blocks = env_int("XL_GEMM_BIG_MIN_BLOCKS", 512);

    GemmParams p;
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.residual = residual; p.aux = aux;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldx = ldx;
    p.epilogue = epilogue; p.out_f32 = (out_dtype == XL_F32); p.alpha = alpha;
    p.p_drop = p_drop; p.inv_keep = 1.0f / (1.0f - p_drop); p.seed = seed; p.ablate = ablate;

    const bool mfma_ok = in_dtype == XL_BF16 && (lda % 8 == 0) && (ldb % 8 == 0) && aligned16(A) && aligned16(B);
    const bool may_split = mfma_ok && out_dtype == XL_F32 && epilogue == XL_EPI_NONE;
    // tile choice: 256x256 halves the L2->LDS traffic per flop but needs enough tiles to fill 256 CUs (1 block/CU)
    int tile = mfma_ok ? 128 : 64;
    if (mfma_ok) {
        const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
        // measured (tools/gemm_bench.py): 256x256 wins only when >= 2 full rounds of tiles AND a deep K amortise its
        // longer prologue/epilogue (the two 10k-codebook contractions); 128x128 wins everywhere else
        if (t256 >= big_min_blocks && K >= 2048 && !may_split) tile = 256;
        if (force_tile == 128 || force_tile == 256) tile = force_tile;
    }
    p.tiles_m = (M + tile - 1) / tile;
    p.tiles_n = (N + tile - 1) / tile;
    const int tiles = p.tiles_m * p.tiles_n;
    // split-K only for the weight-gradient shape (fp32 out, plain epilogue): few output tiles, deep K
    int splitk = 1;
    const int want = tile == 256 ? 512 : 768;
    if (may_split && tiles < want && K >= 1024) {
        splitk = (want + tiles - 1) / tiles;
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
    // 16-byte epilogue accesses need 8-element (bf16) / 4-element (fp32) aligned rows of C / residual / aux
    p.vec_epi = aligned16(C) && (out_dtype == XL_F32 ? ldc % 4 == 0 : ldc % 8 == 0);
    if (epilogue == XL_EPI_RESIDUAL) p.vec_epi = p.vec_epi && aligned16(residual) && ldr % 8 == 0;
    if (epilogue == XL_EPI_GELU || epilogue == XL_EPI_DGELU) p.vec_epi = p.vec_epi && aligned16(aux) && ldx % 8 == 0;
    if (mfma_ok) {
        hipError_t e;
        if (a_kmajor && b_kmajor) e = launch_mfma<true, true>(p, tile, nblk, st);
        else if (a_kmajor && !b_kmajor) e = launch_mfma<true, false>(p, tile, nblk, st);
        else if (!a_kmajor && b_kmajor) e = launch_mfma<false, true>(p, tile, nblk, st);
        else e = launch_mfma<false, false>(p, tile, nblk, st);
        XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    } else if (in_dtype == XL_BF16) {
        hipLaunchKernelGGL((gemm_generic_kernel<bf16_t>), dim3(nblk), dim3(256), 0, st, p, a_kmajor, b_kmajor);
    } else {
        hipLaunchKernelGGL((gemm_generic_kernel<float>), dim3(nblk), dim3(256), 0, st, p, a_kmajor, b_kmajor);
    }
    XL_CHECK_LAUNCH();
    return XL_OK;
}
