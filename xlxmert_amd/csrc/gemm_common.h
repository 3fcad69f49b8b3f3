// Shared pieces of the dense-contraction kernels (gemm.hip: generic + 128x128 MFMA kernel and the host entry point;
// gemm_pp.hip: 256x256 ping-pong kernel): launch parameters, XCD-aware tile order, LDS operand tiles, epilogues.
#pragma once
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
    const uint64_t* step_seed;        // device-resident step part of the dropout seed (common.h with_step_seed), or null
    int tiles_m, tiles_n, ablate;
    unsigned long long* trace;        // debug: per-block timestamps (xl_gemm_trace), normally null
    float* colsum_ws;                 // fused column sums of C: one partial slab [N] per wave tile (64 / 128 rows; fast epilogue only)
    float* slab;                      // split-K through slabs (slab_exchange): partial tiles [tile][split][64 Ki floats], or null
    int* tickets;                     // ... and one arrival counter per output tile (zero between launches)
    int tail_tiles, tail_kper;        // tail split (ping-pong kernel): the last tail_tiles tiles run as `splitk` K slices each
    int overwrite;                    // grouped weight gradients, one writer per tile: store instead of read-modify-write
    int slab_det;                     // slab exchange: the last arriver sums ALL slices from memory in slice order (its own included)
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
            if (p.p_drop > 0.0f) v *= dropout_scale(with_step_seed(p.seed, p.step_seed), (uint32_t)m, (uint32_t)n, p.p_drop, p.inv_keep);
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
        case XL_EPI_GELU_DG: {
            TIn* aux = reinterpret_cast<TIn*>(p.aux);
            Elem<TIn>::st(aux + (size_t)m * p.ldx + n, gelu_erf_grad(v));
            v = gelu_erf(v);
            break;
        }
        case XL_EPI_MULAUX: {
            const TIn* aux = reinterpret_cast<const TIn*>(p.aux);
            v *= Elem<TIn>::ld(aux + (size_t)m * p.ldx + n);
            break;
        }
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
__device__ __forceinline__ int linear_block(int nblk = gridDim.x) {
    const int b = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, pos = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
}
__device__ __forceinline__ void tile_of(const GemmParams& p, int t, int& tm, int& tn);
__device__ __forceinline__ void tile_coords(const GemmParams& p, int& tm, int& tn, int& z) {
    const int L = linear_block();
    const int tiles = p.tiles_m * p.tiles_n;
    z = L / tiles;
    tile_of(p, L - z * tiles, tm, tn);
}
__device__ __forceinline__ void tile_of(const GemmParams& p, int t, int& tm, int& tn) {
    constexpr int GM = 8;
    const int per_group = GM * p.tiles_n;
    const int g = t / per_group;
    const int in_g = t - g * per_group;
    const int gsize = min(GM, p.tiles_m - g * GM);
    tn = in_g / gsize;
    tm = g * GM + (in_g - tn * gsize);
}

// ================================================================== bf16 MFMA building blocks
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
//   M-major: [k][row], k-row pitch ROWS*2 B, byte b of k-row kr stored at b ^ swz(kr): a transpose read takes 64 bytes
//            of each of four consecutive k-rows at once, which must fall into the four quarters of the 256-byte bank
//            period -- ROWS = 128 (pitch 256 B: every k-row starts a period): (kr&3)<<6; ROWS = 64 (pitch 128 B: k-rows
//            kr and kr+2 alias): ((kr>>1)&1)<<6.
template <bool KMAJ, int ROWS>
struct OpTile {
    static_assert(ROWS == 128 || ROWS == 64, "operand tile of 64 or 128 rows");
    static constexpr int BYTES = ROWS * BK * 2;
    static constexpr int RP = ROWS * 2;               // M-major k-row pitch (bytes)
    __device__ static __forceinline__ int swz(int kr) { return ROWS == 128 ? ((kr & 3) << 6) : (((kr >> 1) & 1) << 6); }

    // (strided index, contiguous chunk) of the 16 bytes stored at LDS byte offset o of the tile
    __device__ static __forceinline__ void decode(int o, int& rs, int& c) {
        if (KMAJ) { rs = o >> 7; c = ((o >> 4) & 7) ^ ((rs >> 1) & 7); }
        else { rs = o / RP; c = ((o % RP) ^ swz(rs)) >> 4; }
    }
    __device__ static __forceinline__ int encode(int rs, int c) {
        if (KMAJ) return rs * 128 + ((c ^ ((rs >> 1) & 7)) << 4);
        return rs * RP + ((c << 4) ^ swz(rs));
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
            bf16x4_t lo = lds_tr_read(tile + k0r * RP + (mb ^ swz(k0r)));
            bf16x4_t hi = lds_tr_read(tile + k1r * RP + (mb ^ swz(k1r)));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            const int m = r0 + (lane & 31);
            bf16x8_t f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = s * 16 + (lane >> 5) * 8 + j;
                f[j] = *reinterpret_cast<const short*>(tile + k * RP + ((m * 2) ^ swz(k)));
            }
            return f;
        }
    }
};

// M-major fragments next to IN-FLIGHT LDS-DMA (ping-pong kernel).  In front of the ds_read_b64_tr_b16 BUILTIN hipcc places an
// `s_waitcnt vmcnt(0)` whenever `buffer_load ... lds` requests are outstanding -- it cannot see that the ring slots being read
// and the slots being filled differ -- which drains the whole prefetch once (dX layout) or twice (weight gradients) per K tile:
// the LDS-DMA then has ONE phase to land instead of three, and that, not the second read instruction per fragment, was most of
// the 1.73 us per K tile of the weight-gradient kernel against 1.44 for the K-major layout (whose plain ds_read_b128 gets no such
// wait).  Issued through inline assembly the transpose reads are invisible to that pass; what orders them is what orders the
// K-major reads too -- the kernel's own protocol: a slot is read only after the counted vmcnt + barrier that follow its fill,
// and every read is retired (the explicit lgkmcnt(0) of the phase) before the barrier that lets the slot be refilled.
//   tr_lane_off: byte offset, inside an M-major tile, of the lane's first transpose read of the s = 0 fragment of rows r0..r0+31
//   TrFrag<RP, BASE>::get<S>(addr): fragment s = S; addr = LDS byte address of the BUFFER + tr_lane_off, BASE = the tile's
//   constant byte offset inside the buffer (goes into the instruction's 16-bit offset field together with the s part)
template <int ROWS>
__device__ __forceinline__ uint32_t tr_lane_off(int r0, int lane) {
    using T = OpTile<false, ROWS>;
    const int t = lane & 15;
    const int mb = (r0 + ((lane >> 4) & 1) * 16 + (t & 3) * 4) * 2;
    const int k0r = (lane >> 5) * 8 + (t >> 2);
    return (uint32_t)(k0r * T::RP + (mb ^ T::swz(k0r)));
}
template <int RP, int BASE>
struct TrFrag {
    template <int S>
    __device__ static __forceinline__ bf16x8_t get(uint32_t addr) {
        static_assert(BASE + S * 16 * RP + 4 * RP < 65536, "DS offset field");
        bf16x4_t lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(BASE + S * 16 * RP));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(BASE + S * 16 * RP + 4 * RP));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
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

// ---- epilogues of the MFMA kernels.  C/D layout of v_mfma_f32_32x32x16: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// split-K / accumulate (weight gradients; epilogue NONE): fp32 atomics straight from one 32x32 accumulator -- a wave
// instruction covers 2 rows x 32 consecutive columns (2 cache lines), which is what the L2 atomic units want.
__device__ __forceinline__ void epilogue_atomic_frag(const GemmParams& p, int lane, bool first, int mf, int nf, const f32x16_t& acc) {
    float* Cf = reinterpret_cast<float*>(p.C);
    const int nn = nf + (lane & 31);
    const float bb = (first && p.bias != nullptr && nn < p.N) ? p.bias[nn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = mf + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (mm < p.M && nn < p.N) atomicAdd(Cf + (size_t)mm * p.ldc + nn, acc[r] * p.alpha + bb);
    }
}

// One 64x64 fp32 sub-tile (2x2 accumulators, origin (mq, nq)) goes through 16 KiB of wave-private LDS so that a lane ends
// up with 8 consecutive columns of one row (lane -> row ps*8 + lane/8, columns (lane&7)*8 ..): bias / residual / aux are
// read and C is written with 16-byte accesses.  Plain row-major image: the 32-lane writes are conflict-free, the 16-byte
// reads are 2-way conflicted (cheap: 16 reads per quad), and every LDS address is one lane base + an immediate offset --
// a per-row XOR swizzle costs 32 address registers here and pushed the kernel into scratch.
__device__ __forceinline__ void quad_to_lds(float* wbuf, int lane, const f32x16_t& a00, const f32x16_t& a01,
                                            const f32x16_t& a10, const f32x16_t& a11) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x16_t& a = i == 0 ? (j == 0 ? a00 : a01) : (j == 0 ? a10 : a11);
            const int col = j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                wbuf[row * 64 + col] = a[r];
            }
        }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void quad_row_from_lds(const float* wbuf, int row, int c8, float (&v)[8]) {
    const float4 lo = *reinterpret_cast<const float4*>(wbuf + row * 64 + c8 * 8);
    const float4 hi = *reinterpret_cast<const float4*>(wbuf + row * 64 + c8 * 8 + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}

// Generic quad epilogue: runtime epilogue kind, ragged edges, unaligned C (scalar fallback).
__device__ __forceinline__ void epilogue_quad(const GemmParams& p, float* wbuf, int lane, bool first, int mq, int nq,
                                              const f32x16_t& a00, const f32x16_t& a01, const f32x16_t& a10, const f32x16_t& a11) {
    const int c8 = lane & 7, rr = lane >> 3;
    const bool add_bias = first && p.bias != nullptr;
    quad_to_lds(wbuf, lane, a00, a01, a10, a11);
    const int n = nq + c8 * 8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (add_bias && n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int row = ps * 8 + rr;
        const int m = mq + row;
        float v[8];
        quad_row_from_lds(wbuf, row, c8, v);
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
                        v[e] *= dropout_scale(with_step_seed(p.seed, p.step_seed), (uint32_t)m, (uint32_t)(n + e), p.p_drop, p.inv_keep);
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
            } else if (p.epilogue == XL_EPI_GELU_DG) {
                float gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { gv[e] = gelu_grad_fast(v[e]); v[e] = gelu_fast(v[e]); }
                stvec(reinterpret_cast<bf16_t*>(p.aux) + mn * p.ldx + n, gv);
            } else if (p.epilogue == XL_EPI_MULAUX) {
                float av[8];
                ldvec(reinterpret_cast<const bf16_t*>(p.aux) + mn * p.ldx + n, av);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= av[e];
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
                if (n + e < p.N) epilogue_store<bf16_t>(p, m, n + e, v[e], first);
        }
    }
}

// Fast quad epilogue for interior tiles with 16-byte-aligned C / residual / aux (host: p.vec_epi, device: tile fully in
// range).  The epilogue kind is a template parameter, the operand rows (residual or aux) of the whole quad are requested
// BEFORE the LDS transpose and nothing waits on a store: branch-free, one load latency per quad instead of eight.
struct QuadOperand { uint4 row[8]; };      // residual (RESIDUAL) or saved pre-activation (DGELU): 8 rows x 8 bf16 per lane

// ---- the same pieces for a sub-tile of 64 rows x W columns (W = 64: the quad above; W = 32: the third accumulator column of
// the 256x192 kernel's 64x96 wave tile): W/8 lanes share a row, 512/W rows per pass, W/8 passes.
// 64 rows x 32 columns (two accumulators stacked in M) -> wave-private LDS, row pitch 32 floats
__device__ __forceinline__ void half_to_lds(float* wbuf, int lane, const f32x16_t& a0, const f32x16_t& a1) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x16_t& a = i == 0 ? a0 : a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            wbuf[row * 32 + (lane & 31)] = a[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
}
template <int W>
__device__ __forceinline__ void sub_row_from_lds(const float* wbuf, int row, int c8, float (&v)[8]) {
    const float4 lo = *reinterpret_cast<const float4*>(wbuf + row * W + c8 * 8);
    const float4 hi = *reinterpret_cast<const float4*>(wbuf + row * W + c8 * 8 + 4);
    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
}
template <int EPI, int W, int ROWS = 64>
__device__ __forceinline__ void sub_operand_load(const GemmParams& p, int lane, int mq, int nq, QuadOperand& op) {
    if constexpr (EPI == XL_EPI_RESIDUAL || EPI == XL_EPI_DGELU || EPI == XL_EPI_MULAUX) {
        constexpr int LPR = W / 8, RPP = 64 / LPR, NPS = ROWS / RPP;
        const bf16_t* src = reinterpret_cast<const bf16_t*>(EPI == XL_EPI_RESIDUAL ? p.residual : p.aux);
        const int ld = EPI == XL_EPI_RESIDUAL ? p.ldr : p.ldx;
        const bf16_t* s0 = src + (size_t)(mq + lane / LPR) * ld + nq + (lane % LPR) * 8;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) op.row[ps] = *reinterpret_cast<const uint4*>(s0 + (size_t)(ps * RPP) * ld);
    }
}
template <int W>
__device__ __forceinline__ void sub_load_bias8(const GemmParams& p, int lane, bool first, int nq, float (&bv)[8]) {
    const int n = nq + (lane % (W / 8)) * 8;
    if (first && p.bias != nullptr) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    }
}

template <int EPI, int NPS = 8>           // NPS row groups of 8 rows: 8 = a 64 x 64 quad, 4 = a 32 x 64 half quad
__device__ __forceinline__ void quad_operand_load(const GemmParams& p, int lane, int mq, int nq, QuadOperand& op) {
    if constexpr (EPI == XL_EPI_RESIDUAL || EPI == XL_EPI_DGELU || EPI == XL_EPI_MULAUX) {
        const bf16_t* src = reinterpret_cast<const bf16_t*>(EPI == XL_EPI_RESIDUAL ? p.residual : p.aux);
        const int ld = EPI == XL_EPI_RESIDUAL ? p.ldr : p.ldx;
        const bf16_t* s0 = src + (size_t)(mq + (lane >> 3)) * ld + nq + (lane & 7) * 8;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) op.row[ps] = *reinterpret_cast<const uint4*>(s0 + (size_t)(ps * 8) * ld);
    }
}

// 32 rows x 32 columns (one accumulator) -> 4 KiB of wave-private LDS, row pitch 32 floats (gemm_q.hip: third accumulator of a 32 x 96 wave tile)
__device__ __forceinline__ void acc32_to_lds(float* wbuf, int lane, const f32x16_t& a) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) wbuf[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = a[r];
    __builtin_amdgcn_wave_barrier();
}

// 32 rows x 64 columns (two accumulators side by side) -> 8 KiB of wave-private LDS, row pitch 64 floats (the quad image's upper half)
__device__ __forceinline__ void hquad_to_lds(float* wbuf, int lane, const f32x16_t& a0, const f32x16_t& a1) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x16_t& a = j == 0 ? a0 : a1;
        const int col = j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) wbuf[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 64 + col] = a[r];
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void unpack8(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

// rows of a quad already in LDS (quad_to_lds) -> epilogue math -> 16-byte stores
// Column sums of a wave's output rows, held as 8 per-lane partials (columns (lane&7)*8 + e over the rows this lane
// stored): halving butterfly over the three row bits of the lane id (v_permlane32_swap, v_permlane16_swap, one
// bpermute) -> every lane ends up with ONE finished column, written to the wave's slab of the workspace.
__device__ __forceinline__ void swap_add(float& keep_lo, float& keep_hi, bool rows16) {
    const uint32_t a = __float_as_uint(keep_lo), b = __float_as_uint(keep_hi);
    const auto r = rows16 ? __builtin_amdgcn_permlane16_swap(a, b, false, false) : __builtin_amdgcn_permlane32_swap(a, b, false, false);
    keep_lo = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void colsum_flush(const GemmParams& p, int lane, int slab, int nq, float (&cs)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) swap_add(cs[e], cs[e + 4], false);      // lane bit 5: low half keeps columns 0-3, high 4-7
#pragma unroll
    for (int e = 0; e < 2; ++e) swap_add(cs[e], cs[e + 2], true);       // lane bit 4
    const float x0 = __shfl_xor(cs[0], 8, 64), x1 = __shfl_xor(cs[1], 8, 64);
    const int b3 = (lane >> 3) & 1;
    const float tot = b3 ? cs[1] + x1 : cs[0] + x0;                     // lane bit 3
    const int col = ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + b3;
    p.colsum_ws[(size_t)slab * p.N + nq + (lane & 7) * 8 + col] = tot;
}

// the 8 bias values of a lane's column segment in the fast epilogue (zeros when the launch has no bias or this is not the
// first K split)
__device__ __forceinline__ void load_bias8(const GemmParams& p, int lane, bool first, int nq, float (&bv)[8]) {
    const int n = nq + (lane & 7) * 8;
    if (first && p.bias != nullptr) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    }
}

// the dropout seed of a launch (site seed + device-resident step part): ONE fetch per workgroup, ahead of the K loop
template <int EPI>
__device__ __forceinline__ uint64_t dropout_seed_of(const GemmParams& p) {
    if constexpr (EPI == XL_EPI_RESIDUAL) return p.p_drop > 0.0f ? with_step_seed(p.seed, p.step_seed) : 0;
    else return 0;
}

template <int EPI, int NPS = 8>
__device__ __forceinline__ void epilogue_rows_fast(const GemmParams& p, const float* wbuf, int lane, bool first, int mq, int nq,
                                                   const QuadOperand& op, float (&cs)[8], const float (&bv)[8], uint64_t seed) {
    const int c8 = lane & 7, rr = lane >> 3;
    const int n = nq + c8 * 8;
    const bool drop = p.p_drop > 0.0f;      // (seed: dropout_seed_of(p), fetched by the caller BEFORE its K loop -- fetched here, the
#pragma unroll                              //  load sat behind a vmcnt(0) that also drained every store of the previous row groups)
    for (int ps = 0; ps < NPS; ++ps) {
        const int row = ps * 8 + rr;
        const size_t m = (size_t)(mq + row);
        float v[8];
        quad_row_from_lds(wbuf, row, c8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bv[e];
        if constexpr (EPI == XL_EPI_GELU) {
            stvec(reinterpret_cast<bf16_t*>(p.aux) + m * p.ldx + n, v);
            gelu_fast8(v);
        } else if constexpr (EPI == XL_EPI_RESIDUAL) {
            float rv[8];
            unpack8(op.row[ps], rv);
            if (drop) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dropout_scale(seed, (uint32_t)m, (uint32_t)(n + e), p.p_drop, p.inv_keep);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
        } else if constexpr (EPI == XL_EPI_DGELU) {
            float av[8];
            unpack8(op.row[ps], av);
            gelu_grad_mul8(v, av);
        } else if constexpr (EPI == XL_EPI_GELU_DG) {
            float gv[8];
            gelu_fast8_dg(v, gv);
            stvec(reinterpret_cast<bf16_t*>(p.aux) + m * p.ldx + n, gv);
        } else if constexpr (EPI == XL_EPI_MULAUX) {
            float av[8];
            unpack8(op.row[ps], av);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= av[e];
        }
        if constexpr (EPI == XL_EPI_ROWMAX) {
            // (max, sum exp(x - max), argmax) of this row's 64-column segment: 8 columns per lane, then the 8 lanes of the row
            float mx = v[0];
            int idx = n;
#pragma unroll
            for (int e = 1; e < 8; ++e) if (v[e] > mx) { mx = v[e]; idx = n + e; }
            float se = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) se += __expf(v[e] - mx);
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                const float omx = __shfl_xor(mx, o, 64), ose = __shfl_xor(se, o, 64);
                const int oi = __shfl_xor(idx, o, 64);
                const float nm = fmaxf(mx, omx);
                se = se * __expf(mx - nm) + ose * __expf(omx - nm);
                idx = (omx > mx || (omx == mx && oi < idx)) ? oi : idx;
                mx = nm;
            }
            if (c8 == 0)
                reinterpret_cast<float4*>(p.aux)[(size_t)(nq >> 6) * p.M + m] = make_float4(mx, se, __int_as_float(idx), 0.f);
        } else if (p.out_f32) {
            float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] += v[e];
        } else {
            uint4 t;
            t.x = pack2bf(v[0], v[1]); t.y = pack2bf(v[2], v[3]); t.z = pack2bf(v[4], v[5]); t.w = pack2bf(v[6], v[7]);
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n) = t;
            if (p.colsum_ws != nullptr) {           // sums of the values as stored (what a separate pass over C would read)
                float sv[8];
                unpack8(t, sv);
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] += sv[e];
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // one row group at a time: interleaving all eight spills
    }
}

// rows of a 64 x W sub-tile already in LDS (sub_to_lds) -> epilogue math -> 16-byte stores (no fused column sums)
template <int EPI, int W, bool CS = false, int ROWS = 64>
__device__ __forceinline__ void sub_rows_fast(const GemmParams& p, const float* wbuf, int lane, int mq, int nq,
                                              const QuadOperand& op, const float (&bv)[8], uint64_t seed, float (&cs)[8]) {
    constexpr int LPR = W / 8, RPP = 64 / LPR, NPS = ROWS / RPP;
    const int c8 = lane % LPR, rr = lane / LPR;
    const int n = nq + c8 * 8;
    const bool drop = p.p_drop > 0.0f;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
        const int row = ps * RPP + rr;
        const size_t m = (size_t)(mq + row);
        float v[8];
        sub_row_from_lds<W>(wbuf, row, c8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + bv[e];
        if constexpr (EPI == XL_EPI_GELU) {
            stvec(reinterpret_cast<bf16_t*>(p.aux) + m * p.ldx + n, v);
            gelu_fast8(v);
        } else if constexpr (EPI == XL_EPI_RESIDUAL) {
            float rv[8];
            unpack8(op.row[ps], rv);
            if (drop) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dropout_scale(seed, (uint32_t)m, (uint32_t)(n + e), p.p_drop, p.inv_keep);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
        } else if constexpr (EPI == XL_EPI_DGELU) {
            float av[8];
            unpack8(op.row[ps], av);
            gelu_grad_mul8(v, av);
        } else if constexpr (EPI == XL_EPI_GELU_DG) {
            float gv[8];
            gelu_fast8_dg(v, gv);
            stvec(reinterpret_cast<bf16_t*>(p.aux) + m * p.ldx + n, gv);
        } else if constexpr (EPI == XL_EPI_MULAUX) {
            float av[8];
            unpack8(op.row[ps], av);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= av[e];
        }
        if (p.out_f32) {
            float* c = reinterpret_cast<float*>(p.C) + m * p.ldc + n;
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            uint4 t;
            t.x = pack2bf(v[0], v[1]); t.y = pack2bf(v[2], v[3]); t.z = pack2bf(v[4], v[5]); t.w = pack2bf(v[6], v[7]);
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n) = t;
            if constexpr (CS) {                     // sums of the values as stored (persistent kernel: fused column sums)
                float sv[8];
                unpack8(t, sv);
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] += sv[e];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int EPI, int W>
__device__ __forceinline__ void sub_rows_fast(const GemmParams& p, const float* wbuf, int lane, int mq, int nq,
                                              const QuadOperand& op, const float (&bv)[8], uint64_t seed) {
    float none[8];
    sub_rows_fast<EPI, W, false>(p, wbuf, lane, mq, nq, op, bv, seed, none);
}

template <int EPI>
__device__ __forceinline__ void epilogue_quad_fast(const GemmParams& p, float* wbuf, int lane, bool first, int mq, int nq,
                                                   const QuadOperand& op, const f32x16_t& a00, const f32x16_t& a01,
                                                   const f32x16_t& a10, const f32x16_t& a11) {
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bv[8];
    load_bias8(p, lane, first, nq, bv);
    quad_to_lds(wbuf, lane, a00, a01, a10, a11);
    epilogue_rows_fast<EPI>(p, wbuf, lane, first, mq, nq, op, cs, bv, dropout_seed_of<EPI>(p));
    if (p.colsum_ws != nullptr) colsum_flush(p, lane, mq >> 6, nq, cs);          // one slab per 64 rows
}

// ---- split-K without atomics on the output (ping-pong kernel).  The splits of one output tile meet in memory: every
// workgroup writes its partial accumulators to its slab (lane-linear 16-byte stores: the reader has the same thread ->
// element mapping, so the layout is private), publishes it (agent-scope release), and takes a ticket; the LAST arriver
// acquires, adds the other slabs to its registers and runs the ordinary epilogue -- ONE pass over the output (plain 16-byte
// stores, or read-modify-write for an accumulating weight gradient) instead of one pass of fp32 atomics per split (measured:
// ~45 us of a 110 us weight-gradient launch).  Placement-independent: nothing is assumed about which CU / XCD runs which
// split; no workgroup ever waits for another.  The last arriver resets the ticket for the next launch on the stream.
constexpr size_t SLAB_FLOATS = 65536;                 // 512 threads x 128 accumulator registers
constexpr size_t SLAB_TICKET_BYTES = 16384;           // 4096 tickets in front of the slabs
template <int R, int C>
__device__ __forceinline__ bool slab_exchange(const GemmParams& p, uint8_t* smem, int tid, int tile, int z, int nz,
                                              f32x16_t (&acc)[R][C]) {
    float4* mine = reinterpret_cast<float4*>(p.slab + ((size_t)tile * p.splitk + z) * SLAB_FLOATS) + tid;
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int j = 0; j < C; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                mine[(size_t)((i * C + j) * 4 + q) * 512] =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    volatile int* flag = reinterpret_cast<volatile int*>(smem);          // the staging ring is dead: all fragment reads are behind a barrier
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the write-back must not be overtaken by the ticket
        const int t = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == nz - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *flag = t;
    }
    __syncthreads();
    const int ticket = *flag;
    __syncthreads();                  // the ticket word lies in wave 0's epilogue buffer: everyone has read it before anyone goes on
    if (ticket != nz - 1) return false;
    // slab_det (launches whose OUTPUT feeds the forward pass: K split of a launch with an epilogue): which slice arrives last varies
    // from run to run, and fp32 addition is not associative -- so the last arriver drops its registers and adds every slice's slab
    // in slice order, its own (already written above, still in L2) included: the same bits every run
    if (p.slab_det) {
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < C; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    for (int o = 0; o < nz; ++o) {
        if (o == z && !p.slab_det) continue;
        const float4* s = reinterpret_cast<const float4*>(p.slab + ((size_t)tile * p.splitk + o) * SLAB_FLOATS) + tid;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float4 v[C][4];
#pragma unroll
            for (int j = 0; j < C; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[j][q] = s[(size_t)((i * C + j) * 4 + q) * 512];
#pragma unroll
            for (int j = 0; j < C; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[i][j][4 * q] += v[j][q].x; acc[i][j][4 * q + 1] += v[j][q].y;
                    acc[i][j][4 * q + 2] += v[j][q].z; acc[i][j][4 * q + 3] += v[j][q].w;
                }
        }
    }
    return true;
}

// fp32 output quad of the last arriver: C (+)= alpha * acc through the LDS transposition, 16-byte accesses (interior tiles,
// 16-byte-aligned rows).  rmw: accumulate into C (weight gradients) -- a plain read-modify-write, every element of the
// tile belongs to this workgroup alone.
__device__ __forceinline__ void epilogue_quad_accum(const GemmParams& p, float* wbuf, int lane, int mq, int nq, bool rmw,
                                                    const f32x16_t& a00, const f32x16_t& a01, const f32x16_t& a10, const f32x16_t& a11) {
    const int c8 = lane & 7, rr = lane >> 3;
    float* c0 = reinterpret_cast<float*>(p.C) + (size_t)(mq + rr) * p.ldc + nq + c8 * 8;
    float4 old[8][2];
    if (rmw) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            old[ps][0] = *reinterpret_cast<const float4*>(c0 + (size_t)(ps * 8) * p.ldc);
            old[ps][1] = *reinterpret_cast<const float4*>(c0 + (size_t)(ps * 8) * p.ldc + 4);
        }
    }
    quad_to_lds(wbuf, lane, a00, a01, a10, a11);
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        float v[8];
        quad_row_from_lds(wbuf, ps * 8 + rr, c8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        if (rmw) {
            v[0] += old[ps][0].x; v[1] += old[ps][0].y; v[2] += old[ps][0].z; v[3] += old[ps][0].w;
            v[4] += old[ps][1].x; v[5] += old[ps][1].y; v[6] += old[ps][1].z; v[7] += old[ps][1].w;
        }
        float* c = c0 + (size_t)(ps * 8) * p.ldc;
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

constexpr int kGroupOrderMax = 320;
struct GroupProblem {
    const void* A; const void* B; void* C;
    int M, N, K, lda, ldb, ldc, kper, tiles_m, tiles_n, vec;      // vec: C rows 16-byte aligned (vector accumulate)
    int overwrite;                                                // C = result (plain stores by the tile's one writer) instead of C +=
};
struct GroupParams {
    int count, splitk;
    float* slab; int* tickets;         // split-K through slabs (null: fp32 atomics)
    int rmw;                           // no K split and no two problems share any of C: plain vector read-modify-write, no atomics
    int tile_start[9];                 // prefix sums of the problems' 256x256 tile counts
    GroupProblem prob[8];
    // tile walk (host-made, xl_gemm_wgrad_group): entry t = problem << 12 | tm << 6 | tn of the t-th tile in launch order, laid out so
    // that the 1/8 of the tiles that lands on one XCD (one L2) is a few compact rectangles of ONE problem's tile grid -- the tiles of a
    // rectangle run side by side and share their A / B column panels in that L2.  order_n = 0: [problem][tn][tm] order.
    int order_n;
    uint16_t order[kGroupOrderMax];
};
hipError_t launch_pp_group(const GroupParams& g, int nblk, hipStream_t st);

// EPIK: -1 = generic epilogue (runtime kind, ragged edges); XL_EPI_NONE / GELU / RESIDUAL / DGELU = fast epilogue, used by
// the host for launches whose C / residual / aux rows are 16-byte aligned (interior tiles take it, edge tiles fall back)
// bn: 256 (256x256 tile) or 192 (256x192 tile: N a multiple of 192, every tile interior, fast epilogue, no fused column sums)
hipError_t launch_pp(const GemmParams& p, int a_kmajor, int b_kmajor, int epik, int bn, int nblk, hipStream_t st, int bm = 256);
// gemm_pp_pair.hip: two problems per launch (gemm_pp_kernel.h gemm_bf16_pp_pair_kernel), 256x256 tiles, A K-major, fast epilogue;
// hipErrorInvalidValue: no instance for this (layout, epilogue kind) -- forward layout: NONE / RESIDUAL / GELU_DG, dX layout:
// NONE / RESIDUAL / MULAUX.  tiles0: problem 0's tile count (linear tiles [0, tiles0) are its, the rest problem 1's)
struct PairParams { GemmParams p[2]; int tiles0; };
hipError_t launch_pp_pair(const PairParams& pp, int b_kmajor, int epik, int nblk, hipStream_t st);
bool pp_pair_has_instance(int b_kmajor, int epik);
// gemm_q.hip: 128 x 192 tiles by EIGHT waves of 32 x 96, 128 registers per wave, 80 KiB of LDS -- two workgroups per CU that each keep
// two waves per SIMD in the K loop; forward / dX layouts, fast epilogue, M % 128 == N % 192 == K % 64 == 0.  hipErrorInvalidValue: no
// instance for this (layout, epilogue kind)
hipError_t launch_q(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st);
bool q_has_instance(int b_kmajor, int epik);
// gemm_relay.hip: persistent workgroups whose two wave groups trade roles every 256 x 128 tile (K loop | LDS-DMA + previous tile's
// epilogue); forward / dX layouts, fast epilogue, M % 256 == N % 256 == K % 64 == 0, K >= 768; nblk persistent workgroups
hipError_t launch_relay(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st);
bool relay_has_instance(int b_kmajor, int epik);
// persistent variant (gemm_pp_persist.hip): A K-major, bf16 in / out, every tile interior, fast epilogue, several rounds of tiles;
// hipErrorInvalidValue when the (layout, epilogue kind) has no instance
hipError_t launch_pp_persist(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st);

}  // namespace xl
