// HBM-bound row kernels of the X-LXMERT path: LayerNorm fwd/bwd, visual-feature-encoder tail,
// embeddings, codebook gather, column sums, cross-entropy / SmoothL1 heads.
// All are "one wave (64 lanes) per row, 16-byte accesses per lane" kernels; statistics in fp32.
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "common.h"

namespace xl {

constexpr int WPB = 4;   // waves (rows) per 256-thread block

template <typename T, int NIT>
__device__ __forceinline__ void load_row(const T* row, int N, int lane, float (&v)[NIT][Elem<T>::VEC]) {
    constexpr int VEC = Elem<T>::VEC;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (it * 64 + lane) * VEC;
        if (col < N) ldvec(row + col, v[it]);
        else
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[it][i] = 0.f;
    }
}

__device__ __forceinline__ void unpack_raw(const uint4& t, float (&v)[4]) {
    v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
}
__device__ __forceinline__ void unpack_raw(const uint4& t, float (&v)[8]) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

template <int NIT, int VEC>
__device__ __forceinline__ void row_stats(const float (&v)[NIT][VEC], int N, int lane, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int i = 0; i < VEC; ++i) s += v[it][i];
    mean = wave_sum(s) / (float)N;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (it * 64 + lane) * VEC;
        if (col < N)
#pragma unroll
            for (int i = 0; i < VEC; ++i) { const float d = v[it][i] - mean; q += d * d; }
    }
    const float var = wave_sum(q) / (float)N;
    rstd = 1.0f / sqrtf(var + eps);
}

// ------------------------------------------------------------------ LayerNorm forward
template <typename T, int NIT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                     int M, int N, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[NIT][VEC];
    load_row<T, NIT>(x + (size_t)row * N, N, lane, v);
    float mean, rstd;
    row_stats<NIT, VEC>(v, N, lane, eps, mean, rstd);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (it * 64 + lane) * VEC;
        if (col < N) {
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] = (v[it][i] - mean) * rstd * gamma[col + i] + beta[col + i];
            stvec(y + (size_t)row * N + col, o);
        }
    }
}

// per-lane column partials -> block reduce over the 4 waves -> either plain stores into this block's workspace
// slot (two-stage reduction, finished by reduce_partials_kernel) or atomics straight into `out` (no workspace).
template <int NIT, int VEC, int W = WPB>
__device__ __forceinline__ void flush_colsums(float (&acc)[NIT][VEC], float* out, int N, float* red /*[W][64*VEC]*/,
                                              float* ws_slot = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        __syncthreads();
        // [i][wave][lane]: neighbouring lanes hit neighbouring banks (lane-major [wave][lane][i] was an 8-way conflict on every
        // access: 9.5 % of the LayerNorm backward's wave cycles by SQ_LDS_BANK_CONFLICT, profiles/r03e)
#pragma unroll
        for (int i = 0; i < VEC; ++i) red[(i * W + wave) * 64 + lane] = acc[it][i];
        __syncthreads();
        // every wave finishes the columns i = wave, wave + W, ... of the lanes' vectors (one wave doing all VEC of them was a
        // serial chain of VEC x W LDS reads per flush: ~4 us of the LayerNorm backward's fixed cost); same order of the W addends
        {
            const int col = (it * 64 + lane) * VEC;
            if (col < N)
#pragma unroll
                for (int i0 = 0; i0 < VEC; i0 += W) {
                    const int i = i0 + wave;
                    if (i < VEC) {
                        float s = 0.f;
#pragma unroll
                        for (int w = 0; w < W; ++w) s += red[(i * W + w) * 64 + lane];
                        if (ws_slot != nullptr) ws_slot[col + i] = s;
                        else atomicAdd(out + col + i, s);
                    }
                }
        }
    }
}

// Three accumulator sets at once (LayerNorm backward: d gamma, d beta, d bias of the dense layer in front): one pair of barriers per
// vector index instead of three -- the flush is pure latency at the end of every block (12 barrier round trips -> 4).
// red: [3][VEC][W][64] floats.  Same addends in the same order as three flush_colsums calls.
template <int NIT, int VEC, int W>
__device__ __forceinline__ void flush_colsums3(float (&a0)[NIT][VEC], float (&a1)[NIT][VEC], float (&a2)[NIT][VEC], bool with2,
                                               float* o0, float* o1, float* o2, int N, float* red, float* ws_slot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            red[((0 * VEC + i) * W + wave) * 64 + lane] = a0[it][i];
            red[((1 * VEC + i) * W + wave) * 64 + lane] = a1[it][i];
            red[((2 * VEC + i) * W + wave) * 64 + lane] = a2[it][i];
        }
        __syncthreads();
        const int col = (it * 64 + lane) * VEC;
        if (col < N)
#pragma unroll
            for (int j0 = 0; j0 < 3 * VEC; j0 += W) {            // 3 * VEC (set, column) pairs dealt to the W waves
                const int j = j0 + wave;
                if (j < 3 * VEC) {
                    const int v = j / VEC, i = j % VEC;
                    if (v < 2 || with2) {
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < W; ++w) sum += red[(j * W + w) * 64 + lane];
                        if (ws_slot != nullptr) ws_slot[(size_t)v * N + col + i] = sum;
                        else atomicAdd((v == 0 ? o0 : v == 1 ? o1 : o2) + col + i, sum);
                    }
                }
            }
    }
}

// out[v][n*stride] += sum_g ws[(g*nvec + v)*N + n], DETERMINISTIC: a block owns 32 consecutive (vector, column) pairs and sums
// the G partial slabs in 32 fixed slices (lane -> column: a wave reads two 128-byte row segments per step), the slices meet in LDS
// and are added in slice order by the thread that then makes ONE plain read-modify-write of the output element -- no atomics, the
// same bits every run.  (Rounds 1-4: grid.y slices met through up to 16 fp32 atomics per element, whose order varied from run to
// run -- the last non-reproducible reduction of the step together with the feature encoder's)  The caller
// guarantees that nothing else writes the outputs concurrently: launches on one stream are ordered, and entries of one batched
// launch never alias (xl_flush_reductions_on cuts the batch there).
constexpr int kRedSlices = 32;          // slices of the G partial slabs per output element (block = 32 columns x 32 slices)
__device__ __forceinline__ void reduce_partials_entry(const float* __restrict__ ws, int G, int nvec, int N, const ReduceOuts& outs,
                                                      int group, float (*red)[33]) {
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = group * 32 + c;
    const bool in = idx < nvec * N;
    const int v = in ? idx / N : 0, n = in ? idx - v * N : 0;
    const bool live = in && outs.p[v] != nullptr;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (live) {
        const int per = (G + kRedSlices - 1) / kRedSlices;
        const int g0 = sl * per, g1 = min(G, g0 + per);
        const float* col = ws + (size_t)v * N + n;
        const size_t pitch = (size_t)nvec * N;
        int g = g0;
        for (; g + 4 <= g1; g += 4) {
            s0 += col[(size_t)(g + 0) * pitch];
            s1 += col[(size_t)(g + 1) * pitch];
            s2 += col[(size_t)(g + 2) * pitch];
            s3 += col[(size_t)(g + 3) * pitch];
        }
        for (; g < g1; ++g) s0 += col[(size_t)g * pitch];
    }
    red[sl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && live) {
        float t = red[0][c];
#pragma unroll
        for (int k = 1; k < kRedSlices; ++k) t += red[k][c];
        float* o = outs.p[v] + (size_t)n * outs.stride[v];
        *o += t;
    }
}
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ ws, int G, int nvec, int N, ReduceOuts outs) {
    __shared__ float red[kRedSlices][33];
    reduce_partials_entry(ws, G, nvec, N, outs, blockIdx.x, red);
}

// ------------------------------------------------------------------ LayerNorm backward
constexpr int LNB_W = 8;   // waves per block of the LayerNorm backward: twice the rows in flight for the same number of partial slabs
template <typename T, int NIT, int W>
__global__ __launch_bounds__(W * 64, 4) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i, T* __restrict__ dx,
                                                     float* dgamma, float* dbeta, float* dbias_prev, int M, int N, float* ws,
                                                     T* __restrict__ dx_drop, float p_drop, float inv_keep, uint64_t seed,
                                                     const uint64_t* __restrict__ step_seed) {
    constexpr int VEC = Elem<T>::VEC;
    seed = with_step_seed(seed, step_seed);
    __shared__ float red[3 * W * 64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[NIT][VEC] = {}, ab[NIT][VEC] = {}, ax[NIT][VEC] = {};
    // gamma lives in LDS, not in 16 registers: 4 waves per SIMD instead of 3.  Layout [it][quad of 4 columns][lane][4]: a lane's
    // 16-byte read sits next to its neighbours' (lane stride 16 B, conflict-free ds_read_b128); in column order the lane stride
    // is 32 B and every read was a 2-way bank conflict -- 13 % of the kernel's wave cycles by SQ_LDS_BANK_CONFLICT (r03c)
    constexpr int QD = VEC / 4;
    __shared__ __attribute__((aligned(16))) float sgamma[NIT * 64 * VEC];
    for (int c = threadIdx.x; c < NIT * 64 * VEC; c += W * 64) {
        const int it = c / (64 * VEC), r = c % (64 * VEC), ln = r / VEC, i = r % VEC;
        sgamma[((it * QD + i / 4) * 64 + ln) * 4 + (i & 3)] = c < N ? gamma[c] : 0.f;
    }
    __syncthreads();
    for (int row = blockIdx.x * W + wave; row < M; row += gridDim.x * W) {
        // the row stays in registers as loaded (16 bytes per vector) and is unpacked in both passes: fp32 copies of x and
        // dy would cost 16 more registers, one wave per SIMD
        uint4 xr[NIT], dr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            xr[it] = dr[it] = make_uint4(0, 0, 0, 0);
            if (col < N) {
                xr[it] = *reinterpret_cast<const uint4*>(x + (size_t)row * N + col);
                dr[it] = *reinterpret_cast<const uint4*>(dy + (size_t)row * N + col);
            }
        }
        int goff = lane * 4;
        asm volatile("" : "+v"(goff));          // keep the LDS reads inside the loop (hoisted, they cost the registers back)
        const float mean = mean_i[row], rstd = rstd_i[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            if (col < N) {
                float xv[VEC], dv[VEC], gm[VEC];
                unpack_raw(xr[it], xv);
                unpack_raw(dr[it], dv);
#pragma unroll
                for (int q = 0; q < QD; ++q)          // one ds_read_b128 per four columns (lane stride 16 B: conflict-free)
                    *reinterpret_cast<float4*>(&gm[4 * q]) = *reinterpret_cast<const float4*>(__builtin_assume_aligned(&sgamma[(it * QD + q) * 256 + goff], 16));
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float xh = (xv[i] - mean) * rstd;
                    const float gd = gm[i] * dv[i];
                    s1 += gd; s2 += gd * xh;
                    ag[it][i] += dv[i] * xh;
                    ab[it][i] += dv[i];
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)N, c2 = wave_sum(s2) / (float)N;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            if (col < N) {
                float xv[VEC], dv[VEC], o[VEC], gm[VEC];
                unpack_raw(xr[it], xv);
                unpack_raw(dr[it], dv);
#pragma unroll
                for (int q = 0; q < QD; ++q)
                    *reinterpret_cast<float4*>(&gm[4 * q]) = *reinterpret_cast<const float4*>(__builtin_assume_aligned(&sgamma[(it * QD + q) * 256 + goff], 16));
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    o[i] = rstd * (gm[i] * dv[i] - c1 - (xv[i] - mean) * rstd * c2);
                stvec(dx + (size_t)row * N + col, o);
                if (dx_drop != nullptr) {          // gradient through the dropout of the dense layer feeding this LN
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        o[i] *= dropout_scale(seed, (uint32_t)row, (uint32_t)(col + i), p_drop, inv_keep);
                    stvec(dx_drop + (size_t)row * N + col, o);
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) ax[it][i] += o[i];
            }
        }
    }
    float* slot = ws ? ws + (size_t)blockIdx.x * 3 * N : nullptr;
    flush_colsums3<NIT, VEC, W>(ag, ab, ax, dbias_prev != nullptr, dgamma, dbeta, dbias_prev, N, red, slot);
}

// The same kernel with the NEXT row of a wave requested by LDS-DMA (buffer_load ... lds: global -> LDS without passing through
// registers) while the current row is processed: per row the plain kernel is a dependent chain -- two row loads, two wave
// reductions, two passes, the stores -- with ONE row's loads in flight per wave, and at four waves per SIMD the launch sits at
// ~3 TB/s (100 MB in 33 us at 16384 rows).  A software pipeline through registers costs the occupancy it needs (measured, round 3:
// 37 us); through LDS it costs none: a wave owns two 4 KiB slots [x | dy][NIT][64 lanes x 16 B] (lane-linear, so the row comes
// back with one ds_read_b128 per vector), 64 KiB per 8-wave block, two blocks per CU.  Stores and loads share vmcnt on gfx950 and
// complete out of order with each other, so the pipeline is: wait for EVERYTHING outstanding (this row's DMA and the previous
// row's stores) -> request the next row -> read this row from LDS -> reductions, both passes, stores: the next row's latency
// runs under this row's arithmetic.  bf16 rows of at most NIT x 1 KiB; same arithmetic in the same order as ln_bwd_kernel
// (bit-identical results: tests/test_hip_kernels.py).
#ifndef XL_LNB_DEBUG
#define XL_LNB_DEBUG 0
#endif
// LDS in front of gamma: the row ring [W][2 slots][2 NIT KiB] or the flush buffer [3][8][W][64] floats, whichever is larger
constexpr int lnb_dma_ring_bytes(int nit, int w) {
    return w * 2 * (2 * nit * 1024) > 3 * 8 * w * 64 * 4 ? w * 2 * (2 * nit * 1024) : 3 * 8 * w * 64 * 4;
}
template <int NIT, int W>
__global__ __launch_bounds__(W * 64, 4) void ln_bwd_dma_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean_i,
                                                         const float* __restrict__ rstd_i, bf16_t* __restrict__ dx,
                                                         float* dgamma, float* dbeta, float* dbias_prev, int M, int N, float* ws,
                                                         bf16_t* __restrict__ dx_drop, float p_drop, float inv_keep, uint64_t seed,
                                                         const uint64_t* __restrict__ step_seed) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int VEC = 8, QD = 2;
    constexpr int SLOT = 2 * NIT * 1024;                 // bytes of one row pair: [x | dy][NIT][64 x 16]
    // the ring doubles as the column-sum flush buffer `red` ([3][VEC][W][64] floats = 48 KiB at W = 8): at NIT = 1 the ring alone
    // (32 KiB) is SMALLER than that, so the region in front of gamma is the larger of the two (lnb_dma_ring_bytes, launcher)
    constexpr int RING = lnb_dma_ring_bytes(NIT, W);
    static_assert(RING >= W * 2 * SLOT && RING >= 3 * VEC * W * 64 * 4, "ring region holds the row slots and the flush buffer");
    seed = with_step_seed(seed, step_seed);
    extern __shared__ __attribute__((aligned(16))) uint8_t lnb_lds[];      // [W][2 slots][SLOT] (+ pad) | gamma;  `red` reuses the ring
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(NIT <= 2, "two 1 KiB vectors per tensor and row");
    uint8_t* ring = lnb_lds + wave * 2 * SLOT;
    const uint32_t ring_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)ring;
    float* sgamma = reinterpret_cast<float*>(lnb_lds + RING);
    float* red = reinterpret_cast<float*>(lnb_lds);
    float ag[NIT][VEC] = {}, ab[NIT][VEC] = {}, ax[NIT][VEC] = {};
    for (int c = threadIdx.x; c < NIT * 64 * VEC; c += W * 64) {
        const int it = c / (64 * VEC), r = c % (64 * VEC), ln = r / VEC, i = r % VEC;
        sgamma[((it * QD + i / 4) * 64 + ln) * 4 + (i & 3)] = c < N ? gamma[c] : 0.f;
    }
    const auto rsrc_of = [](const void* ptr, uint32_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const uint32_t bytes = (uint32_t)((size_t)M * N * 2);
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(x, bytes), rd = rsrc_of(dy, bytes);
    uint32_t lane_off[NIT];                              // byte offset of the lane's vector inside a row; out of range -> zeros
#pragma unroll
    for (int it = 0; it < NIT; ++it) lane_off[it] = (it * 64 + lane) * VEC < N ? (uint32_t)((it * 64 + lane) * VEC * 2) : 0x7FFFFFF0u;
    const auto request = [&](int row, int slot) {
        const uint32_t base = (uint32_t)row * (uint32_t)N * 2u;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const uint32_t vo = lane_off[it] == 0x7FFFFFF0u ? 0x7FFFFFF0u : base + lane_off[it];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(ring + slot * SLOT + it * 1024), 16,
                                                     (int)vo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (__attribute__((address_space(3))) void*)(ring + slot * SLOT + (NIT + it) * 1024), 16,
                                                     (int)vo, 0, 0, 0);
        }
    };
    __syncthreads();                                     // gamma is in place
    // gamma, too, is read through inline assembly (see the row reads below): [it][quad][lane][4] floats, 1 KiB per (it, quad)
    const uint32_t g_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)lnb_lds + (uint32_t)RING + (uint32_t)lane * 16u;
    const auto read_gamma = [&](int it, float (&gm)[VEC]) {
        u32x4 a, b;
        // (the reads and their wait are ONE statement: as separate asm statements hipcc moved the second read behind the wait)
        if (it == 0)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b) : "v"(g_addr) : "memory");
        else
            asm volatile("ds_read_b128 %0, %2 offset:2048\n\tds_read_b128 %1, %2 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(a), "=&v"(b) : "v"(g_addr) : "memory");
        gm[0] = __uint_as_float(a.x); gm[1] = __uint_as_float(a.y); gm[2] = __uint_as_float(a.z); gm[3] = __uint_as_float(a.w);
        gm[4] = __uint_as_float(b.x); gm[5] = __uint_as_float(b.y); gm[6] = __uint_as_float(b.z); gm[7] = __uint_as_float(b.w);
    };
    const int stride = gridDim.x * W;
    int row = blockIdx.x * W + wave, slot = 0;
    if (row < M) request(row, 0);
    for (; row < M; row += stride, slot ^= 1) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this row has landed; the previous row's stores are out
        if (row + stride < M) request(row + stride, slot ^ 1);
        // (the row comes out of LDS through inline assembly: in front of a plain LDS read hipcc drains every outstanding
        //  buffer_load ... lds -- it cannot see that the slot being read and the slot being filled differ -- which would wait
        //  for the row just requested; cf. gemm_common.h TrFrag)
        u32x4 xq[NIT], dq[NIT];
        const uint32_t rd_addr = ring_lds + (uint32_t)(slot * SLOT) + (uint32_t)lane * 16u;
#if XL_LNB_DEBUG == 3
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            uint4 a = make_uint4(0, 0, 0, 0), b = make_uint4(0, 0, 0, 0);
            if (col < N) { a = *reinterpret_cast<const uint4*>(x + (size_t)row * N + col); b = *reinterpret_cast<const uint4*>(dy + (size_t)row * N + col); }
            xq[it].x = a.x; xq[it].y = a.y; xq[it].z = a.z; xq[it].w = a.w;
            dq[it].x = b.x; dq[it].y = b.y; dq[it].z = b.z; dq[it].w = b.w;
        }
#elif XL_LNB_DEBUG == 1
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const uint4 a = *reinterpret_cast<const uint4*>(ring + slot * SLOT + it * 1024 + lane * 16);
            const uint4 b = *reinterpret_cast<const uint4*>(ring + slot * SLOT + (NIT + it) * 1024 + lane * 16);
            xq[it].x = a.x; xq[it].y = a.y; xq[it].z = a.z; xq[it].w = a.w;
            dq[it].x = b.x; dq[it].y = b.y; dq[it].z = b.z; dq[it].w = b.w;
        }
#else
#if XL_LNB_DEBUG == 2
        asm volatile("s_sleep 4" ::: "memory");
#endif
        if constexpr (NIT == 1)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(xq[0]), "=&v"(dq[0]) : "v"(rd_addr) : "memory");
        else
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(xq[0]), "=&v"(xq[NIT - 1]), "=&v"(dq[0]), "=&v"(dq[NIT - 1]) : "v"(rd_addr) : "memory");
#endif
        uint4 xr[NIT], dr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            xr[it] = make_uint4(xq[it].x, xq[it].y, xq[it].z, xq[it].w);
            dr[it] = make_uint4(dq[it].x, dq[it].y, dq[it].z, dq[it].w);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            if (col < N) {
                float xv[VEC], dv[VEC], gm[VEC];
                unpack_raw(xr[it], xv);
                unpack_raw(dr[it], dv);
                read_gamma(it, gm);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float xh = (xv[i] - mean) * rstd;
                    const float gd = gm[i] * dv[i];
                    s1 += gd; s2 += gd * xh;
                    ag[it][i] += dv[i] * xh;
                    ab[it][i] += dv[i];
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)N, c2 = wave_sum(s2) / (float)N;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            if (col < N) {
                float xv[VEC], dv[VEC], o[VEC], gm[VEC];
                unpack_raw(xr[it], xv);
                unpack_raw(dr[it], dv);
                read_gamma(it, gm);
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    o[i] = rstd * (gm[i] * dv[i] - c1 - (xv[i] - mean) * rstd * c2);
                stvec(dx + (size_t)row * N + col, o);
                if (dx_drop != nullptr) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i)
                        o[i] *= dropout_scale(seed, (uint32_t)row, (uint32_t)(col + i), p_drop, inv_keep);
                    stvec(dx_drop + (size_t)row * N + col, o);
                }
#pragma unroll
                for (int i = 0; i < VEC; ++i) ax[it][i] += o[i];
            }
        }
    }
    float* slot_ws = ws ? ws + (size_t)blockIdx.x * 3 * N : nullptr;
    flush_colsums3<NIT, VEC, W>(ag, ab, ax, dbias_prev != nullptr, dgamma, dbeta, dbias_prev, N, red, slot_ws);
}

template <int NIT>
static hipError_t launch_ln_bwd_dma(int grid, hipStream_t st, const void* dy, const void* x, const float* gamma, const float* mean,
                                    const float* rstd, void* dx, float* dgamma, float* dbeta, float* dbias_prev, int M, int N,
                                    float* workspace, void* dx_dropped, float p_drop, uint64_t seed) {
    constexpr int lds = lnb_dma_ring_bytes(NIT, LNB_W) + NIT * 64 * 8 * 4;
    auto k = ln_bwd_dma_kernel<NIT, LNB_W>;
    static bool attr = false;
    hipError_t e = hipSuccess;
    if (!attr) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(LNB_W * 64), lds, st, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (bf16_t*)dx,
                       dgamma, dbeta, dbias_prev, M, N, workspace, (bf16_t*)dx_dropped, p_drop, 1.0f / (1.0f - p_drop), seed,
                       xl::ctx().step_seed);
    return e;
}

// ------------------------------------------------------------------ visual feature encoder tail (HF:468-476)
template <typename T, int NIT>
__global__ __launch_bounds__(256) void visn_ln_fwd_kernel(const T* __restrict__ xv, const float* __restrict__ pos,
                                                          const float* __restrict__ wbox, const float* __restrict__ bbox,
                                                          const float* __restrict__ gv, const float* __restrict__ bv,
                                                          const float* __restrict__ gb, const float* __restrict__ bb,
                                                          T* __restrict__ y, float* mean_v, float* rstd_v,
                                                          float* mean_b, float* rstd_b, int M, int N, int P, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[NIT][VEC], bx[NIT][VEC];
    load_row<T, NIT>(xv + (size_t)row * N, N, lane, v);
    float pr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) pr[q] = q < P ? pos[(size_t)row * P + q] : 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (it * 64 + lane) * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float s = 0.f;
            if (col < N) {
                s = bbox[col + i];
                for (int q = 0; q < P; ++q) s = fmaf(pr[q], wbox[(size_t)(col + i) * P + q], s);
            }
            bx[it][i] = s;
        }
    }
    float mv, rv, mb, rb;
    row_stats<NIT, VEC>(v, N, lane, eps, mv, rv);
    row_stats<NIT, VEC>(bx, N, lane, eps, mb, rb);
    if (lane == 0) { mean_v[row] = mv; rstd_v[row] = rv; mean_b[row] = mb; rstd_b[row] = rb; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (it * 64 + lane) * VEC;
        if (col < N) {
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float a = (v[it][i] - mv) * rv * gv[col + i] + bv[col + i];
                const float b = (bx[it][i] - mb) * rb * gb[col + i] + bb[col + i];
                o[i] = (a + b) / 2;
            }
            stvec(y + (size_t)row * N + col, o);
        }
    }
}

template <typename T, int NIT>
__global__ __launch_bounds__(256) void visn_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ xv,
                                                          const float* __restrict__ pos, const float* __restrict__ wbox,
                                                          const float* __restrict__ bbox, const float* __restrict__ gv,
                                                          const float* __restrict__ gb, const float* __restrict__ mean_v,
                                                          const float* __restrict__ rstd_v, const float* __restrict__ mean_b,
                                                          const float* __restrict__ rstd_b, T* __restrict__ dxv,
                                                          float* dgv, float* dbv, float* dgb, float* dbb, float* dwbox,
                                                          float* dbbox, float* dbias_visn, int M, int N, int P, float* ws) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[WPB * 64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float agv[NIT][VEC] = {}, abv[NIT][VEC] = {}, agb[NIT][VEC] = {}, abx[NIT][VEC] = {}, axv[NIT][VEC] = {};
    float aw[NIT][VEC][4] = {};          // d(box_fc.weight) partials for pos dims 0..3 (P<=4 stays in registers)
    for (int row = blockIdx.x * WPB + wave; row < M; row += gridDim.x * WPB) {
        float v[NIT][VEC], d[NIT][VEC], bx[NIT][VEC];
        load_row<T, NIT>(xv + (size_t)row * N, N, lane, v);
        load_row<T, NIT>(dy + (size_t)row * N, N, lane, d);
        float pr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pr[q] = q < P ? pos[(size_t)row * P + q] : 0.f;
        const float mv = mean_v[row], rv = rstd_v[row], mb = mean_b[row], rb = rstd_b[row];
        float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                if (col < N) {
                    float s = bbox[col + i];
                    for (int q = 0; q < P; ++q) s = fmaf(pr[q], wbox[(size_t)(col + i) * P + q], s);
                    const float dh = d[it][i] * 0.5f;            // d(LN out) of either branch
                    const float xh = (v[it][i] - mv) * rv, bh = (s - mb) * rb;
                    v[it][i] = xh; bx[it][i] = bh; d[it][i] = dh;
                    const float g1 = gv[col + i] * dh, g2 = gb[col + i] * dh;
                    s1 += g1; s2 += g1 * xh; t1 += g2; t2 += g2 * bh;
                    agv[it][i] += dh * xh; abv[it][i] += dh; agb[it][i] += dh * bh;
                } else { v[it][i] = 0.f; bx[it][i] = 0.f; d[it][i] = 0.f; }
            }
        }
        const float c1 = wave_sum(s1) / (float)N, c2 = wave_sum(s2) / (float)N;
        const float e1 = wave_sum(t1) / (float)N, e2 = wave_sum(t2) / (float)N;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            if (col < N) {
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    o[i] = rv * (gv[col + i] * d[it][i] - c1 - v[it][i] * c2);
                    axv[it][i] += o[i];
                    const float dbx = rb * (gb[col + i] * d[it][i] - e1 - bx[it][i] * e2);   // d(box pre-LN)
                    abx[it][i] += dbx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) aw[it][i][q] += dbx * pr[q];
                    for (int q = 4; q < P; ++q) atomicAdd(dwbox + (size_t)(col + i) * P + q, dbx * pr[q]);
                }
                stvec(dxv + (size_t)row * N + col, o);
            }
        }
    }
    if (ws == nullptr) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            if (col < N)
#pragma unroll
                for (int i = 0; i < VEC; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q < P) atomicAdd(dwbox + (size_t)(col + i) * P + q, aw[it][i][q]);
        }
    }
    float* slot = ws ? ws + (size_t)blockIdx.x * 10 * N : nullptr;
    if (ws != nullptr) {                                  // d(box_fc.weight)[:, q] as vectors 6..9 of the slab
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float t[NIT][VEC];
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int i = 0; i < VEC; ++i) t[it][i] = aw[it][i][q];
            flush_colsums<NIT, VEC>(t, nullptr, N, red, slot + (6 + q) * N);
        }
    }
    flush_colsums<NIT, VEC>(agv, dgv, N, red, slot);
    flush_colsums<NIT, VEC>(abv, dbv, N, red, slot ? slot + N : nullptr);
    flush_colsums<NIT, VEC>(agb, dgb, N, red, slot ? slot + 2 * N : nullptr);
    flush_colsums<NIT, VEC>(abv, dbb, N, red, slot ? slot + 3 * N : nullptr);      // d(beta_box) = sum dh (same as d(beta_v))
    flush_colsums<NIT, VEC>(abx, dbbox, N, red, slot ? slot + 4 * N : nullptr);
    if (dbias_visn != nullptr) flush_colsums<NIT, VEC>(axv, dbias_visn, N, red, slot ? slot + 5 * N : nullptr);
}

// ---- the same two kernels for P <= 4 (X-LXMERT: 4 box coordinates), built around instruction count: the generic ones
// issue ~9 global loads per element for the per-column constants.  Here the constants sit in LDS as column vectors
// (16-byte reads along the 8 columns a lane owns), rows stay packed in registers, blocks loop over rows.
constexpr int VISN_W = 8;          // waves per block
template <int NC, int CAP>
struct VisnLds { float c[NC][CAP]; };
enum { VC_BBOX = 0, VC_W0 = 1, VC_GV = 5, VC_GB = 6, VC_BV = 7, VC_BB = 8 };

template <int NC, int CAP>
__device__ __forceinline__ void visn_fill(VisnLds<NC, CAP>& L, const float* wbox, const float* bbox, const float* gv,
                                          const float* gb, const float* bv, const float* bb, int N, int P, int nthreads) {
    for (int c = threadIdx.x; c < CAP; c += nthreads) {
        const bool in = c < N;
        L.c[VC_BBOX][c] = in ? bbox[c] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) L.c[VC_W0 + q][c] = (in && q < P) ? wbox[(size_t)c * P + q] : 0.f;
        L.c[VC_GV][c] = in ? gv[c] : 0.f;
        L.c[VC_GB][c] = in ? gb[c] : 0.f;
        if constexpr (NC > VC_BV) {
            L.c[VC_BV][c] = in ? bv[c] : 0.f;
            L.c[VC_BB][c] = in ? bb[c] : 0.f;
        }
    }
    __syncthreads();
}

template <int VEC>
__device__ __forceinline__ void lds_vec(const float* base, int off, float (&v)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; j += 4) {
        const float4 t = *reinterpret_cast<const float4*>(base + off + j);
        v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w;
    }
}

// box_fc pre-activation of the VEC columns at `off`: bbox + sum_q pos[q] * W[col][q], accumulated in the order q = 0..3
template <int NC, int CAP, int VEC>
__device__ __forceinline__ void visn_box(const VisnLds<NC, CAP>& L, int off, const float (&pr)[4], float (&s)[VEC]) {
    lds_vec<VEC>(L.c[VC_BBOX], off, s);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float w[VEC];
        lds_vec<VEC>(L.c[VC_W0 + q], off, w);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] = fmaf(pr[q], w[i], s[i]);
    }
}

template <typename T, int NIT>
__global__ __launch_bounds__(VISN_W * 64) void visn_ln_fwd_lds_kernel(
        const T* __restrict__ xv, const float* __restrict__ pos, const float* __restrict__ wbox, const float* __restrict__ bbox,
        const float* __restrict__ gv, const float* __restrict__ bv, const float* __restrict__ gb, const float* __restrict__ bb,
        T* __restrict__ y, float* mean_v, float* rstd_v, float* mean_b, float* rstd_b, int M, int N, int P, float eps) {
    constexpr int VEC = Elem<T>::VEC, CAP = NIT * 64 * VEC;
    __shared__ VisnLds<9, CAP> L;
    visn_fill(L, wbox, bbox, gv, gb, bv, bb, N, P, VISN_W * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * VISN_W + wave; row < M; row += gridDim.x * VISN_W) {
        float v[NIT][VEC], bx[NIT][VEC];
        load_row<T, NIT>(xv + (size_t)row * N, N, lane, v);
        float pr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pr[q] = q < P ? pos[(size_t)row * P + q] : 0.f;
        int loff = lane * VEC;
        asm volatile("" : "+v"(loff));                      // constants are re-read from LDS per row, not hoisted into registers
#pragma unroll
        for (int it = 0; it < NIT; ++it) visn_box<9, CAP, VEC>(L, it * 64 * VEC + loff, pr, bx[it]);    // 0 past N
        float mv, rv, mb, rb;
        row_stats<NIT, VEC>(v, N, lane, eps, mv, rv);
        row_stats<NIT, VEC>(bx, N, lane, eps, mb, rb);
        if (lane == 0) { mean_v[row] = mv; rstd_v[row] = rv; mean_b[row] = mb; rstd_b[row] = rb; }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC, off = it * 64 * VEC + loff;
            if (col < N) {
                float o[VEC], cgv[VEC], cbv[VEC], cgb[VEC], cbb[VEC];
                lds_vec<VEC>(L.c[VC_GV], off, cgv); lds_vec<VEC>(L.c[VC_BV], off, cbv);
                lds_vec<VEC>(L.c[VC_GB], off, cgb); lds_vec<VEC>(L.c[VC_BB], off, cbb);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float a = (v[it][i] - mv) * rv * cgv[i] + cbv[i];
                    const float b = (bx[it][i] - mb) * rb * cgb[i] + cbb[i];
                    o[i] = (a + b) / 2;
                }
                stvec(y + (size_t)row * N + col, o);
            }
        }
    }
}

template <typename T, int NIT>
__global__ __launch_bounds__(VISN_W * 64, 2) void visn_ln_bwd_lds_kernel(
        const T* __restrict__ dy, const T* __restrict__ xv, const float* __restrict__ pos, const float* __restrict__ wbox,
        const float* __restrict__ bbox, const float* __restrict__ gv, const float* __restrict__ gb,
        const float* __restrict__ mean_v, const float* __restrict__ rstd_v, const float* __restrict__ mean_b,
        const float* __restrict__ rstd_b, T* __restrict__ dxv, float* dgv, float* dbv, float* dgb, float* dbb, float* dwbox,
        float* dbbox, float* dbias_visn, int M, int N, int P, float* ws) {
    constexpr int VEC = Elem<T>::VEC, CAP = NIT * 64 * VEC;
    __shared__ VisnLds<7, CAP> L;
    visn_fill(L, wbox, bbox, gv, gb, nullptr, nullptr, N, P, VISN_W * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float agv[NIT][VEC] = {}, abv[NIT][VEC] = {}, agb[NIT][VEC] = {}, abx[NIT][VEC] = {}, axv[NIT][VEC] = {};
    // d(box_fc.weight)[:, q] partials live in LDS, one private slice per wave (index [q][it][i][lane]: conflict-free plain
    // read-modify-write).  In registers they cost 64 VGPRs and push the kernel into scratch; as ds_add_f32 on a slice shared
    // by the block they cost 230 us (LDS float atomics run a few lanes per clock).  156 KB of LDS: one block per CU.
    __shared__ float s_aw[VISN_W][4 * CAP];
    float* red = &s_aw[0][0];            // reused for the block reductions once the slices are summed
    float* my_aw = s_aw[wave];
    for (int c = lane; c < 4 * CAP; c += 64) my_aw[c] = 0.f;
    for (int row = blockIdx.x * VISN_W + wave; row < M; row += gridDim.x * VISN_W) {
        uint4 xr[NIT], dr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            xr[it] = dr[it] = make_uint4(0, 0, 0, 0);
            if (col < N) {
                xr[it] = *reinterpret_cast<const uint4*>(xv + (size_t)row * N + col);
                dr[it] = *reinterpret_cast<const uint4*>(dy + (size_t)row * N + col);
            }
        }
        float pr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pr[q] = q < P ? pos[(size_t)row * P + q] : 0.f;
        const float mv = mean_v[row], rv = rstd_v[row], mb = mean_b[row], rb = rstd_b[row];
        int loff = lane * VEC;
        asm volatile("" : "+v"(loff));
        float s1 = 0.f, s2 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC, off = it * 64 * VEC + loff;
            if (col < N) {
                float x[VEC], d[VEC], s[VEC], cgv[VEC], cgb[VEC];
                unpack_raw(xr[it], x);
                unpack_raw(dr[it], d);
                visn_box<7, CAP, VEC>(L, off, pr, s);
                lds_vec<VEC>(L.c[VC_GV], off, cgv);
                lds_vec<VEC>(L.c[VC_GB], off, cgb);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float dh = d[i] * 0.5f;                    // d(LN out) of either branch
                    const float xh = (x[i] - mv) * rv, bh = (s[i] - mb) * rb;
                    const float g1 = cgv[i] * dh, g2 = cgb[i] * dh;
                    s1 += g1; s2 += g1 * xh; t1 += g2; t2 += g2 * bh;
                    agv[it][i] += dh * xh; abv[it][i] += dh; agb[it][i] += dh * bh;
                }
            }
        }
        const float c1 = wave_sum(s1) / (float)N, c2 = wave_sum(s2) / (float)N;
        const float e1 = wave_sum(t1) / (float)N, e2 = wave_sum(t2) / (float)N;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC, off = it * 64 * VEC + loff;
            if (col < N) {
                float x[VEC], d[VEC], s[VEC], cgv[VEC], cgb[VEC], o[VEC];
                unpack_raw(xr[it], x);
                unpack_raw(dr[it], d);
                visn_box<7, CAP, VEC>(L, off, pr, s);
                lds_vec<VEC>(L.c[VC_GV], off, cgv);
                lds_vec<VEC>(L.c[VC_GB], off, cgb);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float dh = d[i] * 0.5f;
                    const float xh = (x[i] - mv) * rv, bh = (s[i] - mb) * rb;
                    o[i] = rv * (cgv[i] * dh - c1 - xh * c2);
                    axv[it][i] += o[i];
                    const float dbx = rb * (cgb[i] * dh - e1 - bh * e2);       // d(box pre-LN)
                    abx[it][i] += dbx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) my_aw[((q * NIT + it) * VEC + i) * 64 + lane] += dbx * pr[q];
                }
                stvec(dxv + (size_t)row * N + col, o);
            }
        }
    }
    float* slot = ws ? ws + (size_t)blockIdx.x * 10 * N : nullptr;
    __syncthreads();
    for (int c = threadIdx.x; c < 4 * CAP; c += VISN_W * 64) {       // vectors 6..9 of the slab, or atomics without workspace
        const int l = c & 63, i = (c >> 6) % VEC, it = (c / (64 * VEC)) % NIT, q = c / CAP;
        const int col = (it * 64 + l) * VEC + i;
        if (col < N && q < P) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < VISN_W; ++w) t += s_aw[w][c];
            if (ws != nullptr) slot[(size_t)(6 + q) * N + col] = t;
            else atomicAdd(dwbox + (size_t)col * P + q, t);
        }
    }
    flush_colsums<NIT, VEC, VISN_W>(agv, dgv, N, red, slot);
    flush_colsums<NIT, VEC, VISN_W>(abv, dbv, N, red, slot ? slot + N : nullptr);
    flush_colsums<NIT, VEC, VISN_W>(agb, dgb, N, red, slot ? slot + 2 * N : nullptr);
    flush_colsums<NIT, VEC, VISN_W>(abv, dbb, N, red, slot ? slot + 3 * N : nullptr);      // d(beta_box) = sum dh (same as d(beta_v))
    flush_colsums<NIT, VEC, VISN_W>(abx, dbbox, N, red, slot ? slot + 4 * N : nullptr);
    if (dbias_visn != nullptr) flush_colsums<NIT, VEC, VISN_W>(axv, dbias_visn, N, red, slot ? slot + 5 * N : nullptr);
}

// ------------------------------------------------------------------ embeddings
template <typename T, int NIT>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ tt,
                                                           const T* __restrict__ word, const T* __restrict__ pos,
                                                           const T* __restrict__ type, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ y,
                                                           T* __restrict__ pre, float* mean_o, float* rstd_o,
                                                           int M, int L, int N, float eps) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t id = ids[row], ty = tt ? tt[row] : 0;
    const int l = row % L;
    float v[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (it * 64 + lane) * VEC;
        if (col < N) {
            float a[VEC], b[VEC], c[VEC];
            ldvec(word + (size_t)id * N + col, a);
            ldvec(pos + (size_t)l * N + col, b);
            ldvec(type + (size_t)ty * N + col, c);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[it][i] = a[i] + b[i] + c[i];
            stvec(pre + (size_t)row * N + col, v[it]);
            // LayerNorm sees what was stored (bf16 mode: the rounded sum), so backward is consistent
            ldvec(pre + (size_t)row * N + col, v[it]);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[it][i] = 0.f;
        }
    }
    float mean, rstd;
    row_stats<NIT, VEC>(v, N, lane, eps, mean, rstd);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (it * 64 + lane) * VEC;
        if (col < N) {
            float o[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) o[i] = (v[it][i] - mean) * rstd * gamma[col + i] + beta[col + i];
            stvec(y + (size_t)row * N + col, o);
        }
    }
}

// Scatter of d(pre-LN) into the word and token-type tables, DETERMINISTIC (round 5; rounds 1-4: one fp32 atomic per element, whose
// order -- and with it the last bits of every frequent token's gradient -- varied from run to run).  Every table row has ONE writer,
// which adds the rows that map to it in row order and makes one plain read-modify-write.  16-byte accesses (a lane owns 8 / 4
// consecutive columns per 64-lane pass, NIT passes), eight rows in flight per wave.  padding_idx = 0 rows are frozen (HF:411-413).
template <typename T, int NIT>
__device__ __forceinline__ void embed_add_rows(const T* __restrict__ dpre, int N, int lane, int rr, int n,
                                               float (&acc)[NIT][Elem<T>::VEC]) {
    // acc += dpre[row of lane t], t = 0 .. n-1 (n <= 64, wave-uniform), in that order; sixteen rows requested at a time, held as
    // loaded (16 bytes per vector) until they are added
    constexpr int VEC = Elem<T>::VEC;
    constexpr int R = 16;
    for (int t0 = 0; t0 < n; t0 += R) {
        uint4 raw[R][NIT];
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (t0 + u < n) {
                const T* row = dpre + (size_t)__shfl(rr, t0 + u, 64) * N;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int col = (it * 64 + lane) * VEC;
                    raw[u][it] = col < N ? *reinterpret_cast<const uint4*>(row + col) : make_uint4(0, 0, 0, 0);
                }
            }
#pragma unroll
        for (int u = 0; u < R; ++u)
            if (t0 + u < n)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    float v[VEC];
                    unpack_raw(raw[u][it], v);
#pragma unroll
                    for (int i = 0; i < VEC; ++i) acc[it][i] += v[i];
                }
    }
}
template <typename T, int NIT>
__device__ __forceinline__ void embed_commit(float* __restrict__ table_row, int N, int lane, const float (&acc)[NIT][Elem<T>::VEC]) {
    constexpr int VEC = Elem<T>::VEC;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int col = (it * 64 + lane) * VEC;
        if (col < N)
#pragma unroll
            for (int i = 0; i < VEC; ++i) table_row[col + i] += acc[it][i];
    }
}

// without a row order from the caller: a wave per row r; the wave of the FIRST row that holds a token id owns the table row and
// finds the later occurrences by scanning the ids 64 at a time (compare + ballot); every other wave finds an earlier occurrence and
// leaves.  Correct for any caller, ~50x slower than the sorted form at B*L = 5120.
template <typename T, int NIT>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* __restrict__ dpre, const int64_t* __restrict__ ids,
                                                        float* dword, int M, int N) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t id = ids[row];
    if (id == 0) return;
    for (int r0 = 0; r0 < row; r0 += 64) {              // an earlier row with this id owns the table row
        const int r = r0 + lane;
        if (__ballot(r < row && ids[r] == id) != 0ull) return;
    }
    float acc[NIT][Elem<T>::VEC];
    load_row<T, NIT>(dpre + (size_t)row * N, N, lane, acc);
    for (int r0 = row + 1; r0 < M; r0 += 64) {
        const int r = r0 + lane;
        const bool same = r < M && ids[r] == id;
        const unsigned long long m = __ballot(same);
        if (m == 0ull) continue;
        // the matching rows, in row order, to lanes 0 .. n-1
        const int n = __builtin_popcountll(m);
        int rr = 0;
        unsigned long long mm = m;
        for (int t = 0; t < n; ++t) {
            if (lane == t) rr = r0 + __builtin_ctzll(mm);
            mm &= mm - 1ull;
        }
        embed_add_rows<T, NIT>(dpre, N, lane, rr, n, acc);
    }
    embed_commit<T, NIT>(dword + (size_t)id * N, N, lane, acc);
}

// With the rows handed over SORTED by (token id, row) -- `order`, a stable argsort of the ids computed where the ids are made (data
// loader; the engine sorts on the device otherwise): a BLOCK per sorted position; the block at the first position of an id owns that
// table row (all others leave after three loads), and its eight waves walk the id's run 32 candidates at a time, wave w the chunks
// w, w + 8, ... (one load of the order, one of the ids, a ballot; sorted: the matches are a prefix); the eight partial sums meet in
// LDS and are added in wave order -- a fixed tree over the run, the same bits every time.  A token in every sentence ([CLS], [SEP]:
// 256 occurrences) costs each wave 32 rows (two requests of 16) instead of one wave 256 dependent ones.
template <typename T, int NIT>
__global__ __launch_bounds__(512) void embed_bwd_sorted_kernel(const T* __restrict__ dpre, const int64_t* __restrict__ ids,
                                                               const int32_t* __restrict__ order, float* dword, int M, int N) {
    constexpr int VEC = Elem<T>::VEC, W = 8;
    __shared__ float part[W - 1][NIT][64][VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x;
    const int r = order[j];
    const int64_t id = ids[r];
    if (id == 0) return;
    if (j > 0 && ids[order[j - 1]] == id) return;
    float acc[NIT][VEC];
    if (wave == 0) load_row<T, NIT>(dpre + (size_t)r * N, N, lane, acc);
    else {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[it][i] = 0.f;
    }
    // wave w: the run's positions j + 1 + 32 w .. + 31, then + 32 W further on ... (32 candidates per step)
    for (int j0 = j + 1 + 32 * wave; j0 < M; j0 += 32 * W) {
        const int jj = j0 + lane;
        const bool cand = lane < 32 && jj < M;
        const int rr = cand ? order[jj] : 0;
        const unsigned long long m = __ballot(cand && ids[rr] == id) | 0xFFFFFFFF00000000ull;
        const int n = m == ~0ull ? 32 : __builtin_ctzll(~m);         // length of the run's part in this chunk
        embed_add_rows<T, NIT>(dpre, N, lane, rr, n, acc);
        if (n < 32) break;
    }
    if (wave > 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int i = 0; i < VEC; ++i) part[wave - 1][it][lane][i] = acc[it][i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < W - 1; ++w)
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[it][i] += part[w][it][lane][i];
        embed_commit<T, NIT>(dword + (size_t)id * N, N, lane, acc);
    }
}

// Token-type table (type_vocab_size rows, 2 in BERT): type t's gradient is the sum over the rows that carry it.  A block per
// (type >= 1, quarter of the column passes); its four waves take a quarter of the rows each, 64 type ids per step (a chunk without
// the type costs one load + ballot), the matching rows in row order; the quarters meet in LDS and are added in order.  Type 0 is
// frozen (padding_idx).
template <typename T, int NIT>
__global__ __launch_bounds__(256) void embed_bwd_type_kernel(const T* __restrict__ dpre, const int64_t* __restrict__ tt,
                                                             float* dtype_tab, int M, int N) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float part[4][NIT][64][VEC];
    const int t = blockIdx.x + 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = ((M + 3) / 4 + 63) / 64 * 64;
    const int r0 = wave * per, r1 = min(M, r0 + per);
    float acc[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[it][i] = 0.f;
    for (int c0 = r0; c0 < r1; c0 += 64) {
        const int r = c0 + lane;
        unsigned long long m = __ballot(r < r1 && tt[r] == t);
        while (m != 0ull) {                               // up to 8 of the chunk's matching rows at a time, in row order
            int rr = 0, n = 0;
            unsigned long long mm = m;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (mm != 0ull) { if (lane == u) rr = c0 + __builtin_ctzll(mm); mm &= mm - 1ull; ++n; }
            m = mm;
            embed_add_rows<T, NIT>(dpre, N, lane, rr, n, acc);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int i = 0; i < VEC; ++i) part[wave][it][lane][i] = acc[it][i];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int col = (it * 64 + lane) * VEC;
            if (col < N)
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    dtype_tab[(size_t)t * N + col + i] += ((part[0][it][lane][i] + part[1][it][lane][i]) + part[2][it][lane][i]) + part[3][it][lane][i];
        }
    }
}

// Position table: position l is shared by all B examples, so its gradient is a strided column sum, not a scatter.  A block per
// (position, 64 columns): its eight waves sum an eighth of the batch each (16 rows requested at a time), the partial sums meet in
// LDS and are added in wave order by the wave that makes the one read-modify-write of the table row -- deterministic (rounds 1-4:
// 8 batch slices, one atomic each).  Position 0 is frozen (padding_idx).
template <typename T>
__global__ __launch_bounds__(512) void embed_bwd_pos_kernel(const T* __restrict__ dpre, float* dpos, int B, int L, int N) {
    __shared__ float part[8][64];
    const int l = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + lane;
    if (l == 0) return;
    const int per = (B + 7) / 8;
    const int b0 = wave * per, b1 = min(B, b0 + per);
    float s = 0.f;
    if (col < N) {
        for (int b = b0; b < b1; b += 16) {
            float t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = b + u < b1 ? Elem<T>::ld(dpre + ((size_t)(b + u) * L + l) * N + col) : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) s += t[u];
        }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < N) {
        float t = part[0][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) t += part[w][lane];
        dpos[(size_t)l * N + col] += t;
    }
}

// ------------------------------------------------------------------ codebook gather + [MASK] substitution
template <typename T>
__global__ __launch_bounds__(256) void codebook_gather_kernel(const int64_t* __restrict__ cid, const uint8_t* __restrict__ mask,
                                                              const T* __restrict__ cent, const float* __restrict__ mask_feat,
                                                              T* __restrict__ feats, int M, int F) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= M) return;
    const bool masked = mask != nullptr && mask[row] != 0;
    const T* src = cent + (size_t)cid[row] * F;
    for (int col = lane * VEC; col < F; col += 64 * VEC) {
        float v[VEC];
        if (masked) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[i] = mask_feat[col + i];
        } else ldvec(src + col, v);
        stvec(feats + (size_t)row * F + col, v);
    }
}

// out[n] += sum_{m (masked)} x[m,n]; grid (col blocks, row chunks)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask,
                                                     float* out, int M, int N, int ldx, int rows_per_block, float* ws) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[WPB * 64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * VEC;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float acc[1][VEC] = {};
    if (col < N)
        for (int row = r0 + wave; row < r1; row += WPB) {
            if (mask != nullptr && mask[row] == 0) continue;
            float v[VEC];
            ldvec(x + (size_t)row * ldx + col, v);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[0][i] += v[i];
        }
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[(wave * 64 + lane) * VEC + i] = acc[0][i];
    __syncthreads();
    if (wave == 0 && col < N)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WPB; ++w) s += red[(w * 64 + lane) * VEC + i];
            if (ws != nullptr) ws[(size_t)blockIdx.y * N + col + i] = s;
            else atomicAdd(out + col + i, s);
        }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ dx, int64_t nvec) {
    constexpr int VEC = Elem<T>::VEC;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float a[VEC], b[VEC], o[VEC];
        ldvec(dy + i * VEC, a);
        ldvec(pre + i * VEC, b);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = a[j] * gelu_erf_grad(b[j]);
        stvec(dx + i * VEC, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int M, int N, int ldx, int ldy,
                                                      float p_drop, float inv_keep, uint64_t seed,
                                                      const uint64_t* __restrict__ step_seed) {
    constexpr int VEC = Elem<T>::VEC;
    seed = with_step_seed(seed, step_seed);
    const int cpr = N / VEC;                                   // chunks per row
    const int64_t total = (int64_t)M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / cpr), c = (int)(i - (int64_t)m * cpr) * VEC;
        float v[VEC];
        ldvec(x + (size_t)m * ldx + c, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] *= dropout_scale(seed, (uint32_t)m, (uint32_t)(c + j), p_drop, inv_keep);
        stvec(y + (size_t)m * ldy + c, v);
    }
}

// ------------------------------------------------------------------ head losses
__global__ __launch_bounds__(1024) void mask_counts_kernel(const int64_t* __restrict__ labels, const uint8_t* __restrict__ vis_mask,
                                                           float* counts, float* nmask, int B, int V) {
    // one 16-wave block (the result is two tiny vectors; what matters is the number of loads in flight)
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float c = 0.f;
    for (int i = threadIdx.x; i < B * V; i += 1024) c += labels[i] != -100 ? 1.f : 0.f;
    c = wave_sum(c);
    if (lane == 0) red[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        counts[0] = t;
    }
    for (int b = wave; b < B; b += 16) {
        float s = 0.f;
        for (int v = lane; v < V; v += 64) s += vis_mask[b * V + v] ? 1.f : 0.f;
        s = wave_sum(s);
        if (lane == 0) nmask[b] = s;
    }
}

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// one 256-thread block per row of fp32 logits
template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 const float* __restrict__ counts, T* __restrict__ dlogits,
                                                 float* loss_out, float* row_lse, int32_t* row_argmax, float* row_maxprob,
                                                 int M, int K, int ldl, int lddl, float grad_scale) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int row = blockIdx.x;
    const int64_t lab = labels ? labels[row] : -100;
    const bool valid = lab >= 0 && lab < K;
    const bool want_aux = row_lse != nullptr || row_argmax != nullptr || row_maxprob != nullptr;
    const float* lr = logits + (size_t)row * ldl;
    if (!valid && !want_aux) {
        if (dlogits) for (int k = threadIdx.x; k < K; k += 256) Elem<T>::st(dlogits + (size_t)row * lddl + k, 0.f);
        return;
    }
    float mx = -INFINITY; int am = 0;
    for (int k = threadIdx.x; k < K; k += 256) { const float v = lr[k]; if (v > mx) { mx = v; am = k; } }
    const float bm = block_max(mx, red);
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) s += __expf(lr[k] - bm);
    const float tot = block_sum(s, red);
    const float lse = bm + logf(tot);
    if (want_aux) {
        // argmax: smallest index among the maxima (torch.max semantics)
        int cand = (mx == bm) ? am : 0x7fffffff;
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = cand;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (row_argmax) row_argmax[row] = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
            if (row_lse) row_lse[row] = lse;
            if (row_maxprob) row_maxprob[row] = 1.0f / tot;      // exp(max - lse)
        }
    }
    if (valid) {
        const float inv_count = 1.0f / fmaxf(counts[0], 1.0f);
        if (threadIdx.x == 0 && loss_out) atomicAdd(loss_out, (lse - lr[lab]) * inv_count);
        if (dlogits) {
            const float sc = grad_scale * inv_count;
            for (int k = threadIdx.x; k < K; k += 256) {
                float g = __expf(lr[k] - lse);
                if (k == (int)lab) g -= 1.0f;
                Elem<T>::st(dlogits + (size_t)row * lddl + k, g * sc);
            }
        }
    } else if (dlogits) {
        for (int k = threadIdx.x; k < K; k += 256) Elem<T>::st(dlogits + (size_t)row * lddl + k, 0.f);
    }
}

// Same contract, row held in registers: one read of the logits (16-byte loads), 16-byte gradient stores.  Needs K % 8 == 0,
// K <= 256 * 8 * CH, ldl % 4 == 0, lddl % 8 == 0 (the 10 000-way codebook head: CH = 5, 40 logits per thread).
__device__ __forceinline__ void store8(float* p, const float (&g)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(g[0], g[1], g[2], g[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(g[4], g[5], g[6], g[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&g)[8]) { stvec(p, g); }

template <int NW>
__device__ __forceinline__ float block_max_n(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = fmaxf(r, red[w]);
    return r;
}
template <int NW>
__device__ __forceinline__ float block_sum_n(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r += red[w];
    return r;
}

// NT threads x CH chunks of 8 logits per thread (10 000-way codebook: 256 x 5; 30 522-way vocabulary: 1024 x 4).  K need not be
// a multiple of 8: the row stride is (ldl, lddl >= K rounded up to 8), slots past K count as -inf and get a zero gradient.
template <typename T, int CH, int NT>
__global__ __launch_bounds__(NT) void ce_row_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ counts, T* __restrict__ dlogits,
                                                     float* loss_out, float* row_lse, int32_t* row_argmax, float* row_maxprob,
                                                     int M, int K, int ldl, int lddl, float grad_scale) {
    constexpr int NW = NT / 64;
    __shared__ float red[NW];
    __shared__ int redi[NW];
    const bool want_aux = row_lse != nullptr || row_argmax != nullptr || row_maxprob != nullptr;
    const float inv_count = (labels && counts) ? 1.0f / fmaxf(counts[0], 1.0f) : 0.f;
    float loss_acc = 0.f;                         // thread 0: ONE loss atomic per block (per row they serialise on one address)
    for (int row = blockIdx.x; row < M; row += gridDim.x) {
    const int64_t lab = labels ? labels[row] : -100;
    const bool valid = lab >= 0 && lab < K;
    const float* lr = logits + (size_t)row * ldl;
    T* dr = dlogits ? dlogits + (size_t)row * lddl : nullptr;
    if (!valid && !want_aux) {
        if (dr) {
            const float z[8] = {};
            for (int k = threadIdx.x * 8; k < K; k += NT * 8) store8(dr + k, z);
        }
        continue;
    }
    float v[CH][8];
    float mx = -INFINITY; int am = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int k = (c * NT + threadIdx.x) * 8;
        if (k < K) {
            const float4 a = *reinterpret_cast<const float4*>(lr + k), b = *reinterpret_cast<const float4*>(lr + k + 4);
            v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w;
            v[c][4] = b.x; v[c][5] = b.y; v[c][6] = b.z; v[c][7] = b.w;
            if (k + 8 > K) {
#pragma unroll
                for (int i = 0; i < 8; ++i) if (k + i >= K) v[c][i] = -INFINITY;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = -INFINITY;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) if (v[c][i] > mx) { mx = v[c][i]; am = k + i; }
    }
    const float bm = block_max_n<NW>(mx, red);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += __expf(v[c][i] - bm);                 // exp(-inf) = 0 for the slots past K
    const float tot = block_sum_n<NW>(s, red);
    const float lse = bm + logf(tot);
    if (want_aux) {
        int cand = (mx == bm) ? am : 0x7fffffff;                                // smallest index among the maxima (torch.max)
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = cand;
        __syncthreads();
        if (threadIdx.x == 0) {
            if (row_argmax) {
                int r = redi[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) r = min(r, redi[w]);
                row_argmax[row] = r;
            }
            if (row_lse) row_lse[row] = lse;
            if (row_maxprob) row_maxprob[row] = 1.0f / tot;
        }
    }
    if (valid) {
        if (threadIdx.x == 0) loss_acc += (lse - lr[lab]) * inv_count;
        if (dr) {
            const float sc = grad_scale * inv_count;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = (c * NT + threadIdx.x) * 8;
                if (k < K) {
                    float g[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = (__expf(v[c][i] - lse) - (k + i == (int)lab ? 1.0f : 0.f)) * sc;
                    store8(dr + k, g);
                }
            }
        }
    } else if (dr) {
        const float z[8] = {};
        for (int k = threadIdx.x * 8; k < K; k += NT * 8) store8(dr + k, z);
    }
    }
    if (threadIdx.x == 0 && loss_out && loss_acc != 0.f) atomicAdd(loss_out, loss_acc);
}

template <typename T>
__global__ __launch_bounds__(256) void featloss_kernel(const T* __restrict__ pred, const T* __restrict__ cent,
                                                       const int64_t* __restrict__ cid, const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ nmask, T* __restrict__ dpred,
                                                       float* loss_out, int B, int V, int F, float grad_scale,
                                                       const int* __restrict__ rows, int n_rows, const T* __restrict__ targets) {
    constexpr int VEC = Elem<T>::VEC;
    __shared__ float red[WPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;                                             // this wave's share of the loss: ONE atomic per block at the
    for (int row = blockIdx.x * WPB + wave; row < n_rows; row += gridDim.x * WPB) {      // end (8k same-address atomics: 120 us)
        const int gr = rows ? rows[row] : row;                   // (example, grid position) the row of pred / dpred belongs to
        if (gr < 0) {                                            // padding entry of the row list: no loss, zero gradient
            if (dpred) {
                const float z[VEC] = {};
                for (int col = lane * VEC; col < F; col += 64 * VEC) stvec(dpred + (size_t)row * F + col, z);
            }
            continue;
        }
        const int b = gr / V;
        const float w = mask[gr] ? 1.0f / (fmaxf(nmask[b], 1.0f) * (float)B) : 0.f;
        // regression target: the caller's feat_labels row (ref lxrt/modeling.py:275: label_dict['feat_labels'], the real grid
        // features of lxmert_pretrain.py:177-179) or, without one, the centroid of the position's cluster id
        const T* tgt = targets != nullptr ? targets + (size_t)gr * F : cent + (size_t)cid[gr] * F;
        float s = 0.f;
        for (int col = lane * VEC; col < F; col += 64 * VEC) {
            float p[VEC], t[VEC], g[VEC];
            ldvec(pred + (size_t)row * F + col, p);
            ldvec(tgt + col, t);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float d = p[i] - t[i], ad = fabsf(d);
                s += ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
                g[i] = grad_scale * w / (float)F * fminf(fmaxf(d, -1.0f), 1.0f);
            }
            if (dpred) stvec(dpred + (size_t)row * F + col, g);
        }
        s = wave_sum(s);
        if (w != 0.f) acc += w * s / (float)F;
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss_out) {
        const float t = red[0] + red[1] + red[2] + red[3];
        if (t != 0.f) atomicAdd(loss_out, t);
    }
}

// ------------------------------------------------------------------ VQA answer head pieces
// dx = dy * (1 - y^2): backward of the pooler's tanh (HF:566-572)
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, int64_t nvec) {
    constexpr int VEC = Elem<T>::VEC;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float a[VEC], b[VEC], o[VEC];
        ldvec(dy + i * VEC, a);
        ldvec(y + i * VEC, b);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = a[j] * (1.0f - b[j] * b[j]);
        stvec(dx + i * VEC, o);
    }
}

// BCEWithLogitsLoss(reduction='mean') over [M,N] fp32 logits and soft targets (ref tasks/vqa.py:73,187) + its gradient:
//   loss += scale * sum( max(x,0) - x t + log(1 + exp(-|x|)) ),  dlogits = scale * (sigmoid(x) - t),  scale = 1/(M N).
// One block per row slice; pad columns [N, ldd) of dlogits are written as zero (they feed a contraction over ldd).
template <typename T>
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* __restrict__ x, const float* __restrict__ t, T* __restrict__ dx,
                                                         float* loss, int M, int N, int ldx, int ldt, int ldd, float scale) {
    __shared__ float red[WPB];
    const int row = blockIdx.x;
    float acc = 0.f;
    for (int n = threadIdx.x; n < ldd; n += 256) {
        float g = 0.f;
        if (n < N) {
            const float xv = x[(size_t)row * ldx + n], tv = t[(size_t)row * ldt + n];
            const float e = __expf(-fabsf(xv));
            acc += fmaxf(xv, 0.f) - xv * tv + log1pf(e);
            const float sg = xv >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
            g = (sg - tv) * scale;
        }
        if (dx != nullptr) Elem<T>::st(dx + (size_t)row * ldd + n, g);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) s += red[w];
        atomicAdd(loss, s * scale);
    }
}

// ------------------------------------------------------------------ iterative sampler pieces
// Mask-Predict re-masking (ref tasks/imggen_model.py:204-212): vis_mask[b, :] = 1 at the n_mask positions with the lowest
// confidence (ties: lower index first), 0 elsewhere.  One wave per row, V <= 64: rank by all-pairs comparison.
__global__ __launch_bounds__(256) void remask_lowest_kernel(const float* __restrict__ prob, uint8_t* __restrict__ mask,
                                                            int B, int V, int n_mask) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= B) return;
    const float mine = lane < V ? prob[(size_t)row * V + lane] : 3.0e38f;
    int rank = 0;
    for (int j = 0; j < V; ++j) {
        const float other = __shfl(mine, j, 64);
        rank += (other < mine || (other == mine && j < lane)) ? 1 : 0;
    }
    if (lane < V) mask[(size_t)row * V + lane] = rank < n_mask ? 1 : 0;
}

// code ids of the masked positions take the prediction (ref :238-243, with the code kept as an id into the codebook)
__global__ __launch_bounds__(256) void sampler_update_kernel(const int* __restrict__ pred, const uint8_t* __restrict__ mask,
                                                             int64_t* __restrict__ ids, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && mask[i]) ids[i] = pred[i];
}

// ------------------------------------------------------------------ row compaction of the masked-token head
// dst[r, :] = src[rows[r], :] (gather) / dst[rows[r], :] = src[r, :] (scatter); N a multiple of the 16-byte vector.
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(const T* __restrict__ src, const int* __restrict__ rows, T* __restrict__ dst,
                                                        int n_rows, int N, int lds, int ldd) {
    constexpr int VEC = Elem<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int g = rows[r];
    if (g < 0) {                      // padding entry of the row list: a zero row on the way in, nothing on the way out
        if (!SCATTER)
            for (int col = lane * VEC; col < N; col += 64 * VEC) *reinterpret_cast<uint4*>(dst + (size_t)r * ldd + col) = make_uint4(0, 0, 0, 0);
        return;
    }
    const T* s = src + (size_t)(SCATTER ? r : g) * lds;
    T* d = dst + (size_t)(SCATTER ? g : r) * ldd;
    for (int col = lane * VEC; col < N; col += 64 * VEC) *reinterpret_cast<uint4*>(d + col) = *reinterpret_cast<const uint4*>(s + col);
}

// out[r] = rows[r] >= 0 ? labels[rows[r]] : -100   (labels of a compacted row list; padding entries are ignored by the loss)
__global__ __launch_bounds__(256) void gather_labels_kernel(const int64_t* __restrict__ labels, const int* __restrict__ rows,
                                                            int64_t* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = rows[i] >= 0 ? labels[rows[i]] : -100;
}

// Autoregressive sampler step (ref tasks/imggen_model.py:140-153): one position per image takes its prediction and is
// un-masked -- position `fixed_pos` for every image (top-left to bottom-right / host-drawn random order), or, with
// fixed_pos < 0, each image's most confident position among those not visited yet (ties: lower index; ref topk(1)).
__global__ __launch_bounds__(256) void sampler_ar_update_kernel(const float* __restrict__ prob, const int* __restrict__ pred,
                                                                uint8_t* __restrict__ visited, uint8_t* __restrict__ mask,
                                                                int64_t* __restrict__ ids, int B, int V, int fixed_pos) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (row >= B) return;
    int pos = fixed_pos;
    if (fixed_pos < 0) {
        float best = (lane < V && !visited[(size_t)row * V + lane]) ? prob[(size_t)row * V + lane] : -10000.0f;   // ref masked_fill
        int arg = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        pos = arg;
    }
    if (lane == 0) {
        const size_t i = (size_t)row * V + pos;
        ids[i] = pred[i];
        mask[i] = 0;
        if (fixed_pos < 0) visited[i] = 1;
    }
}

}  // namespace xl

using namespace xl;

#define DISPATCH_T(dtype, ...)                                                      \
    {                                                                               \
        if ((dtype) == XL_F32) { typedef float T; __VA_ARGS__ }                     \
        else if ((dtype) == XL_BF16) { typedef bf16_t T; __VA_ARGS__ }              \
        else { xl::set_error("bad dtype %d", (int)(dtype)); return XL_ERR_BAD_DTYPE; } \
    }

// NIT = ceil(N / (64*VEC)) rounded to {1,2,4,8}
#define DISPATCH_NIT2(T, N, ...)   /* kernels that keep per-column state in LDS: rows of at most 2 x 64 vectors */ \
    {                                                                               \
        const int per__ = 64 * Elem<T>::VEC;                                        \
        if ((N) <= per__) { constexpr int NIT = 1; __VA_ARGS__ }                    \
        else { constexpr int NIT = 2; __VA_ARGS__ }                                 \
    }
#define DISPATCH_NIT(T, N, ...)                                                     \
    {                                                                               \
        const int per__ = 64 * Elem<T>::VEC;                                        \
        const int nit__ = ((N) + per__ - 1) / per__;                                \
        if (nit__ <= 1) { constexpr int NIT = 1; __VA_ARGS__ }                      \
        else if (nit__ <= 2) { constexpr int NIT = 2; __VA_ARGS__ }                 \
        else if (nit__ <= 4) { constexpr int NIT = 4; __VA_ARGS__ }                 \
        else if (nit__ <= 8) { constexpr int NIT = 8; __VA_ARGS__ }                 \
        else { xl::set_error("row length %d too large", (int)(N)); return XL_ERR_BAD_SHAPE; } \
    }

static inline int vec_of(int dtype) { return dtype == XL_F32 ? 4 : 8; }
#define CHECK_ROW(N, dtype) XL_CHECK_ARG((N) > 0 && (N) % vec_of(dtype) == 0, XL_ERR_BAD_SHAPE, \
                                         "%s: row length %d must be a multiple of %d", __func__, (int)(N), vec_of(dtype))

extern "C" int xl_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                                float* mean, float* rstd, int M, int N, float eps, int dtype, void* stream) {
    CHECK_ROW(N, dtype);
    XL_CHECK_ARG(M > 0 && x && y && gamma && beta && mean && rstd, XL_ERR_BAD_ARG, "xl_layernorm_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype, DISPATCH_NIT(T, N,
        hipLaunchKernelGGL((ln_fwd_kernel<T, NIT>), dim3((M + WPB - 1) / WPB), dim3(256), 0, st,
                           (const T*)x, gamma, beta, (T*)y, mean, rstd, M, N, eps);));
    XL_CHECK_LAUNCH();
    return XL_OK;
}

// Second stage of the two-stage column reductions.  Nothing reads a bias / LayerNorm-affine gradient before the optimizer
// (or the gradient exchange of its layer), so a caller may DEFER the second stages (xl_set_deferred_reduce) and have all of a
// layer's pending ones combined by ONE launch (xl_flush_reductions): ~110 five-microsecond launches per step become ~35.
// Each deferred producer must have been given its own workspace region.
// (switch and per-stream pending lists: the calling thread's context, common.h Ctx)
constexpr int kBatch = 6;
struct BatchArgs { int n; PendingReduce e[kBatch]; };

__global__ __launch_bounds__(1024) void reduce_partials_batched_kernel(BatchArgs a) {
    __shared__ float red[kRedSlices][33];
    const PendingReduce& e = a.e[blockIdx.z];
    if ((int)blockIdx.x * 32 >= e.nvec * e.N) return;          // (block-uniform: the grid is sized for the widest entry)
    reduce_partials_entry(e.ws, e.G, e.nvec, e.N, e.outs, blockIdx.x, red);
}

// do two pending reductions write any common output element?  (they must not share a batched launch: plain read-modify-write)
static bool reduce_outputs_alias(const PendingReduce& a, const PendingReduce& b) {
    for (int v = 0; v < a.nvec && v < 16; ++v) {
        if (a.outs.p[v] == nullptr) continue;
        const float* a0 = a.outs.p[v];
        const float* a1 = a0 + (size_t)(a.N - 1) * a.outs.stride[v] + 1;
        for (int w = 0; w < b.nvec && w < 16; ++w) {
            if (b.outs.p[w] == nullptr) continue;
            const float* b0 = b.outs.p[w];
            const float* b1 = b0 + (size_t)(b.N - 1) * b.outs.stride[w] + 1;
            if (a0 < b1 && b0 < a1) return true;
        }
    }
    return false;
}

static void launch_reduce(const float* ws, int G, int nvec, int N, ReduceOuts outs, hipStream_t st) {
    for (int v = 0; v < 16; ++v) if (outs.stride[v] == 0) outs.stride[v] = 1;
    Ctx& c = ctx();
    if (c.defer_reduce) {
        std::lock_guard<std::mutex> lk(c.mu);
        c.pending[st].push_back(PendingReduce{ws, G, nvec, N, 1, outs});
        return;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((nvec * N + 31) / 32), dim3(32 * kRedSlices), 0, st, ws, G, nvec, N, outs);
}

extern "C" int xl_set_deferred_reduce(int on) {
    ctx().defer_reduce = on ? 1 : 0;
    return XL_OK;
}

extern "C" int xl_flush_reductions(void* stream) { return xl_flush_reductions_on(stream, stream); }

extern "C" int xl_flush_reductions_on(void* producer_stream, void* launch_stream) {
    hipStream_t st = (hipStream_t)launch_stream;
    std::vector<PendingReduce> todo;
    {
        Ctx& c = ctx();
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.pending.find((hipStream_t)producer_stream);
        if (it == c.pending.end() || it->second.empty()) return XL_OK;
        todo.swap(it->second);
    }
    for (size_t i = 0; i < todo.size();) {
        BatchArgs a;
        a.n = 0;
        int gx = 1;
        // up to kBatch entries per launch, in the order they were registered; an entry that adds into an output some entry of this
        // batch already adds into (the q/k/v bias gradient of a shared cross-attention: two attention backwards, one tensor) opens
        // the next launch -- launches on a stream are ordered, so the sums are added in registration order, every run
        while (i < todo.size() && a.n < kBatch) {
            bool alias = false;
            for (int j = 0; j < a.n && !alias; ++j) alias = reduce_outputs_alias(a.e[j], todo[i]);
            if (alias) break;
            a.e[a.n++] = todo[i];
            gx = std::max(gx, (todo[i].nvec * todo[i].N + 31) / 32);
            ++i;
        }
        hipLaunchKernelGGL(reduce_partials_batched_kernel, dim3(gx, 1, a.n), dim3(32 * kRedSlices), 0, st, a);
    }
    XL_CHECK_LAUNCH();
    return XL_OK;
}

void xl::launch_colsum_reduce(const float* ws, int G, int N, float* out, hipStream_t st) {
    ReduceOuts o = {};
    o.p[0] = out;
    launch_reduce(ws, G, 1, N, o, st);
}

extern "C" int64_t xl_workspace_floats(int N) { return (int64_t)4096 * (N > 0 ? N : 1); }

extern "C" int xl_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, void* dx, float* dgamma, float* dbeta, float* dbias_prev,
                                int M, int N, float* workspace, void* dx_dropped, float p_drop, uint64_t seed,
                                int dtype, void* stream) {
    CHECK_ROW(N, dtype);
    XL_CHECK_ARG(M > 0 && dy && x && gamma && mean && rstd && dx && dgamma && dbeta, XL_ERR_BAD_ARG, "xl_layernorm_bwd: bad args");
    XL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, XL_ERR_BAD_ARG, "xl_layernorm_bwd: p_drop %f", p_drop);
    if (p_drop == 0.f) dx_dropped = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int grid = min((M + LNB_W - 1) / LNB_W, 512);        // 2 blocks x 8 waves per CU = the 4 waves per SIMD the kernel is built for
    // rows that take several passes of the grid and fit two 1 KiB vectors per wave: the variant that requests a wave's next row by
    // LDS-DMA under the current row's arithmetic (XL_LN_BWD_DMA=0: the plain kernel; XL_LN_BWD_DMA_MIN_ROWS: default 2 passes)
    static const int dma_mode = [] { const char* e = getenv("XL_LN_BWD_DMA"); return e ? atoi(e) : 1; }();
    static const int dma_min_rows = [] { const char* e = getenv("XL_LN_BWD_DMA_MIN_ROWS"); return e ? atoi(e) : 2 * 512 * LNB_W; }();
    if (dma_mode && dtype == XL_BF16 && N <= 1024 && N % 8 == 0 && M >= dma_min_rows && (double)M * N * 2 < 2.0e9 &&
        aligned16(dy) && aligned16(x)) {
        const hipError_t e = N <= 512 ? launch_ln_bwd_dma<1>(grid, st, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dbias_prev, M, N, workspace, dx_dropped, p_drop, seed)
                                      : launch_ln_bwd_dma<2>(grid, st, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dbias_prev, M, N, workspace, dx_dropped, p_drop, seed);
        XL_CHECK_ARG(e == hipSuccess, XL_ERR_HIP, "xl_layernorm_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    } else {
        DISPATCH_T(dtype, DISPATCH_NIT(T, N,
            hipLaunchKernelGGL((ln_bwd_kernel<T, NIT, LNB_W>), dim3(grid), dim3(LNB_W * 64), 0, st,
                               (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx, dgamma, dbeta, dbias_prev, M, N, workspace,
                               (T*)dx_dropped, p_drop, 1.0f / (1.0f - p_drop), seed, xl::ctx().step_seed);));
    }
    XL_CHECK_LAUNCH();
    if (workspace) {
        ReduceOuts o = {};
        o.p[0] = dgamma; o.p[1] = dbeta; o.p[2] = dbias_prev;
        launch_reduce(workspace, grid, 3, N, o, st);
        XL_CHECK_LAUNCH();
    }
    return XL_OK;
}

extern "C" int xl_visn_ln_fwd(const void* xv, const float* pos, const float* wbox, const float* bbox,
                              const float* gv, const float* bv, const float* gb, const float* bb,
                              void* y, float* mean_v, float* rstd_v, float* mean_b, float* rstd_b,
                              int M, int N, int P, float eps, int dtype, void* stream) {
    CHECK_ROW(N, dtype);
    XL_CHECK_ARG(P >= 1 && P <= 8, XL_ERR_BAD_SHAPE, "xl_visn_ln_fwd: pos dim %d not in 1..8", P);
    hipStream_t st = (hipStream_t)stream;
    if (P <= 4 && N <= 128 * vec_of(dtype)) {
        const int grid = min((M + VISN_W - 1) / VISN_W, 512);
        DISPATCH_T(dtype, DISPATCH_NIT2(T, N,
            hipLaunchKernelGGL((visn_ln_fwd_lds_kernel<T, NIT>), dim3(grid), dim3(VISN_W * 64), 0, st,
                               (const T*)xv, pos, wbox, bbox, gv, bv, gb, bb, (T*)y, mean_v, rstd_v, mean_b, rstd_b, M, N, P, eps);));
    } else {
        DISPATCH_T(dtype, DISPATCH_NIT(T, N,
            hipLaunchKernelGGL((visn_ln_fwd_kernel<T, NIT>), dim3((M + WPB - 1) / WPB), dim3(256), 0, st,
                               (const T*)xv, pos, wbox, bbox, gv, bv, gb, bb, (T*)y, mean_v, rstd_v, mean_b, rstd_b, M, N, P, eps);));
    }
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_visn_ln_bwd(const void* dy, const void* xv, const float* pos, const float* wbox, const float* bbox,
                              const float* gv, const float* gb,
                              const float* mean_v, const float* rstd_v, const float* mean_b, const float* rstd_b,
                              void* dxv, float* dgv, float* dbv, float* dgb, float* dbb,
                              float* dwbox, float* dbbox, float* dbias_visn,
                              int M, int N, int P, float* workspace, int dtype, void* stream) {
    CHECK_ROW(N, dtype);
    XL_CHECK_ARG(P >= 1 && P <= 8, XL_ERR_BAD_SHAPE, "xl_visn_ln_bwd: pos dim %d not in 1..8", P);
    hipStream_t st = (hipStream_t)stream;
    const bool lds = P <= 4 && N <= 128 * vec_of(dtype);
    const int grid = lds ? min((M + VISN_W - 1) / VISN_W, 256) : min((M + WPB - 1) / WPB, 256);
    if (lds) {
        DISPATCH_T(dtype, DISPATCH_NIT2(T, N,
            hipLaunchKernelGGL((visn_ln_bwd_lds_kernel<T, NIT>), dim3(grid), dim3(VISN_W * 64), 0, st,
                               (const T*)dy, (const T*)xv, pos, wbox, bbox, gv, gb, mean_v, rstd_v, mean_b, rstd_b,
                               (T*)dxv, dgv, dbv, dgb, dbb, dwbox, dbbox, dbias_visn, M, N, P, workspace);));
    } else {
        DISPATCH_T(dtype, DISPATCH_NIT(T, N,
            hipLaunchKernelGGL((visn_ln_bwd_kernel<T, NIT>), dim3(grid), dim3(256), 0, st,
                               (const T*)dy, (const T*)xv, pos, wbox, bbox, gv, gb, mean_v, rstd_v, mean_b, rstd_b,
                               (T*)dxv, dgv, dbv, dgb, dbb, dwbox, dbbox, dbias_visn, M, N, P, workspace);));
    }
    XL_CHECK_LAUNCH();
    if (workspace) {
        ReduceOuts o = {};
        o.p[0] = dgv; o.p[1] = dbv; o.p[2] = dgb; o.p[3] = dbb; o.p[4] = dbbox; o.p[5] = dbias_visn;
        for (int q = 0; q < 4 && q < P; ++q) { o.p[6 + q] = dwbox + q; o.stride[6 + q] = P; }
        launch_reduce(workspace, grid, 10, N, o, st);
        XL_CHECK_LAUNCH();
    }
    return XL_OK;
}

extern "C" int xl_embed_ln_fwd(const int64_t* ids, const int64_t* tt, const void* word, const void* pos,
                               const void* type, const float* gamma, const float* beta,
                               void* y, void* pre, float* mean, float* rstd,
                               int B, int L, int N, float eps, int dtype, void* stream) {
    CHECK_ROW(N, dtype);
    hipStream_t st = (hipStream_t)stream;
    const int M = B * L;
    DISPATCH_T(dtype, DISPATCH_NIT(T, N,
        hipLaunchKernelGGL((embed_ln_fwd_kernel<T, NIT>), dim3((M + WPB - 1) / WPB), dim3(256), 0, st,
                           ids, tt, (const T*)word, (const T*)pos, (const T*)type, gamma, beta, (T*)y, (T*)pre,
                           mean, rstd, M, L, N, eps);));
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_embed_bwd(const void* dpre, const int64_t* ids, const int64_t* tt, const int32_t* order,
                            float* dword, float* dpos, float* dtype_tab, int B, int L, int N, int n_types, int dtype, void* stream) {
    CHECK_ROW(N, dtype);
    XL_CHECK_ARG(dpre && ids && dword && dpos && B > 0 && L > 0 && (tt == nullptr || (dtype_tab != nullptr && n_types >= 1)), XL_ERR_BAD_ARG,
                 "xl_embed_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int M = B * L;
    XL_CHECK_ARG(N <= 64 * 16, XL_ERR_BAD_SHAPE, "xl_embed_bwd: hidden size %d > 1024", N);
    const dim3 grid((M + WPB - 1) / WPB);
#define XL_EMBED_WORD(NIT)                                                                                                           \
    if (order != nullptr) hipLaunchKernelGGL((embed_bwd_sorted_kernel<T, NIT>), dim3(M), dim3(512), 0, st, (const T*)dpre, ids, order, dword, M, N); \
    else hipLaunchKernelGGL((embed_bwd_kernel<T, NIT>), grid, dim3(256), 0, st, (const T*)dpre, ids, dword, M, N);                   \
    if (tt != nullptr && n_types > 1)                                                                                                \
        hipLaunchKernelGGL((embed_bwd_type_kernel<T, NIT>), dim3(n_types - 1), dim3(256), 0, st, (const T*)dpre, tt, dtype_tab, M, N);
    DISPATCH_T(dtype,
        const int passes = (N + 64 * vec_of(dtype) - 1) / (64 * vec_of(dtype));
        if (passes <= 1) { XL_EMBED_WORD(1) }
        else if (passes <= 2) { XL_EMBED_WORD(2) }
        else { XL_EMBED_WORD(4) }
        hipLaunchKernelGGL((embed_bwd_pos_kernel<T>), dim3(L, (N + 63) / 64), dim3(512), 0, st, (const T*)dpre, dpos, B, L, N););
#undef XL_EMBED_WORD
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_codebook_gather(const int64_t* cluster_ids, const uint8_t* vis_mask, const void* centroids,
                                  const float* mask_feat, void* feats, int M, int F, int dtype, void* stream) {
    CHECK_ROW(F, dtype);
    XL_CHECK_ARG(cluster_ids != nullptr && centroids != nullptr && feats != nullptr && M > 0 && (vis_mask == nullptr || mask_feat != nullptr),
                 XL_ERR_BAD_ARG, "xl_codebook_gather: null cluster_ids / centroids / feats (no codebook set: set_visual_embedding / "
                 "set_centroids first), or a vis_mask without mask_feat");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((codebook_gather_kernel<T>), dim3((M + WPB - 1) / WPB), dim3(256), 0, st,
                           cluster_ids, vis_mask, (const T*)centroids, mask_feat, (T*)feats, M, F););
    XL_CHECK_LAUNCH();
    return XL_OK;
}

static int colsum_impl(const void* x, const uint8_t* mask, float* out, int M, int N, int ldx, float* workspace, int dtype,
                       void* stream) {
    XL_CHECK_ARG(N % vec_of(dtype) == 0 && ldx % vec_of(dtype) == 0, XL_ERR_BAD_SHAPE,
                 "xl_colsum: N=%d / ldx=%d must be multiples of %d", N, ldx, vec_of(dtype));
    hipStream_t st = (hipStream_t)stream;
    const int per = 64 * vec_of(dtype);
    int rows_per_block = 128;
    if (workspace) while ((M + rows_per_block - 1) / rows_per_block > 128) rows_per_block *= 2;   // <= 128 partial slabs
    dim3 grid((N + per - 1) / per, (M + rows_per_block - 1) / rows_per_block);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, st, (const T*)x, mask, out, M, N, ldx, rows_per_block, workspace););
    XL_CHECK_LAUNCH();
    if (workspace) {
        ReduceOuts o = {};
        o.p[0] = out;
        launch_reduce(workspace, grid.y, 1, N, o, st);
        XL_CHECK_LAUNCH();
    }
    return XL_OK;
}
extern "C" int xl_masked_colsum(const void* x, const uint8_t* mask, float* out, int M, int N, int ldx, float* workspace,
                                int dtype, void* stream) {
    return colsum_impl(x, mask, out, M, N, ldx, workspace, dtype, stream);
}
extern "C" int xl_colsum(const void* x, float* out, int M, int N, int ldx, float* workspace, int dtype, void* stream) {
    return colsum_impl(x, nullptr, out, M, N, ldx, workspace, dtype, stream);
}

extern "C" int xl_mask_counts(const int64_t* labels, const uint8_t* vis_mask, float* counts, float* nmask,
                              int B, int V, void* stream) {
    hipLaunchKernelGGL(mask_counts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, vis_mask, counts, nmask, B, V);
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_ce_fwd_bwd(const float* logits, const int64_t* labels, const float* counts,
                             void* dlogits, float* loss_out, float* row_lse, int32_t* row_argmax, float* row_maxprob,
                             int M, int K, int ldl, int lddl, float grad_scale, int dtype, void* stream) {
    XL_CHECK_ARG(M > 0 && K > 0 && ldl >= K && logits, XL_ERR_BAD_SHAPE, "xl_ce_fwd_bwd: bad shape");
    if (labels) XL_CHECK_ARG(counts != nullptr, XL_ERR_BAD_ARG, "xl_ce_fwd_bwd: counts missing");
    hipStream_t st = (hipStream_t)stream;
    const int K8 = (K + 7) / 8 * 8;
    const bool in_regs = K8 <= 1024 * 8 * 4 && ldl >= K8 && ldl % 4 == 0 && (dlogits == nullptr || (lddl >= K8 && lddl % 8 == 0)) &&
                         ((uintptr_t)logits & 15) == 0 && ((uintptr_t)dlogits & 15) == 0;
    if (in_regs && K8 > 256 * 8 * 5) {
        DISPATCH_T(dtype,
            hipLaunchKernelGGL((ce_row_kernel<T, 4, 1024>), dim3(min(M, 1024)), dim3(1024), 0, st, logits, labels, counts, (T*)dlogits,
                               loss_out, row_lse, row_argmax, row_maxprob, M, K, ldl, lddl, grad_scale););
    } else if (in_regs && K8 > 256 * 8 * 2) {
        DISPATCH_T(dtype,
            hipLaunchKernelGGL((ce_row_kernel<T, 5, 256>), dim3(min(M, 2048)), dim3(256), 0, st, logits, labels, counts, (T*)dlogits, loss_out,
                               row_lse, row_argmax, row_maxprob, M, K, ldl, lddl, grad_scale););
    } else if (in_regs) {
        DISPATCH_T(dtype,
            hipLaunchKernelGGL((ce_row_kernel<T, 2, 256>), dim3(min(M, 2048)), dim3(256), 0, st, logits, labels, counts, (T*)dlogits, loss_out,
                               row_lse, row_argmax, row_maxprob, M, K, ldl, lddl, grad_scale););
    } else {
        DISPATCH_T(dtype,
            hipLaunchKernelGGL((ce_kernel<T>), dim3(M), dim3(256), 0, st, logits, labels, counts, (T*)dlogits, loss_out,
                               row_lse, row_argmax, row_maxprob, M, K, ldl, lddl, grad_scale););
    }
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_featloss_fwd_bwd(const void* pred, const void* centroids, const int64_t* cluster_ids,
                                   const uint8_t* vis_mask, const float* nmask, void* dpred, float* loss_out,
                                   int B, int V, int F, float grad_scale, const int* rows, int n_rows, const void* targets,
                                   int dtype, void* stream) {
    CHECK_ROW(F, dtype);
    XL_CHECK_ARG(targets != nullptr || (centroids != nullptr && cluster_ids != nullptr), XL_ERR_BAD_ARG,
                 "xl_featloss_fwd_bwd: needs either targets or centroids + cluster_ids");
    hipStream_t st = (hipStream_t)stream;
    const int M = rows ? n_rows : B * V;
    XL_CHECK_ARG(M > 0 && M <= B * V, XL_ERR_BAD_ARG, "xl_featloss_fwd_bwd: n_rows=%d", n_rows);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((featloss_kernel<T>), dim3(min((M + WPB - 1) / WPB, 1024)), dim3(256), 0, st,
                           (const T*)pred, (const T*)centroids, cluster_ids, vis_mask, nmask, (T*)dpred, loss_out, B, V, F, grad_scale,
                           rows, M, (const T*)targets););
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int dtype, void* stream) {
    XL_CHECK_ARG(dy && pre && dx && n > 0 && n % vec_of(dtype) == 0, XL_ERR_BAD_SHAPE, "xl_gelu_bwd: n=%lld must be a multiple of %d",
                 (long long)n, vec_of(dtype));
    hipStream_t st = (hipStream_t)stream;
    const int64_t nvec = n / vec_of(dtype);
    int64_t grid = (nvec + 255) / 256;
    if (grid > 4096) grid = 4096;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3((int)grid), dim3(256), 0, st, (const T*)dy, (const T*)pre, (T*)dx, nvec););
    XL_CHECK_LAUNCH();
    return XL_OK;
}

namespace xl {
__global__ __launch_bounds__(64) void wave_reduce_check_kernel(const float* __restrict__ in, float* sum_new, float* max_new, float* sum_ref) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float v = in[i];
    sum_new[i] = wave_sum(v);
    max_new[i] = wave_max(v);
    sum_ref[i] = wave_sum_shfl(v);
}
}  // namespace xl

extern "C" int xl_wave_reduce_check(const float* in, float* sum_new, float* max_new, float* sum_ref, int n_waves, void* stream) {
    XL_CHECK_ARG(in && sum_new && max_new && sum_ref && n_waves > 0, XL_ERR_BAD_ARG, "xl_wave_reduce_check: bad args");
    hipLaunchKernelGGL(xl::wave_reduce_check_kernel, dim3(n_waves), dim3(64), 0, (hipStream_t)stream, in, sum_new, max_new, sum_ref);
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_dropout(const void* x, void* y, int M, int N, int ldx, int ldy, float p_drop, uint64_t seed, int dtype,
                          void* stream) {
    CHECK_ROW(N, dtype);
    XL_CHECK_ARG(x && y && M > 0 && ldx % vec_of(dtype) == 0 && ldy % vec_of(dtype) == 0 && p_drop >= 0.f && p_drop < 1.f,
                 XL_ERR_BAD_ARG, "xl_dropout: bad args");
    hipStream_t st = (hipStream_t)stream;
    int64_t grid = ((int64_t)M * (N / vec_of(dtype)) + 255) / 256;
    if (grid > 4096) grid = 4096;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((dropout_kernel<T>), dim3((int)grid), dim3(256), 0, st, (const T*)x, (T*)y, M, N, ldx, ldy, p_drop,
                           1.0f / (1.0f - p_drop), seed, xl::ctx().step_seed););
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_tanh_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype, void* stream) {
    XL_CHECK_ARG(dy && y && dx && n > 0 && n % vec_of(dtype) == 0, XL_ERR_BAD_SHAPE, "xl_tanh_bwd: n=%lld must be a multiple of %d",
                 (long long)n, vec_of(dtype));
    hipStream_t st = (hipStream_t)stream;
    const int64_t nvec = n / vec_of(dtype);
    int64_t grid = (nvec + 255) / 256;
    if (grid > 4096) grid = 4096;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((tanh_bwd_kernel<T>), dim3((int)grid), dim3(256), 0, st, (const T*)dy, (const T*)y, (T*)dx, nvec););
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_bce_logits_fwd_bwd(const float* logits, const float* targets, void* dlogits, float* loss,
                                     int M, int N, int ld_logits, int ld_targets, int ld_dlogits, int dtype, void* stream) {
    XL_CHECK_ARG(logits && targets && loss && M > 0 && N > 0 && ld_logits >= N && ld_targets >= N &&
                 (dlogits == nullptr || ld_dlogits >= N), XL_ERR_BAD_ARG, "xl_bce_logits_fwd_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const float scale = 1.0f / ((float)M * (float)N);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((bce_logits_kernel<T>), dim3(M), dim3(256), 0, st, logits, targets, (T*)dlogits, loss, M, N,
                           ld_logits, ld_targets, dlogits ? ld_dlogits : N, scale););
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_remask_lowest(const float* prob, void* vis_mask, int B, int V, int n_mask, void* stream) {
    XL_CHECK_ARG(prob && vis_mask && B > 0 && V > 0 && V <= 64 && n_mask >= 0 && n_mask <= V, XL_ERR_BAD_ARG,
                 "xl_remask_lowest: B=%d V=%d (<= 64) n_mask=%d", B, V, n_mask);
    hipLaunchKernelGGL(remask_lowest_kernel, dim3((B + WPB - 1) / WPB), dim3(256), 0, (hipStream_t)stream, prob,
                       (uint8_t*)vis_mask, B, V, n_mask);
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_sampler_update(const int* pred_ids, const void* vis_mask, int64_t* code_ids, int n, void* stream) {
    XL_CHECK_ARG(pred_ids && vis_mask && code_ids && n > 0, XL_ERR_BAD_ARG, "xl_sampler_update: bad args");
    hipLaunchKernelGGL(sampler_update_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pred_ids,
                       (const uint8_t*)vis_mask, code_ids, n);
    XL_CHECK_LAUNCH();
    return XL_OK;
}

static int move_rows(const void* src, const int* rows, void* dst, int n_rows, int N, int ld_src, int ld_dst, int dtype, void* stream,
                     bool scatter) {
    XL_CHECK_ARG(src && rows && dst && n_rows > 0 && N > 0 && N % vec_of(dtype) == 0 && ld_src % vec_of(dtype) == 0 &&
                 ld_dst % vec_of(dtype) == 0, XL_ERR_BAD_SHAPE, "xl_gather_rows / xl_scatter_rows: N=%d n_rows=%d", N, n_rows);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((n_rows + WPB - 1) / WPB);
    if (scatter) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((move_rows_kernel<T, true>), grid, dim3(256), 0, st, (const T*)src, rows, (T*)dst,
                                             n_rows, N, ld_src, ld_dst););
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((move_rows_kernel<T, false>), grid, dim3(256), 0, st, (const T*)src, rows, (T*)dst,
                                             n_rows, N, ld_src, ld_dst););
    }
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_gather_rows(const void* src, const int* rows, void* dst, int n_rows, int N, int ld_src, int ld_dst, int dtype,
                              void* stream) {
    return move_rows(src, rows, dst, n_rows, N, ld_src, ld_dst, dtype, stream, false);
}

extern "C" int xl_scatter_rows(const void* src, const int* rows, void* dst, int n_rows, int N, int ld_src, int ld_dst, int dtype,
                               void* stream) {
    return move_rows(src, rows, dst, n_rows, N, ld_src, ld_dst, dtype, stream, true);
}

// second half of the GEMM's XL_EPI_ROWMAX epilogue over the [n_seg][M] segment records: 64 rows per block, four threads per
// row (every fourth segment each: four independent load chains per row instead of one of 160), merged through LDS
__device__ __forceinline__ void rowmax_merge(float& mx, float& se, int& idx, float omx, float ose, int oi) {
    const float nm = fmaxf(mx, omx);
    se = se * __expf(mx - nm) + ose * __expf(omx - nm);
    idx = (omx > mx || (omx == mx && oi < idx)) ? oi : idx;
    mx = nm;
}
__global__ __launch_bounds__(256) void rowmax_combine_kernel(const float4* __restrict__ ws, int n_seg, int M, float* row_maxprob,
                                                             int* row_argmax, float* row_lse) {
    __shared__ float4 part[4][64];
    const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + r;
    float mx = -INFINITY, se = 0.f;
    int idx = 0x7fffffff;
    if (m < M)
        for (int s = q; s < n_seg; s += 4) {
            const float4 rec = ws[(size_t)s * M + m];
            rowmax_merge(mx, se, idx, rec.x, rec.y, __float_as_int(rec.z));
        }
    part[q][r] = make_float4(mx, se, __int_as_float(idx), 0.f);
    __syncthreads();
    if (q != 0 || m >= M) return;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const float4 o = part[k][r];
        if (o.y > 0.f) rowmax_merge(mx, se, idx, o.x, o.y, __float_as_int(o.z));
    }
    if (row_argmax) row_argmax[m] = idx;
    if (row_maxprob) row_maxprob[m] = 1.0f / se;
    if (row_lse) row_lse[m] = mx + logf(se);
}

extern "C" int xl_rowmax_combine(const float* ws, int n_seg, int M, float* row_maxprob, int* row_argmax, float* row_lse, void* stream) {
    XL_CHECK_ARG(ws && n_seg > 0 && M > 0 && aligned16(ws), XL_ERR_BAD_ARG, "xl_rowmax_combine: bad args");
    hipLaunchKernelGGL(rowmax_combine_kernel, dim3((M + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(ws), n_seg, M, row_maxprob, row_argmax, row_lse);
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_gather_labels(const int64_t* labels, const int* rows, int64_t* out, int n_rows, void* stream) {
    XL_CHECK_ARG(labels && rows && out && n_rows > 0, XL_ERR_BAD_ARG, "xl_gather_labels: bad args (n_rows=%d)", n_rows);
    hipLaunchKernelGGL(gather_labels_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, labels, rows, out, n_rows);
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_sampler_ar_update(const float* prob, const int* pred_ids, void* visited, void* vis_mask, int64_t* code_ids,
                                    int B, int V, int fixed_pos, void* stream) {
    XL_CHECK_ARG(prob && pred_ids && vis_mask && code_ids && B > 0 && V > 0 && V <= 64 && fixed_pos < V &&
                 (fixed_pos >= 0 || visited != nullptr), XL_ERR_BAD_ARG, "xl_sampler_ar_update: B=%d V=%d fixed_pos=%d", B, V, fixed_pos);
    hipLaunchKernelGGL(sampler_ar_update_kernel, dim3((B + WPB - 1) / WPB), dim3(256), 0, (hipStream_t)stream, prob, pred_ids,
                       (uint8_t*)visited, (uint8_t*)vis_mask, code_ids, B, V, fixed_pos);
    XL_CHECK_LAUNCH();
    return XL_OK;
}
