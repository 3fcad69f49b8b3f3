// Shared device/host helpers for libxlxmert_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/xlxmert_hip.h"

namespace xl {

typedef uint16_t bf16_t;          // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;    // MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ------------------------------------------------------------------ error plumbing
void set_error(const char* fmt, ...);
#define XL_CHECK_ARG(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            xl::set_error(__VA_ARGS__);          \
            return (code);                       \
        }                                        \
    } while (0)
#define XL_CHECK_LAUNCH()                                                          \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            xl::set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,        \
                          hipGetErrorString(e__));                                 \
            return XL_ERR_HIP;                                                     \
        }                                                                          \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ------------------------------------------------------------------ bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round-to-nearest-even, through the gfx950 conversion instruction (v_cvt_pk_bf16_f32): a hand-written
// bit-trick with a NaN test compiles to a divergent branch PER ELEMENT (it was ~30 % of the attention kernels' code).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_hw_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;                 // elements per 16-byte access
    __device__ static float ld(const float* p) { return *p; }
    __device__ static void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector load/store of VEC elements as fp32 registers
__device__ __forceinline__ void ldvec(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void stvec(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void ldvec(const bf16_t* p, float (&v)[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void stvec(bf16_t* p, const float (&v)[8]) {
    uint4 t;
    t.x = pack2bf(v[0], v[1]); t.y = pack2bf(v[2], v[3]);
    t.z = pack2bf(v[4], v[5]); t.w = pack2bf(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
}

// ------------------------------------------------------------------ wave / block reductions (wave = 64)
// The xor butterfly 32, 16, 8, 4, 2, 1 -- every lane ends with the same value -- through the cross-lane data paths of the VALU
// instead of six ds_bpermute round trips through the LDS crossbar (what __shfl_xor compiles to: ~6 x 64 cycles of dependent
// latency per reduction, two or four reductions per LayerNorm row).  Same partner at every step, so the sums are bit-identical to
// the shuffle form:  32: v_permlane32_swap (x = {lo, lo}, y = {hi, hi});  16: v_permlane16_swap (even / odd rows of 16);
// 8: DPP row_ror:8 (a rotation by half a row IS xor 8);  4: DPP row_shl:4 into banks 0 and 2, row_shr:4 into banks 1 and 3;
// 2, 1: DPP quad_perm.
template <int CTRL, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf,
                                                                 BANK_MASK, false));
}
__device__ __forceinline__ float lane_xor4(float v) { return dpp_f<0x114, 0xa>(dpp_f<0x104, 0x5>(v, v), v); }   // row_shl:4 | row_shr:4
template <class Op>
__device__ __forceinline__ float wave_butterfly(float v, Op op) {
    // (the swap builtins are given two copies of v; the second goes through an empty asm so that the optimiser cannot see they are
    //  equal -- with identical operands hipcc 7.0 folds the two results into one and emits op(r0, r0))
    {
        unsigned a = __builtin_bit_cast(unsigned, v), b = a;
        asm volatile("" : "+v"(b));
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        const unsigned r0 = r[0], r1 = r[1];          // (__builtin_bit_cast applied to r[1] directly reads element 0: clang 22)
        v = op(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
    }
    {
        unsigned a = __builtin_bit_cast(unsigned, v), b = a;
        asm volatile("" : "+v"(b));
        const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        const unsigned r0 = r[0], r1 = r[1];
        v = op(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
    }
    v = op(v, dpp_f<0x128>(v, v));          // row_ror:8
    v = op(v, lane_xor4(v));
    v = op(v, dpp_f<0x4e>(v, v));           // quad_perm [2,3,0,1]
    v = op(v, dpp_f<0xb1>(v, v));           // quad_perm [1,0,3,2]
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return wave_butterfly(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_max(float v) { return wave_butterfly(v, [](float a, float b) { return fmaxf(a, b); }); }
// reference form (tests: xl_wave_reduce_check compares the two bit for bit)
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ math
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16-path variants: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, one exp + one rcp); the exponential
// exp(-x^2/2) is shared between erf(x/sqrt2) and the Gaussian pdf of the derivative.
__device__ __forceinline__ void erf_parts(float x, float& erf_v, float& e) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp); __frcp_rn is an IEEE division sequence
    e = __expf(-ax * ax);                       // = exp(-x^2/2)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float r = 1.0f - poly * t * e;
    erf_v = copysignf(r, x);
}
__device__ __forceinline__ float gelu_fast(float x) {
    float er, e;
    erf_parts(x, er, e);
    return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float er, e;
    erf_parts(x, er, e);
    return 0.5f * (1.0f + er) + x * 0.39894228040143267794f * e;
}

// Two elements at a time (v_pk_mul/fma/add_f32 are full rate on gfx950: the 13 non-transcendental operations of the
// Abramowitz-Stegun form cost half; the exp and rcp stay scalar).  Same arithmetic as erf_parts, element for element.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void erf_parts2(f32x2_t x, f32x2_t& erf_v, f32x2_t& e) {
    f32x2_t ax = x * 0.70710678118654752440f;
    ax.x = fabsf(ax.x); ax.y = fabsf(ax.y);
    const f32x2_t den = ax * 0.3275911f + 1.0f;
    f32x2_t t;
    t.x = __builtin_amdgcn_rcpf(den.x); t.y = __builtin_amdgcn_rcpf(den.y);
    // exp(-ax^2) as a bare v_exp_f32 (2^x): __expf adds a denormal-range fix-up of ~4 instructions per element, and below
    // 2^-126 the term only ever meets 1.0 (erf) or a value it cannot change (pdf term of the derivative)
    const f32x2_t nx2 = ax * ax * -1.44269504088896340736f;
    e.x = __builtin_amdgcn_exp2f(nx2.x); e.y = __builtin_amdgcn_exp2f(nx2.y);
    f32x2_t poly = t * 1.061405429f + -1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t + -0.284496736f;
    poly = poly * t + 0.254829592f;
    const f32x2_t r = 1.0f - poly * t * e;
    erf_v.x = copysignf(r.x, x.x); erf_v.y = copysignf(r.y, x.y);
}
__device__ __forceinline__ void gelu_fast8(float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2_t x = {v[i], v[i + 1]};
        f32x2_t er, e;
        erf_parts2(x, er, e);
        const f32x2_t y = x * 0.5f * (er + 1.0f);
        v[i] = y.x; v[i + 1] = y.y;
    }
}
// v[i] = gelu(v[i]), g[i] = gelu'(v[i]) from one erf / exp evaluation
__device__ __forceinline__ void gelu_fast8_dg(float (&v)[8], float (&g)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2_t x = {v[i], v[i + 1]};
        f32x2_t er, e;
        erf_parts2(x, er, e);
        const f32x2_t h = (er + 1.0f) * 0.5f;
        const f32x2_t y = x * h;
        const f32x2_t d = h + x * 0.39894228040143267794f * e;
        v[i] = y.x; v[i + 1] = y.y;
        g[i] = d.x; g[i + 1] = d.y;
    }
}
// v[i] *= gelu'(a[i])
__device__ __forceinline__ void gelu_grad_mul8(float (&v)[8], const float (&a)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2_t x = {a[i], a[i + 1]};
        f32x2_t er, e;
        erf_parts2(x, er, e);
        const f32x2_t g = (er + 1.0f) * 0.5f + x * 0.39894228040143267794f * e;
        v[i] *= g.x; v[i + 1] *= g.y;
    }
}

// ------------------------------------------------------------------ counter-based dropout
// keep-mask draw for element (row, col) under (seed): one 32-bit multiply-xorshift mix ("lowbias32" constants) of
// (row, col >> 1, seed) yields TWO 16-bit draws, for the even and the odd column of the pair; keep iff draw >= p * 2^16.
// Integer multiplies are quarter rate on CDNA, and the mask is evaluated up to three times per element of a step (GEMM
// epilogue / LayerNorm backward / attention forward + two backward passes): an 8-column row segment now costs 1 + 4 x 3
// multiplies instead of 8 x 5 (one hash per element over a 64-bit linear index with three rounds).
__device__ __forceinline__ uint32_t dropout_seed_mix(uint64_t seed) { return (uint32_t)seed ^ (uint32_t)(seed >> 32) * 0xC2B2AE3Du; }
__device__ __forceinline__ uint32_t dropout_draw16(uint32_t seedmix, uint32_t row, uint32_t col) {
    uint32_t h = row * 0x9E3779B1u ^ (col >> 1) * 0x85EBCA77u ^ seedmix;
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return (col & 1u) ? (h >> 16) : (h & 0xffffu);
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t row, uint32_t col, float p_drop) {
    const uint32_t thr = (uint32_t)(p_drop * 65536.0f);
    return dropout_draw16(dropout_seed_mix(seed), row, col) >= thr;
}
// returns 1/(1-p) if kept, 0 if dropped
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint32_t row, uint32_t col, float p_drop, float inv_keep) {
    return dropout_keep(seed, row, col, p_drop) ? inv_keep : 0.0f;
}

// Step seed in DEVICE memory (xl_set_step_seed_ptr): every dropout site's seed is  site_seed + 1000003 * *step  -- the site part
// is a launch argument, the step part is read by the kernel, so a captured launch sequence (hipGraph) draws fresh masks on
// every replay after one 8-byte update.  Null pointer: the site seed alone.
__device__ __forceinline__ uint64_t with_step_seed(uint64_t site_seed, const uint64_t* __restrict__ step) {
    return step != nullptr ? site_seed + *step * 1000003ull : site_seed;
}
// ------------------------------------------------------------------ library context (include/xlxmert_hip.h xl_ctx_*)
// Everything a caller can SET on the library lives in a context object, never in a process global: the dropout step-seed
// pointer, the deferred-reduction switch and its per-stream pending lists, the slab workspaces registered per stream, and the
// kernel-choice / debug switches.  A host object that drives the library (one trainer, one sampler engine, one nn.Module) owns
// one context and binds it to its thread before it calls (xl_ctx_bind; a recorded launch plan carries the bind as its first
// entry): two of them interleaved in one process cannot see each other's settings.  Callers that never create a context share
// the default one.
struct SlabWs { uint8_t* ptr; size_t bytes; };
struct ReduceOuts { float* p[16]; int stride[16]; };
struct PendingReduce { const float* ws; int G, nvec, N, gy; ReduceOuts outs; };
struct Ctx {
    const uint64_t* step_seed = nullptr;      // xl_set_step_seed_ptr
    int use_tr_read = 1;                      // xl_set_lds_transpose_read
    int gemm_pp = -1;                         // xl_set_gemm_pingpong; -1 = XL_GEMM_PP (default 1) at first use
    int gemm_bn192 = -1;                      // xl_set_gemm_tile192;  -1 = XL_GEMM_BN192 (default 0)
    int tail_max = -1, tail_min_k = 4096;     // xl_set_gemm_tail_split; -1 = XL_GEMM_TAIL_MAX (64) / XL_GEMM_TAIL_MIN_K (4096)
    int wgrad_slabs = -1;                     // xl_set_gemm_wgrad_slabs; -1 = XL_GEMM_WGRAD_SLABS (default 0)
    int gemm_duo = -1;                        // xl_set_gemm_duo; -1 = XL_GEMM_DUO (default 1: small launches)
    int gemm_q = -1;                // 128x192 tiles, eight 128-register waves, two workgroups per CU (xl_set_gemm_q; -1: env XL_GEMM_Q)
    int gemm_relay_wgs = -1;        // persistent workgroups of a relay launch (xl_set_gemm_relay_wgs; -1: env XL_GEMM_RELAY_WGS, default 256)
    int gemm_relay = -1;            // role-trading persistent kernel, epilogue under the next tile's K loop (xl_set_gemm_relay; -1: env XL_GEMM_RELAY)
    int gemm_pair = -1;             // two-problem launches (xl_gemm_pair / xl_set_gemm_pair; -1: env XL_GEMM_PAIR, default 1 = on)
    int gemm_split_epi = -1;        // K split of few-tile launches with an epilogue (xl_set_gemm_split_epi; -1: env XL_GEMM_SPLIT_EPI, default 0 = off)
    int gemm_persist = -1;                    // xl_set_gemm_persistent; -1 = XL_GEMM_PERSIST (default 0)
    int defer_reduce = 0;                     // xl_set_deferred_reduce
    unsigned long long* gemm_trace = nullptr; // xl_gemm_trace
    std::mutex mu;                            // guards the two maps (a context may be bound by more than one thread)
    std::unordered_map<hipStream_t, SlabWs> slab_ws;                          // xl_gemm_set_workspace
    std::unordered_map<hipStream_t, std::vector<PendingReduce>> pending;      // deferred second stages, per stream
};
Ctx& ctx();                 // the context bound to the calling thread (optim.hip)

// out[n] += sum_g ws[g*N + n] (second stage of the two-stage column reductions; rowops.hip)
void launch_colsum_reduce(const float* ws, int G, int N, float* out, hipStream_t st);

}  // namespace xl
