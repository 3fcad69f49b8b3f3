// Optimizer-side kernels (HBM-bound streams over the flat parameter buffers) and library plumbing.
//   grad-norm (ref lxmert_pretrain.py:343-353), transformers==4.1.1 AdamW (ref :110-141), dtype casts.
#include <stdarg.h>
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace xl {

static thread_local char g_err[512] = "";

// contexts: handle = index + 1 into the registry, 0 = the default context; the binding is per thread
static std::mutex g_ctx_mu;
static std::vector<Ctx*> g_ctxs;
static Ctx g_default_ctx;
static thread_local Ctx* t_ctx = nullptr;
Ctx& ctx() { return t_ctx != nullptr ? *t_ctx : g_default_ctx; }

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Sum of squares, DETERMINISTIC: every block leaves its partial in a scratch slot, the last block to arrive (ticket) adds the
// partials in index order and makes the single update of *out.  With one fp32 atomic per block (as before) the summation order --
// and with it the last bit of the gradient norm, the clip factor and every parameter -- differed from rank to rank: found by the
// two-rank test on one GPU (replicas 1 ulp apart after one step), invisible to one-rank runs and to host restatements.
constexpr int SUMSQ_MAX_BLOCKS = 512;
struct SumsqScratch { float part[SUMSQ_MAX_BLOCKS]; unsigned int ticket; };
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, float* out, int64_t n, SumsqScratch* sc) {
    __shared__ float red[4];
    __shared__ bool last;
    float s = 0.f;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four independent 16-byte loads in flight per thread
        const f32x4v a = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(g4) + i);
        const f32x4v b = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(g4) + i + stride);
        const f32x4v c = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(g4) + i + 2 * stride);
        const f32x4v d = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(g4) + i + 3 * stride);
        s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        s2 += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
        s3 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    }
    for (; i < n4; i += stride) {
        const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(g4) + i);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = (s + s1) + (s2 + s3);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&sc->part[blockIdx.x], (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int t = __hip_atomic_fetch_add(&sc->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    // fixed order: thread t sums slots t, t + 256, ...; then the same wave / block tree as above
    float a = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) a += __hip_atomic_load(&sc->part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = wave_sum(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        *out += (red[0] + red[1]) + (red[2] + red[3]);
        __hip_atomic_store(&sc->ticket, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
}

__device__ __forceinline__ void store4(float* dst, const float (&a)[4]) {
    *reinterpret_cast<float4*>(dst) = make_float4(a[0], a[1], a[2], a[3]);
}
__device__ __forceinline__ void store4(bf16_t* dst, const float (&a)[4]) {      // one 8-byte store, hardware conversion
    uint2 w;
    w.x = pack2bf(a[0], a[1]); w.y = pack2bf(a[2], a[3]);
    *reinterpret_cast<uint2*>(dst) = w;
}

// one thread = 4 consecutive parameters (n is padded to a multiple of 256 by the caller's layout)
template <typename T, int TH = 256>
__global__ __launch_bounds__(TH) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, T* __restrict__ pc, const uint8_t* __restrict__ decay,
                                                    const int* __restrict__ chunk_steps,
                                                    const float* __restrict__ sumsq, const float* __restrict__ lrs, int64_t n4,
                                                    float beta1, float beta2, float eps, float wd, float max_norm, float gscale,
                                                    int zero_grad) {
    const float lr = lrs[0], bc1 = lrs[1], bc2 = lrs[2];
    float clip = gscale;
    if (max_norm > 0.f && sumsq != nullptr) {
        const float norm = sqrtf(sumsq[0]) * gscale;
        clip *= fminf(1.0f, max_norm / (norm + 1e-6f));
    }
    const float step0 = lr * sqrtf(bc2) / bc1;
    for (int64_t i = (int64_t)blockIdx.x * TH + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TH) {
        const uint8_t fl = decay != nullptr ? decay[i >> 6] : 0;                  // 256-element chunks: bit 0 decay, bit 1 skip
        if (fl & 2) continue;              // tensor without a gradient this step: the reference's AdamW does not touch it
        float step = step0;
        if (chunk_steps != nullptr) {      // per-tensor update count (transformers AdamW keeps state["step"] per parameter)
            const float t = (float)chunk_steps[i >> 6];
            step = lr * sqrtf(1.0f - exp2f(t * log2f(beta2))) / (1.0f - exp2f(t * log2f(beta1)));
        }
        const float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        if (zero_grad && !(fl & 4)) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // the step's optimizer.zero_grad(), in
                                                                                                         // this pass (bit 2: overwritten next step)
        const float4 mv = reinterpret_cast<float4*>(m)[i];
        const float4 vv = reinterpret_cast<float4*>(v)[i];
        const bool dec = wd > 0.f && (fl & 1);
        float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
        float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = ga[j] * clip;
            ma[j] = ma[j] * beta1 + gg * (1.0f - beta1);
            va[j] = va[j] * beta2 + gg * gg * (1.0f - beta2);
            pa[j] -= step * (ma[j] / (sqrtf(va[j]) + eps));
            if (dec) pa[j] -= lr * wd * pa[j];
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
        reinterpret_cast<float4*>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
        if (pc != nullptr) store4(pc + i * 4, pa);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) Elem<T>::st(dst + i, src[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = Elem<T>::ld(src + i);
}

// one thread: advance the device-side update counter and derive the step's scalars from it (transformers
// get_linear_schedule_with_warmup + the 4.1.1 AdamW bias corrections): no host buffer a running-ahead host could overwrite
__global__ void schedule_step_kernel(int64_t* step, float base_lr, int warmup, int total, float beta1, float beta2, float* out) {
    const int64_t t = *step + 1;                       // 1-based index of the update that follows
    *step = t;
    const double done = (double)(t - 1);               // completed updates = the scheduler's step
    double f;
    if (done < (double)warmup) f = done / fmax(1.0, (double)warmup);
    else f = fmax(0.0, ((double)total - done) / fmax(1.0, (double)(total - warmup)));
    out[0] = (float)((double)base_lr * f);
    out[1] = (float)(1.0 - pow((double)beta1, (double)t));
    out[2] = (float)(1.0 - pow((double)beta2, (double)t));
    out[3] = (float)t;
}

static inline int stream_grid(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace xl

using namespace xl;

extern "C" const char* xl_last_error(void) { return g_err; }
extern "C" int xl_version(void) { return 1; }
extern "C" int xl_set_lds_transpose_read(int enable) { ctx().use_tr_read = enable ? 1 : 0; return XL_OK; }
extern "C" int xl_set_step_seed_ptr(const uint64_t* step_seed) { ctx().step_seed = step_seed; return XL_OK; }

extern "C" int64_t xl_ctx_create(void) {
    Ctx* c = new Ctx();
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_ctxs.push_back(c);
    return (int64_t)g_ctxs.size();
}

extern "C" int xl_ctx_bind(int64_t handle) {
    if (handle == 0) { t_ctx = nullptr; return XL_OK; }
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    XL_CHECK_ARG(handle >= 1 && handle <= (int64_t)g_ctxs.size() && g_ctxs[handle - 1] != nullptr, XL_ERR_BAD_ARG,
                 "xl_ctx_bind: unknown context %lld", (long long)handle);
    t_ctx = g_ctxs[handle - 1];
    return XL_OK;
}

extern "C" int xl_ctx_destroy(int64_t handle) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    XL_CHECK_ARG(handle >= 1 && handle <= (int64_t)g_ctxs.size(), XL_ERR_BAD_ARG, "xl_ctx_destroy: unknown context %lld", (long long)handle);
    Ctx* c = g_ctxs[handle - 1];
    if (c != nullptr) {
        if (t_ctx == c) t_ctx = nullptr;          // (other threads that still have it bound must rebind first: caller's contract)
        delete c;
        g_ctxs[handle - 1] = nullptr;
    }
    return XL_OK;
}

extern "C" int xl_schedule_step(int64_t* step, float base_lr, int warmup_steps, int total_steps, float beta1, float beta2,
                                float* lr_and_steps, void* stream) {
    XL_CHECK_ARG(step && lr_and_steps && warmup_steps >= 0 && total_steps > 0, XL_ERR_BAD_ARG, "xl_schedule_step: bad args");
    hipLaunchKernelGGL(schedule_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, base_lr, warmup_steps, total_steps,
                       beta1, beta2, lr_and_steps);
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int64_t xl_sumsq_scratch_bytes(void) { return (int64_t)sizeof(SumsqScratch); }

extern "C" int xl_sumsq(const float* g, float* sumsq, int64_t n, void* scratch, void* stream) {
    XL_CHECK_ARG(g && sumsq && scratch && n > 0 && aligned16(g) && aligned16(scratch), XL_ERR_BAD_ARG, "xl_sumsq: bad args");
    // few, long-lived blocks: every block ends with one ticket on a shared address (4096 of them cost more than the whole pass)
    // (202 M elements: 139 us = 5.8 TB/s with 512 blocks -- the rate of the atomic version -- 189 us with 1024, 341 us with 4096)
    const int grid = std::min(stream_grid(n >> 2), 512);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, sumsq, n, reinterpret_cast<SumsqScratch*>(scratch));
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_adamw(float* p, float* g, float* m, float* v, void* p_compute,
                        const uint8_t* decay_flags, const int* chunk_steps, const float* sumsq, const float* lr_and_steps,
                        int64_t n, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                        float grad_scale, int zero_grad, int dtype, void* stream) {
    XL_CHECK_ARG(p && g && m && v && lr_and_steps, XL_ERR_BAD_ARG, "xl_adamw: null pointer");
    XL_CHECK_ARG(n > 0 && n % 256 == 0, XL_ERR_BAD_SHAPE, "xl_adamw: n=%lld must be a positive multiple of 256", (long long)n);
    XL_CHECK_ARG(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), XL_ERR_UNALIGNED, "xl_adamw: unaligned buffer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n4 = n >> 2;
    // One persistent workgroup of 1024 threads per CU (a 128 KiB LDS request keeps it at one), grid-stride: 1134 -> 1008 us for the
    // step's 202 M parameters (6.9 GB: 6.8 TB/s) against 4096 blocks of 256 threads -- fewer, longer-lived waves keep more
    // 16-byte loads in flight per CU and leave no tail of half-empty CUs.
    constexpr int TH = 1024, LDS = 131072;
    static const int max_blocks = [] { const char* e = getenv("XL_ADAMW_BLOCKS"); const int v = e ? atoi(e) : 256; return v > 0 ? v : 256; }();
    const int grid = (int)std::min<int64_t>(max_blocks, (n4 + TH - 1) / TH);
    static bool attr_b = false, attr_f = false;
    if (dtype == XL_BF16) {
        auto k = adamw_kernel<bf16_t, TH>;
        if (!attr_b) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_b = true; }
        hipLaunchKernelGGL(k, dim3(grid), dim3(TH), LDS, st, p, g, m, v, (bf16_t*)p_compute,
                           decay_flags, chunk_steps, sumsq, lr_and_steps, n4, beta1, beta2, eps, weight_decay, max_norm, grad_scale, zero_grad);
    } else if (dtype == XL_F32) {
        auto k = adamw_kernel<float, TH>;
        if (!attr_f) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_f = true; }
        hipLaunchKernelGGL(k, dim3(grid), dim3(TH), LDS, st, p, g, m, v, (float*)p_compute,
                           decay_flags, chunk_steps, sumsq, lr_and_steps, n4, beta1, beta2, eps, weight_decay, max_norm, grad_scale, zero_grad);
    }
    else { set_error("xl_adamw: bad dtype %d", dtype); return XL_ERR_BAD_DTYPE; }
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_cast_from_f32(const float* src, void* dst, int64_t n, int dtype, void* stream) {
    XL_CHECK_ARG(src && dst && n > 0, XL_ERR_BAD_ARG, "xl_cast_from_f32: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == XL_BF16) hipLaunchKernelGGL((cast_from_f32_kernel<bf16_t>), dim3(stream_grid(n)), dim3(256), 0, st, src, (bf16_t*)dst, n);
    else if (dtype == XL_F32) hipLaunchKernelGGL((cast_from_f32_kernel<float>), dim3(stream_grid(n)), dim3(256), 0, st, src, (float*)dst, n);
    else { set_error("xl_cast_from_f32: bad dtype %d", dtype); return XL_ERR_BAD_DTYPE; }
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_cast_to_f32(const void* src, float* dst, int64_t n, int dtype, void* stream) {
    XL_CHECK_ARG(src && dst && n > 0, XL_ERR_BAD_ARG, "xl_cast_to_f32: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == XL_BF16) hipLaunchKernelGGL((cast_to_f32_kernel<bf16_t>), dim3(stream_grid(n)), dim3(256), 0, st, (const bf16_t*)src, dst, n);
    else if (dtype == XL_F32) hipLaunchKernelGGL((cast_to_f32_kernel<float>), dim3(stream_grid(n)), dim3(256), 0, st, (const float*)src, dst, n);
    else { set_error("xl_cast_to_f32: bad dtype %d", dtype); return XL_ERR_BAD_DTYPE; }
    XL_CHECK_LAUNCH();
    return XL_OK;
}

// ---------------------------------------------------------------- sparse fp32 side car of the sharded exchange (trainer gather="bf16")
// A reduce-scattered slice of the flat parameter buffer goes back to every rank as its bf16 compute copy (half the bytes of the
// fp32 master); the few elements that are READ in fp32 -- biases, LayerNorm affines, box_fc, mask_feat: ~0.1 % of a slice, scattered
// through it -- travel separately: every rank packs the ones it owns (zeros elsewhere), one small sum all-reduce makes the pack
// whole (x + 0 + ... = x), and every rank puts it back into its master buffer.
__global__ __launch_bounds__(256) void take_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int n,
                                                       int own_lo, int own_hi, float* __restrict__ dst) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int i = idx[j];
    dst[j] = (i >= own_lo && i < own_hi) ? src[i] : 0.f;
}
__global__ __launch_bounds__(256) void put_f32_kernel(float* __restrict__ dst, const int32_t* __restrict__ idx, int n,
                                                      const float* __restrict__ src) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) dst[idx[j]] = src[j];
}

extern "C" int xl_take_f32(const float* src, const int32_t* idx, int n, int own_lo, int own_hi, float* dst, void* stream) {
    XL_CHECK_ARG(src && idx && dst && n > 0, XL_ERR_BAD_ARG, "xl_take_f32: bad args");
    hipLaunchKernelGGL(take_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, idx, n, own_lo, own_hi, dst);
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_put_f32(float* dst, const int32_t* idx, int n, const float* src, void* stream) {
    XL_CHECK_ARG(src && idx && dst && n > 0, XL_ERR_BAD_ARG, "xl_put_f32: bad args");
    hipLaunchKernelGGL(put_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dst, idx, n, src);
    XL_CHECK_LAUNCH();
    return XL_OK;
}
