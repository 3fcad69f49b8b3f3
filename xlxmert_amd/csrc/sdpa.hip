// Attention core of LxmertAttention (HF:247-263) for the four (nq,nk) shapes of the path
// ({20,64} x {20,64}): one wavefront per (batch, head) problem, everything on chip.
//
//   sdpa_*_generic<T>   exact-fp32 arithmetic (XL_F32 parity path; also any dh <= 64), one lane per query row.
//   sdpa_*_mfma         bf16 operands on v_mfma_f32_32x32x16_bf16.  The scores are computed TRANSPOSED
//                       (S^T = K Q^T) so that a lane owns one query column: softmax needs a single
//                       lane^32 exchange, and the probabilities are already laid out as the B operand of
//                       O^T = V^T P^T (no cross-lane movement).  Forward: Q/K fragments (contraction along dh)
//                       come straight from global memory, V is staged in LDS and gathered transposed.  Backward:
//                       the Q, K, V, dO head tiles are staged once in LDS (coalesced 16-byte loads) and serve both
//                       the ds_read_b128 row fragments and the transposed gathers (ds_read_b64_tr_b16 or 16-bit
//                       reads).  The k-slot <-> index mapping of the MFMA is applied identically to both operands.
#include "common.h"

namespace xl {

constexpr int MAXN = 64;   // nq, nk <= 64 (8x8 grid, <=20 text tokens)

// Packed (variable-length) sequences: with q_off / k_off (int32 [B+1], ascending) the rows of batch element b on that side
// are [off[b], off[b+1]) of the matrix instead of [b*n, (b+1)*n) -- the language rows of a batch with the [PAD] positions
// removed (a third of B x 20 at sentence lengths U{6..20}).  n stays the capacity (per-example maximum, <= 64): the
// log-sum-exp buffer and the dropout counters keep their [B, H, n] indexing.  Rows [off[B], pad) -- the tail that rounds the
// packed row count up to the GEMM row tile -- are written as ZEROS by the workgroups of the last batch element, so that every
// consumer of the outputs (the out-projection, and above all the weight-gradient contractions over all rows) reads zeros there.
struct VarLen {
    const int* q_off; const int* k_off; int q_pad, k_pad;
    __device__ __forceinline__ int row0_q(int b, int nq) const { return q_off != nullptr ? q_off[b] : b * nq; }
    __device__ __forceinline__ int row0_k(int b, int nk) const { return k_off != nullptr ? k_off[b] : b * nk; }
    __device__ __forceinline__ int len_q(int b, int nq) const { return q_off != nullptr ? min(nq, q_off[b + 1] - q_off[b]) : nq; }
    __device__ __forceinline__ int len_k(int b, int nk) const { return k_off != nullptr ? min(nk, k_off[b + 1] - k_off[b]) : nk; }
};
template <typename T>
__device__ __forceinline__ void zero_pad_rows(const int* off, int pad, int b, int B, T* __restrict__ out, int ld, int c0, int ncols,
                                              int tid, int nthr) {
    if (off == nullptr || b != B - 1) return;
    const int r0 = off[B];
    constexpr int VEC = 16 / (int)sizeof(T);
    const int cpr = ncols / VEC;                       // (16-byte stores on the MFMA path; the generic kernels may step by element)
    if (ncols % VEC == 0 && ld % VEC == 0 && c0 % VEC == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
        for (int i = tid; i < (pad - r0) * cpr; i += nthr)
            *reinterpret_cast<uint4*>(out + (size_t)(r0 + i / cpr) * ld + c0 + (i % cpr) * VEC) = make_uint4(0, 0, 0, 0);
    } else {
        for (int i = tid; i < (pad - r0) * ncols; i += nthr) out[(size_t)(r0 + i / ncols) * ld + c0 + i % ncols] = (T)0;
    }
}

// Saved dropout decisions of the on-chip MFMA kernels (nq, nk <= 64): per (batch, head) problem and 32-query fragment j,
// NKF * 32 dwords in the forward kernel's REGISTER order -- dword (i*16 + r)*2 + h holds, bit t = query j*32 + t, the keep
// decisions for key i*32 + acc_row(r, h).  Dwords 2n, 2n+1 together are the 64-lane mask of accumulator register n of a wave
// whose lanes own the queries (forward, backward phase 1); a lane that owns a key (backward phase 2) finds its word at dword().
struct KeepBits {
    __host__ __device__ static constexpr size_t word0(int bh, int nqf, int nkf, int j) { return ((size_t)bh * nqf + j) * (size_t)(nkf * 32); }
    // key (0..31 inside key fragment i) -> dword index inside the fragment's 32:  key = (r&3) + 8*(r>>2) + 4*h
    __device__ static __forceinline__ int dword(int i, int key31) {
        const int h = (key31 >> 2) & 1, r = (key31 & 3) + 4 * (key31 >> 3);
        return (i * 16 + r) * 2 + h;
    }
};

// ================================================================== generic (fp32 math)
template <typename T>
__global__ __launch_bounds__(64) void sdpa_fwd_generic(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                       const uint8_t* __restrict__ key_mask, T* __restrict__ o,
                                                       float* __restrict__ lse, int H, int nq, int nk, int dh,
                                                       int ldq, int ldk, int ldv, int ldo, float scale,
                                                       float p_drop, float inv_keep, uint64_t seed,
                                                       const uint64_t* __restrict__ step_seed, VarLen vl) {
    seed = with_step_seed(seed, step_seed);
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = threadIdx.x;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    zero_pad_rows<T>(vl.q_off, vl.q_pad, b, (int)gridDim.x / H, o, ldo, h * dh, dh, threadIdx.x, 64);
    if (qi >= nq) return;
    const T* qr = q + (size_t)(q0 + qi) * ldq + h * dh;
    float qv[MAXN], s[MAXN], acc[MAXN];
    for (int d = 0; d < dh; ++d) { qv[d] = Elem<T>::ld(qr + d); acc[d] = 0.f; }
    float mx = -INFINITY;
    for (int j = 0; j < nk; ++j) {
        const T* kr = k + (size_t)(k0 + j) * ldk + h * dh;
        float d0 = 0.f;
        for (int d = 0; d < dh; ++d) d0 = fmaf(qv[d], Elem<T>::ld(kr + d), d0);
        d0 *= scale;
        if (key_mask && !key_mask[b * nk_cap + j]) d0 = -INFINITY;
        s[j] = d0;
        mx = fmaxf(mx, d0);
    }
    if (mx == -INFINITY) mx = 0.f;
    float sum = 0.f;
    for (int j = 0; j < nk; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;          // (an example without a single key: zero output, no NaN)
    for (int j = 0; j < nk; ++j) {
        float p = s[j] * inv;
        if (p_drop > 0.f) p *= dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)j, p_drop, inv_keep);
        const T* vr = v + (size_t)(k0 + j) * ldv + h * dh;
        for (int d = 0; d < dh; ++d) acc[d] = fmaf(p, Elem<T>::ld(vr + d), acc[d]);
    }
    T* orow = o + (size_t)(q0 + qi) * ldo + h * dh;
    for (int d = 0; d < dh; ++d) Elem<T>::st(orow + d, acc[d]);
    lse[(size_t)bh * nq_cap + qi] = mx + logf(sum);
}

template <typename T>
__global__ __launch_bounds__(64) void sdpa_bwd_generic(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                       const uint8_t* __restrict__ key_mask, const T* __restrict__ dout,
                                                       const float* __restrict__ lse, T* __restrict__ dq, T* __restrict__ dk,
                                                       T* __restrict__ dv, int H, int nq, int nk, int dh,
                                                       int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv,
                                                       float scale, float p_drop, float inv_keep, uint64_t seed,
                                                       const uint64_t* __restrict__ step_seed, VarLen vl) {
    seed = with_step_seed(seed, step_seed);
    __shared__ float sdk[MAXN * MAXN];
    __shared__ float sdv[MAXN * MAXN];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = threadIdx.x;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    zero_pad_rows<T>(vl.q_off, vl.q_pad, b, (int)gridDim.x / H, dq, lddq, h * dh, dh, threadIdx.x, 64);
    zero_pad_rows<T>(vl.k_off, vl.k_pad, b, (int)gridDim.x / H, dk, lddk, h * dh, dh, threadIdx.x, 64);
    zero_pad_rows<T>(vl.k_off, vl.k_pad, b, (int)gridDim.x / H, dv, lddv, h * dh, dh, threadIdx.x, 64);
    for (int i = threadIdx.x; i < nk * dh; i += 64) { sdk[i] = 0.f; sdv[i] = 0.f; }
    __syncthreads();
    if (qi < nq) {
        const T* qr = q + (size_t)(q0 + qi) * ldq + h * dh;
        const T* dor = dout + (size_t)(q0 + qi) * ldo + h * dh;
        float qv[MAXN], dov[MAXN], dqv[MAXN], p[MAXN], dp[MAXN];
        for (int d = 0; d < dh; ++d) { qv[d] = Elem<T>::ld(qr + d); dov[d] = Elem<T>::ld(dor + d); dqv[d] = 0.f; }
        const float l = lse[(size_t)bh * nq_cap + qi];
        float delta = 0.f;
        for (int j = 0; j < nk; ++j) {
            const T* kr = k + (size_t)(k0 + j) * ldk + h * dh;
            const T* vr = v + (size_t)(k0 + j) * ldv + h * dh;
            float s0 = 0.f, g = 0.f;
            for (int d = 0; d < dh; ++d) { s0 = fmaf(qv[d], Elem<T>::ld(kr + d), s0); g = fmaf(dov[d], Elem<T>::ld(vr + d), g); }
            float pj = expf(s0 * scale - l);
            if (key_mask && !key_mask[b * nk_cap + j]) pj = 0.f;
            float msk = 1.f;
            if (p_drop > 0.f) msk = dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)j, p_drop, inv_keep);
            p[j] = pj;
            dp[j] = g * msk;                 // d(p) through the dropout mask
            delta += pj * dp[j];
            const float pt = pj * msk;       // dropped probability used in O = P~ V
            for (int d = 0; d < dh; ++d) atomicAdd(&sdv[j * dh + d], pt * dov[d]);
        }
        for (int j = 0; j < nk; ++j) {
            const float ds = p[j] * (dp[j] - delta) * scale;
            const T* kr = k + (size_t)(k0 + j) * ldk + h * dh;
            for (int d = 0; d < dh; ++d) {
                dqv[d] = fmaf(ds, Elem<T>::ld(kr + d), dqv[d]);
                atomicAdd(&sdk[j * dh + d], ds * qv[d]);
            }
        }
        T* dqr = dq + (size_t)(q0 + qi) * lddq + h * dh;
        for (int d = 0; d < dh; ++d) Elem<T>::st(dqr + d, dqv[d]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nk * dh; i += 64) {
        const int j = i / dh, d = i % dh;
        Elem<T>::st(dk + (size_t)(k0 + j) * lddk + h * dh + d, sdk[i]);
        Elem<T>::st(dv + (size_t)(k0 + j) * lddv + h * dh + d, sdv[i]);
    }
}

// ================================================================== long sequences (nq or nk in 65..512)
// --max_text_length above 64 (ref param.py:140; the position table has 512 rows): not the benchmarked shapes, so these are the
// plain kernels -- one lane per query (resp. key), fp32 arithmetic, the other side streamed from global memory -- the fp32 path and a
// CORRECTNESS FALLBACK at scalar-FMA speed (bf16 launches take sdpa_*_flash below, ~100x faster; the benchmarked configuration, 20 text
// tokens, takes neither), built for any length up to MAXLONG, with the same conventions as the kernels above (dropout counters, log-sum-exp layout,
// packed rows, zeroed pad rows).  Forward: online softmax (running max / sum, the accumulator rescaled), dropout applied to the
// un-normalised terms (linear).  Backward in two launches: per query, delta_i = sum_j p_ij dp_ij and dQ_i (delta also goes to a
// caller-owned fp32 scratch [B, H, nq]); per key, dK_j and dV_j over all queries with the saved delta -- no atomics.
constexpr int MAXLONG = 512;

template <typename T>
__global__ __launch_bounds__(64) void sdpa_fwd_long(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                    const uint8_t* __restrict__ key_mask, T* __restrict__ o, float* __restrict__ lse,
                                                    int H, int nq, int nk, int dh, int ldq, int ldk, int ldv, int ldo, float scale,
                                                    float p_drop, float inv_keep, uint64_t seed, const uint64_t* __restrict__ step_seed,
                                                    VarLen vl) {
    seed = with_step_seed(seed, step_seed);
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * 64 + threadIdx.x;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    if (blockIdx.y == 0) zero_pad_rows<T>(vl.q_off, vl.q_pad, b, (int)gridDim.x / H, o, ldo, h * dh, dh, threadIdx.x, 64);
    if (qi >= nq) return;
    const T* qr = q + (size_t)(q0 + qi) * ldq + h * dh;
    float qv[MAXN], acc[MAXN];
    for (int d = 0; d < dh; ++d) { qv[d] = Elem<T>::ld(qr + d); acc[d] = 0.f; }
    float mx = -INFINITY, sum = 0.f;
    for (int j = 0; j < nk; ++j) {
        if (key_mask && !key_mask[b * nk_cap + j]) continue;
        const T* kr = k + (size_t)(k0 + j) * ldk + h * dh;
        float s0 = 0.f;
        for (int d = 0; d < dh; ++d) s0 = fmaf(qv[d], Elem<T>::ld(kr + d), s0);
        s0 *= scale;
        const float nm = fmaxf(mx, s0);
        const float corr = mx == -INFINITY ? 0.f : expf(mx - nm), e = expf(s0 - nm);
        sum = sum * corr + e;
        float pe = e;
        if (p_drop > 0.f) pe *= dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)j, p_drop, inv_keep);
        const T* vr = v + (size_t)(k0 + j) * ldv + h * dh;
        for (int d = 0; d < dh; ++d) acc[d] = fmaf(pe, Elem<T>::ld(vr + d), acc[d] * corr);
        mx = nm;
    }
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    T* orow = o + (size_t)(q0 + qi) * ldo + h * dh;
    for (int d = 0; d < dh; ++d) Elem<T>::st(orow + d, acc[d] * inv);
    lse[(size_t)bh * nq_cap + qi] = (mx == -INFINITY ? 0.f : mx) + logf(sum);
}

template <typename T>
__global__ __launch_bounds__(64) void sdpa_bwd_long_q(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                      const uint8_t* __restrict__ key_mask, const T* __restrict__ dout,
                                                      const float* __restrict__ lse, T* __restrict__ dq, float* __restrict__ delta_out,
                                                      int H, int nq, int nk, int dh, int ldq, int ldk, int ldv, int ldo, int lddq,
                                                      float scale, float p_drop, float inv_keep, uint64_t seed,
                                                      const uint64_t* __restrict__ step_seed, VarLen vl) {
    seed = with_step_seed(seed, step_seed);
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * 64 + threadIdx.x;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    if (blockIdx.y == 0) zero_pad_rows<T>(vl.q_off, vl.q_pad, b, (int)gridDim.x / H, dq, lddq, h * dh, dh, threadIdx.x, 64);
    if (qi >= nq) return;
    const T* qr = q + (size_t)(q0 + qi) * ldq + h * dh;
    const T* dor = dout + (size_t)(q0 + qi) * ldo + h * dh;
    float qv[MAXN], dov[MAXN], dqv[MAXN];
    for (int d = 0; d < dh; ++d) { qv[d] = Elem<T>::ld(qr + d); dov[d] = Elem<T>::ld(dor + d); dqv[d] = 0.f; }
    const float l = lse[(size_t)bh * nq_cap + qi];
    float delta = 0.f;
    for (int pass = 0; pass < 2; ++pass)
        for (int j = 0; j < nk; ++j) {
            if (key_mask && !key_mask[b * nk_cap + j]) continue;
            const T* kr = k + (size_t)(k0 + j) * ldk + h * dh;
            const T* vr = v + (size_t)(k0 + j) * ldv + h * dh;
            float s0 = 0.f, g = 0.f;
            for (int d = 0; d < dh; ++d) { s0 = fmaf(qv[d], Elem<T>::ld(kr + d), s0); g = fmaf(dov[d], Elem<T>::ld(vr + d), g); }
            const float pj = expf(s0 * scale - l);
            if (p_drop > 0.f) g *= dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)j, p_drop, inv_keep);
            if (pass == 0) delta += pj * g;
            else {
                const float ds = pj * (g - delta) * scale;
                for (int d = 0; d < dh; ++d) dqv[d] = fmaf(ds, Elem<T>::ld(kr + d), dqv[d]);
            }
        }
    T* dqr = dq + (size_t)(q0 + qi) * lddq + h * dh;
    for (int d = 0; d < dh; ++d) Elem<T>::st(dqr + d, dqv[d]);
    delta_out[(size_t)bh * nq_cap + qi] = delta;
}

template <typename T>
__global__ __launch_bounds__(64) void sdpa_bwd_long_k(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                      const uint8_t* __restrict__ key_mask, const T* __restrict__ dout,
                                                      const float* __restrict__ lse, const float* __restrict__ delta_in,
                                                      T* __restrict__ dk, T* __restrict__ dv, int H, int nq, int nk, int dh,
                                                      int ldq, int ldk, int ldv, int ldo, int lddk, int lddv, float scale,
                                                      float p_drop, float inv_keep, uint64_t seed, const uint64_t* __restrict__ step_seed,
                                                      VarLen vl) {
    seed = with_step_seed(seed, step_seed);
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int kj = blockIdx.y * 64 + threadIdx.x;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    if (blockIdx.y == 0) {
        zero_pad_rows<T>(vl.k_off, vl.k_pad, b, (int)gridDim.x / H, dk, lddk, h * dh, dh, threadIdx.x, 64);
        zero_pad_rows<T>(vl.k_off, vl.k_pad, b, (int)gridDim.x / H, dv, lddv, h * dh, dh, threadIdx.x, 64);
    }
    if (kj >= nk) return;
    const T* kr = k + (size_t)(k0 + kj) * ldk + h * dh;
    const T* vr = v + (size_t)(k0 + kj) * ldv + h * dh;
    float kv[MAXN], vv[MAXN], dkv[MAXN], dvv[MAXN];
    for (int d = 0; d < dh; ++d) { kv[d] = Elem<T>::ld(kr + d); vv[d] = Elem<T>::ld(vr + d); dkv[d] = 0.f; dvv[d] = 0.f; }
    const bool masked = key_mask && !key_mask[b * nk_cap + kj];
    if (!masked)
        for (int i = 0; i < nq; ++i) {
            const T* qr = q + (size_t)(q0 + i) * ldq + h * dh;
            const T* dor = dout + (size_t)(q0 + i) * ldo + h * dh;
            float s0 = 0.f, g = 0.f;
            for (int d = 0; d < dh; ++d) { s0 = fmaf(Elem<T>::ld(qr + d), kv[d], s0); g = fmaf(Elem<T>::ld(dor + d), vv[d], g); }
            const float pj = expf(s0 * scale - lse[(size_t)bh * nq_cap + i]);
            float msk = 1.f;
            if (p_drop > 0.f) msk = dropout_scale(seed, (uint32_t)(bh * nq_cap + i), (uint32_t)kj, p_drop, inv_keep);
            const float ds = pj * (g * msk - delta_in[(size_t)bh * nq_cap + i]) * scale, pt = pj * msk;
            for (int d = 0; d < dh; ++d) {
                dkv[d] = fmaf(ds, Elem<T>::ld(qr + d), dkv[d]);
                dvv[d] = fmaf(pt, Elem<T>::ld(dor + d), dvv[d]);
            }
        }
    T* dkr = dk + (size_t)(k0 + kj) * lddk + h * dh;
    T* dvr = dv + (size_t)(k0 + kj) * lddv + h * dh;
    for (int d = 0; d < dh; ++d) { Elem<T>::st(dkr + d, dkv[d]); Elem<T>::st(dvr + d, dvv[d]); }
}

// attention probabilities of one layer, recomputed from the saved log-sum-exp (LxmertModel.forward(output_attentions=True),
// HF:238-266 returns softmax(scores) AFTER its dropout): probs[b, h, q, key] fp32, dense [B, H, nq, nk] whatever the row
// packing; rows of queries / columns of keys beyond a packed example's length and masked keys are zeros.  Not on the hot path.
template <typename T>
__global__ __launch_bounds__(64) void attn_probs_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const uint8_t* __restrict__ key_mask, const float* __restrict__ lse,
                                                        float* __restrict__ probs, int H, int nq, int nk, int dh, int ldq, int ldk,
                                                        float scale, float p_drop, float inv_keep, uint64_t seed,
                                                        const uint64_t* __restrict__ step_seed, VarLen vl) {
    seed = with_step_seed(seed, step_seed);
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int qi = blockIdx.y * 64 + threadIdx.x;
    const int nq_cap = nq, nk_cap = nk;
    if (qi >= nq_cap) return;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    float* out = probs + ((size_t)bh * nq_cap + qi) * nk_cap;
    if (qi >= nq) {
        for (int j = 0; j < nk_cap; ++j) out[j] = 0.f;
        return;
    }
    const T* qr = q + (size_t)(q0 + qi) * ldq + h * dh;
    float qv[MAXN];
    for (int d = 0; d < dh; ++d) qv[d] = Elem<T>::ld(qr + d);
    const float l = lse[(size_t)bh * nq_cap + qi];
    for (int j = 0; j < nk_cap; ++j) {
        float p = 0.f;
        if (j < nk && (key_mask == nullptr || key_mask[b * nk_cap + j] != 0)) {
            const T* kr = k + (size_t)(k0 + j) * ldk + h * dh;
            float s0 = 0.f;
            for (int d = 0; d < dh; ++d) s0 = fmaf(qv[d], Elem<T>::ld(kr + d), s0);
            p = expf(s0 * scale - l);
            if (p_drop > 0.f) p *= dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)j, p_drop, inv_keep);
        }
        out[j] = p;
    }
}

// ================================================================== bf16 MFMA kernels
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4bf16s_t;
typedef __attribute__((ext_vector_type(8))) __bf16 v8bf16s_t;

__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf16s_t, a), __builtin_bit_cast(v8bf16s_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
// accumulator row of register r for lane-half hi:  (r&3) + 8*(r>>2) + 4*hi
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// fragment with the contraction along dh, straight from global: row `row` (zero if >= nrows), 8 elems at col
__device__ __forceinline__ bf16x8_t gfrag(const bf16_t* __restrict__ base, int ld, int row, int nrows, int col) {
    bf16x8_t f = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nrows) f = *reinterpret_cast<const bf16x8_t*>(base + (size_t)row * ld + col);
    return f;
}

// key mask of one batch element as a 64-bit word (bit = key attends): ONE byte load per lane + a ballot, instead of a
// dependent byte load per accumulator register (32 serialised round trips per wave and phase)
__device__ __forceinline__ uint64_t key_bits(const uint8_t* __restrict__ key_mask, int base, int nk_cap, int lane) {
    if (key_mask == nullptr) return ~0ull;
    const bool on = lane < nk_cap && key_mask[base + lane] != 0;
    return __ballot(on);
}

// (the transposed-operand gathers and the K-major fragments below both read tiles staged ONCE per problem with coalesced
//  16-byte loads: per-lane strided global fragment loads touch 64 cache lines per instruction and were ~6x slower)
// SW (backward kernel, DH = 64): no row padding -- 128-byte rows whose 16-byte chunk c sits at chunk c ^ ((row >> 1) & 7), the
// layout of the GEMM's K-major operand tile: ds_read_b128 row fragments are conflict-free without the 16 pad bytes per row,
// and three 8 KiB tiles per (batch, head) problem let six workgroups share a CU instead of four.
template <int DH, bool SW = false> struct Tile {
    static_assert(!SW || DH == 64, "swizzled tile: 128-byte rows");
    static constexpr int DHP = DH < 32 ? 32 : DH;       // padded so that every MFMA row d<32 exists (zeros)
    static constexpr int PITCH = SW ? DHP * 2 : DHP * 2 + 16;      // bytes per sequence row
    static constexpr int BYTES = MAXN * PITCH;
    // byte offset of element (row, d)
    __device__ static __forceinline__ int off(int row, int d) {
        if (SW) return row * PITCH + ((((d >> 3) ^ (row >> 1)) & 7) << 4) + (d & 7) * 2;
        return row * PITCH + d * 2;
    }
};

// stage rows [0,ROWS) x [0,DH) of a [n, ld] matrix (head slice at column c0) into an LDS tile; rows >= n and
// columns >= DH are zero.  Two halves, so that a kernel can issue the loads of ALL its tiles before it waits for any of
// them: as one loop (load, wait, write) the compiler kept a single 16-byte load in flight per thread, and the 12 dependent
// round trips of the backward's three tiles were most of its 21 us "load and stage" phase.
template <int DH, int ROWS, int NTHR, bool SW = false>
struct TileStage {
    static constexpr int CH = Tile<DH, SW>::DHP / 8;           // 16-byte chunks per row
    static constexpr int NIT = (ROWS * CH + NTHR - 1) / NTHR;
    uint4 v[NIT];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int ld, int n, int c0, int tid) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx / CH, c = idx % CH;
            v[i] = make_uint4(0, 0, 0, 0);
            if (idx < ROWS * CH && row < n && c * 8 < DH) v[i] = *reinterpret_cast<const uint4*>(base + (size_t)row * ld + c0 + c * 8);
        }
    }
    __device__ __forceinline__ void store(uint8_t* tile, int tid) const {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + i * NTHR;
            if (idx < ROWS * CH) *reinterpret_cast<uint4*>(tile + Tile<DH, SW>::off(idx / CH, (idx % CH) * 8)) = v[i];
        }
    }
};

// K-major fragment from a staged tile: row `row`, 8 features at d = s*16 + (lane>>5)*8 (rows >= n are zero in the tile)
template <int DH, bool SW = false>
__device__ __forceinline__ bf16x8_t lfrag(const uint8_t* tile, int row, int s, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(tile + Tile<DH, SW>::off(row, s * 16 + (lane >> 5) * 8));
}

// A operand "X^T": MFMA row = feature d0+(lane&31); k-slots j=0..7 <-> sequence index
//   seq = sbase + 8*(j>>2) + 4*(lane>>5) + (j&3)           (the order in which a lane's accumulator rows come)
template <int DH, bool TR, bool SW = false>
__device__ __forceinline__ bf16x8_t tfrag(const uint8_t* tile, int d0, int sbase, int lane) {
    using TL = Tile<DH, SW>;
    const int hi = lane >> 5;
    if (TR) {
        const int t = lane & 15;
        const int dcol = d0 + ((lane >> 4) & 1) * 16 + (t & 3) * 4;
        const int s0 = sbase + 4 * hi + (t >> 2);
        auto p0 = (__attribute__((address_space(3))) v4bf16s_t*)(tile + TL::off(s0, dcol));
        auto p1 = (__attribute__((address_space(3))) v4bf16s_t*)(tile + TL::off(s0 + 8, dcol));
        bf16x4_t lo = __builtin_bit_cast(bf16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p0));
        bf16x4_t hi4 = __builtin_bit_cast(bf16x4_t, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p1));
        return __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
        const int d = d0 + (lane & 31);
        bf16x8_t f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = sbase + 8 * (j >> 2) + 4 * hi + (j & 3);
            f[j] = *reinterpret_cast<const short*>(tile + TL::off(s, d));
        }
        return f;
    }
}

// registers 8u..8u+7 of an accumulator -> bf16 B operand (k-slots in the same order as tfrag's)
__device__ __forceinline__ bf16x8_t acc_to_frag(const f32x16_t& a, int u) {
    bf16x8_t f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (short)f2bf(a[8 * u + j]);
    return f;
}

// store an "X^T" accumulator (MFMA rows = features d0 .. d0+31, columns = sequence rows) as rows of X.  A lane owns sequence
// row (l&31) and, per register quad g, the 4 consecutive features 8g + 4*(l>>5) ..: the two half-waves hold the two halves of
// every 8-feature group, so they trade quads (v_permlane32_swap) -- the low half ends up with the whole groups 0 and 2, the
// high half with 1 and 3 -- and each lane writes two 16-byte segments instead of four 8-byte ones (the store tail of these
// kernels is issue-bound: half the store instructions).
template <int DH>
__device__ __forceinline__ void store_rows(const f32x16_t& a, bf16_t* __restrict__ base, int ld, int row, int nrows, int c0,
                                           int d0, int lane, float mul) {
    const int hi = lane >> 5;
    uint32_t w[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        w[g][0] = pack2bf(a[4 * g] * mul, a[4 * g + 1] * mul);
        w[g][1] = pack2bf(a[4 * g + 2] * mul, a[4 * g + 3] * mul);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {          // group pair (2p, 2p+1)
        const auto x = __builtin_amdgcn_permlane32_swap(w[2 * p][0], w[2 * p + 1][0], false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(w[2 * p][1], w[2 * p + 1][1], false, false);
        // low half: {own quad of group 2p, partner's quad of 2p}; high half: {partner's quad of 2p+1, own quad of 2p+1}
        const int d = d0 + 8 * (2 * p + hi);
        if (row < nrows && d < DH)
            *reinterpret_cast<uint4*>(base + (size_t)row * ld + c0 + d) = make_uint4(x[0], y[0], x[1], y[1]);
    }
}

// x in the lanes whose bit of the wave-uniform lane mask m is set: ONE v_cndmask with the SGPR pair as its condition
// (llvm.amdgcn.inverse.ballot: the compiler sees an ordinary select and keeps its hazard bookkeeping -- as inline assembly the
// select consumed v_exp_f32 results one instruction after they were issued, inside the transcendental forwarding hazard the
// compiler pads for its own instructions only: wrong values in a third of the attention tests; as C++ on the lane index --
// (m >> lane) & 1 ? x : 0 -- it is a 64-bit shift, an and, a compare and the select, per element)
__device__ __forceinline__ float lanes_of(float x, uint64_t m, float otherwise = 0.f) {
    return __builtin_amdgcn_inverse_ballot_w64(m) ? x : otherwise;
}
// the 64-lane mask "low half-wave iff lo, high half-wave iff hi" (scalar arithmetic when lo / hi are wave-uniform)
__device__ __forceinline__ uint64_t half_masks(bool lo, bool hi) {
    return (lo ? 0x00000000FFFFFFFFull : 0ull) | (hi ? 0xFFFFFFFF00000000ull : 0ull);
}

template <int DH, int NQF, int NKF, bool TR, bool DROP>
__global__ __launch_bounds__(NQF * 64) void sdpa_fwd_mfma(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                    const bf16_t* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                    bf16_t* __restrict__ o, float* __restrict__ lse, int H, int nq, int nk,
                                                    int ldq, int ldk, int ldv, int ldo, float scale,
                                                    float p_drop, float inv_keep, uint64_t seed,
                                                    const uint64_t* __restrict__ step_seed, VarLen vl,
                                                    uint32_t* __restrict__ keep_bits) {
    if (DROP) seed = with_step_seed(seed, step_seed);
    // one wave per 32-query fragment (NQF waves share the staged V tile of the (batch, head) problem)
    __shared__ __attribute__((aligned(16))) uint8_t vt[NKF * 32 * Tile<DH>::PITCH];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, j = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l31 = lane & 31;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    zero_pad_rows<bf16_t>(vl.q_off, vl.q_pad, b, (int)gridDim.x / H, o, ldo, h * DH, DH, tid, NQF * 64);
    const bf16_t* qb = q + (size_t)q0 * ldq;
    const bf16_t* kb = k + (size_t)k0 * ldk;
    const bf16_t* vb = v + (size_t)k0 * ldv;
    TileStage<DH, NKF * 32, NQF * 64> sv;
    sv.load(vb, ldv, nk, h * DH, tid);

    // S^T[key][q] for this wave's queries
    f32x16_t st[NKF];
#pragma unroll
    for (int i = 0; i < NKF; ++i) st[i] = zero16();
    {
        // every Q / K fragment of the wave is requested before the first MFMA (as `mfma(load, load)` in one loop the compiler
        // waited for each pair in turn: eight dependent round trips)
        bf16x8_t fq[DH / 16], fk[NKF][DH / 16];
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) {
            fq[s] = gfrag(qb, ldq, j * 32 + l31, nq, h * DH + s * 16 + hi * 8);
#pragma unroll
            for (int i = 0; i < NKF; ++i) fk[i][s] = gfrag(kb, ldk, i * 32 + l31, nk, h * DH + s * 16 + hi * 8);
        }
#pragma unroll
        for (int s = 0; s < DH / 16; ++s)
#pragma unroll
            for (int i = 0; i < NKF; ++i) st[i] = mfma32(fk[i][s], fq[s], st[i]);
    }
    sv.store(vt, tid);                    // (issued before the Q / K fragment loads above: one wait covers them all)
    // the keys that attend (mask and length) as one wave-uniform word: per accumulator register the "attends" decision is a lane
    // mask of two half-waves, built by scalar instructions and applied by one v_cndmask (cf. the backward kernel)
    const uint64_t kbits = key_bits(key_mask, b * nk_cap, nk_cap, lane) & (nk >= 64 ? ~0ull : (1ull << nk) - 1ull);
    const int qi = j * 32 + l31;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NKF; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key0 = i * 32 + acc_row(r, 0);                               // the low half-wave's key; the high one's is key0 + 4
            const float sc = lanes_of(st[i][r] * scale, half_masks((kbits >> key0) & 1ull, (kbits >> (key0 + 4)) & 1ull), -INFINITY);
            st[i][r] = sc;
            mx = fmaxf(mx, sc);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == -INFINITY) mx = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NKF; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = __expf(st[i][r] - mx); st[i][r] = e; sum += e; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    if (hi == 0 && qi < nq) lse[(size_t)bh * nq_cap + qi] = mx + __logf(sum);
    // The keep decisions of the wave's 32 x (NKF * 32) elements leave the kernel as well (keep_bits, see KeepBits): the
    // comparison of register (i, r) over the 64 lanes IS a 64-bit lane mask in a scalar register pair -- its two halves are the
    // 32-query keep words of keys i*32 + acc_row(r, 0) and i*32 + acc_row(r, 1) -- so lane (i*16 + r)*2 + h of `kw` receives
    // half h (v_writelane) and the wave stores NKF * 32 dwords once.  The backward then spends one v_cndmask per element (the
    // same scalar pair as the condition) or a bit test instead of evaluating the counter hash again, twice: the hash was a
    // quarter of its instruction stream (quarter-rate integer multiplies), and the mask cannot differ from the forward's.
    uint32_t kw = 0;
#pragma unroll
    for (int i = 0; i < NKF; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pv = st[i][r] * inv;
            if (DROP) {
                const bool keep = dropout_keep(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)(i * 32 + acc_row(r, hi)), p_drop);
                const uint64_t m = __ballot(keep);
                // (no v_writelane builtin in this compiler.  The ballot is a VALU write of an SGPR pair, and gfx950 wants two wait
                //  states before a VALU instruction reads such a register -- the compiler pads its own instructions, not the
                //  inside of an asm statement: hence the s_nop)
                asm("s_nop 1\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
                    : "+v"(kw) : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "n"((i * 16 + r) * 2), "n"((i * 16 + r) * 2 + 1));
                pv = keep ? pv * inv_keep : 0.f;
            }
            st[i][r] = pv;
        }
    if (DROP && keep_bits != nullptr && lane < NKF * 32) keep_bits[KeepBits::word0(bh, NQF, NKF, j) + lane] = kw;
    __syncthreads();                      // V tile staged by all waves
    // O^T[d][q] = sum_key V^T[d][key] P^T[key][q]
    constexpr int ND = (DH + 31) / 32;
#pragma unroll
    for (int id = 0; id < ND; ++id) {
        f32x16_t oa = zero16();
#pragma unroll
        for (int i = 0; i < NKF; ++i)
#pragma unroll
            for (int u = 0; u < 2; ++u) oa = mfma32(tfrag<DH, TR>(vt, id * 32, i * 32 + u * 16, lane), acc_to_frag(st[i], u), oa);
        store_rows<DH>(oa, o + (size_t)q0 * ldo, ldo, j * 32 + l31, nq, h * DH, id * 32, lane, 1.0f);
    }
}

template <int DH, int NQF, int NKF, bool TR, bool DROP, int NW, bool BITS>
__global__ __launch_bounds__(NW * 64, NW == 2 ? 3 : 2) void sdpa_bwd_mfma(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                    const bf16_t* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                    const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                    bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                                                    int H, int nq, int nk, int ldq, int ldk, int ldv, int ldo,
                                                    int lddq, int lddk, int lddv, float scale,
                                                    float p_drop, float inv_keep, uint64_t seed, float* __restrict__ cs_ws,
                                                    const uint64_t* __restrict__ step_seed, VarLen vl,
                                                    const uint32_t* __restrict__ keep_bits) {
    static_assert(!BITS || DROP, "saved keep decisions only exist with dropout");
    if (DROP && !BITS) seed = with_step_seed(seed, step_seed);
    // K, Q, dO are needed both as row fragments and transposed: staged in LDS (unpadded swizzled tiles for DH = 64).  V is only
    // ever read as row fragments -- its own row by the lane that owns the key (phase 2), the same rows as the A operand of
    // dP^T = V dO^T (phase 1) -- so every wave keeps the V fragments in registers, straight from global: three tiles instead
    // of four, 26 KiB per workgroup at 64 x 64, six workgroups (12 waves) per CU instead of four.  The kernel is latency-
    // bound (54 % of its wave cycles wait, 21 % issue VALU): occupancy is what it lacked.
    constexpr bool SW = DH == 64;
    using TL = Tile<DH, SW>;
    __shared__ __attribute__((aligned(16))) uint8_t tk[NKF * 32 * TL::PITCH];     // K   [key][d]
    __shared__ __attribute__((aligned(16))) uint8_t tq[NQF * 32 * TL::PITCH];     // Q   [q][d]
    __shared__ __attribute__((aligned(16))) uint8_t tdo[NQF * 32 * TL::PITCH];    // dO  [q][d]
    __shared__ float s_delta[MAXN];
    __shared__ float s_lse[MAXN];
    // bias gradients of the q/k/v projections without re-reading dQ/dK/dV: column sums factor through per-token scalars,
    //   colsum(dK)[d] = sum_q Q[q][d] * (sum_key dS[q][key]),  colsum(dV)[d] = sum_q dO[q][d] * (sum_key P~[q][key]),
    //   colsum(dQ)[d] = sum_key K[key][d] * (sum_q dS[q][key]),
    // and each inner sum runs over the registers of the lane that owns the token (phase 1: query, phase 2: key).
    __shared__ float s_ds_q[MAXN], s_p_q[MAXN], s_ds_k[MAXN];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    // NW waves share the staged tiles of one (batch, head) problem; the 32-query fragments of phase 1 and the 32-key
    // fragments of phase 2 are independent tasks dealt round-robin to the waves.
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l31 = lane & 31;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    zero_pad_rows<bf16_t>(vl.q_off, vl.q_pad, b, (int)gridDim.x / H, dq, lddq, h * DH, DH, tid, NW * 64);
    zero_pad_rows<bf16_t>(vl.k_off, vl.k_pad, b, (int)gridDim.x / H, dk, lddk, h * DH, DH, tid, NW * 64);
    zero_pad_rows<bf16_t>(vl.k_off, vl.k_pad, b, (int)gridDim.x / H, dv, lddv, h * DH, DH, tid, NW * 64);
    const bf16_t* qb = q + (size_t)q0 * ldq;
    const bf16_t* kb = k + (size_t)k0 * ldk;
    const bf16_t* vb = v + (size_t)k0 * ldv;
    const bf16_t* dob = dout + (size_t)q0 * ldo;
    bf16x8_t vf[NKF][DH / 16];                  // V row fragments: row i*32 + (lane & 31), features s*16 + (lane >> 5)*8 ..
#pragma unroll
    for (int i = 0; i < NKF; ++i)
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) vf[i][s] = gfrag(vb, ldv, i * 32 + l31, nk, h * DH + s * 16 + hi * 8);
    {
        TileStage<DH, NKF * 32, NW * 64, SW> sk;
        TileStage<DH, NQF * 32, NW * 64, SW> sq, sd;
        sk.load(kb, ldk, nk, h * DH, tid);
        sq.load(qb, ldq, nq, h * DH, tid);
        sd.load(dob, ldo, nq, h * DH, tid);
        sk.store(tk, tid);
        sq.store(tq, tid);
        sd.store(tdo, tid);
    }
    if (tid < MAXN) s_lse[tid] = tid < nq ? lse[(size_t)bh * nq_cap + tid] : 0.f;
    // Which elements exist at all is wave-uniform per accumulator register and half-wave -- the keys that attend (mask and
    // length) as a 64-bit word, the queries below nq likewise -- so the "exists" decision of register (i, r) is a LANE MASK
    // put together by scalar instructions and applied by one v_cndmask (lanes_of), not five vector instructions per element.
    const uint64_t kbits = key_bits(key_mask, b * nk_cap, nk_cap, lane) & (nk >= 64 ? ~0ull : (1ull << nk) - 1ull);
    const uint64_t qbits = nq >= 64 ? ~0ull : (1ull << nq) - 1ull;
    constexpr int ND = (DH + 31) / 32;
    __syncthreads();

    // ---------------- phase 1: lane owns a query column.  dQ and delta.  One 32-query fragment at a time (a runtime
    // loop: the accumulators of one fragment are live, not all of them -> 2 waves / SIMD instead of 1).
#pragma unroll 1
    for (int j = wave; j < NQF; j += NW) {
        f32x16_t st[NKF], dpt[NKF];
#pragma unroll
        for (int i = 0; i < NKF; ++i) { st[i] = zero16(); dpt[i] = zero16(); }
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) {
            const bf16x8_t fq = lfrag<DH, SW>(tq, j * 32 + l31, s, lane), fdo = lfrag<DH, SW>(tdo, j * 32 + l31, s, lane);
#pragma unroll
            for (int i = 0; i < NKF; ++i) {
                st[i] = mfma32(lfrag<DH, SW>(tk, i * 32 + l31, s, lane), fq, st[i]);
                dpt[i] = mfma32(vf[i][s], fdo, dpt[i]);
            }
        }
        const int qi = j * 32 + l31;
        const float l = s_lse[qi];
        float delta = 0.f, psum = 0.f;
        const uint32_t qw = (uint32_t)(qbits >> (j * 32));                        // the fragment's queries that exist, bit = lane & 31
        const uint64_t qmask = (uint64_t)qw | ((uint64_t)qw << 32);
        // (BITS) the forward's keep decisions of this fragment: dwords 2n, 2n+1 = the lane mask of register n = i*16 + r
        const uint64_t* __restrict__ kmask = reinterpret_cast<const uint64_t*>(keep_bits + KeepBits::word0(bh, NQF, NKF, j));
#pragma unroll
        for (int i = 0; i < NKF; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key0 = i * 32 + acc_row(r, 0);                           // the low half-wave's key; the high one's is key0 + 4
                const uint64_t ok = half_masks((kbits >> key0) & 1ull, (kbits >> (key0 + 4)) & 1ull) & qmask;
                const float pv = lanes_of(__expf(st[i][r] * scale - l), ok);
                float dp = dpt[i][r];
                if (DROP) {
                    float msk;
                    if (BITS) msk = lanes_of(inv_keep, kmask[i * 16 + r]);
                    else msk = dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)(key0 + 4 * hi), p_drop, inv_keep);
                    dp *= msk;
                    psum += pv * msk;
                } else psum += pv;
                st[i][r] = pv;
                dpt[i][r] = dp;
                delta += pv * dp;
            }
        delta += __shfl_xor(delta, 32, 64);
        if (hi == 0) s_delta[qi] = delta;
        float dssum = 0.f;
#pragma unroll
        for (int i = 0; i < NKF; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[i][r] = st[i][r] * (dpt[i][r] - delta) * scale;       // dS^T
                dssum += st[i][r];
            }
        if (cs_ws != nullptr) {
            psum += __shfl_xor(psum, 32, 64);
            dssum += __shfl_xor(dssum, 32, 64);
            if (hi == 0) { s_p_q[qi] = psum; s_ds_q[qi] = dssum; }
        }
        // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int id = 0; id < ND; ++id) {
            f32x16_t qa = zero16();
#pragma unroll
            for (int i = 0; i < NKF; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    qa = mfma32(tfrag<DH, TR, SW>(tk, id * 32, i * 32 + u * 16, lane), acc_to_frag(st[i], u), qa);
            store_rows<DH>(qa, dq + (size_t)q0 * lddq, lddq, j * 32 + l31, nq, h * DH, id * 32, lane, 1.0f);
        }
    }
    __syncthreads();
    // ---------------- phase 2: lane owns a key column.  dV and dK.  One 32-key fragment at a time.
#pragma unroll 1
    for (int i = wave; i < NKF; i += NW) {
        f32x16_t s2[NQF], dp2[NQF];
#pragma unroll
        for (int j = 0; j < NQF; ++j) { s2[j] = zero16(); dp2[j] = zero16(); }
#pragma unroll
        for (int s = 0; s < DH / 16; ++s) {
            const bf16x8_t fk = lfrag<DH, SW>(tk, i * 32 + l31, s, lane);
            bf16x8_t fv = vf[0][s];             // fragment i of the register copy (i is a run-time wave index: select, do not index)
#pragma unroll
            for (int ii = 1; ii < NKF; ++ii) fv = (i == ii) ? vf[ii][s] : fv;
#pragma unroll
            for (int j = 0; j < NQF; ++j) {
                s2[j] = mfma32(lfrag<DH, SW>(tq, j * 32 + l31, s, lane), fk, s2[j]);
                dp2[j] = mfma32(lfrag<DH, SW>(tdo, j * 32 + l31, s, lane), fv, dp2[j]);
            }
        }
        const int key = i * 32 + l31;
        const uint32_t kw32 = (uint32_t)(kbits >> (i * 32));                        // the fragment's keys that attend, bit = lane & 31
        const uint64_t kmask2 = (uint64_t)kw32 | ((uint64_t)kw32 << 32);
        // (BITS) this lane's key: its keep word per query fragment (bit t = query j*32 + t), pre-shifted by the half-wave's 4
        uint32_t kwd[NQF];
#pragma unroll
        for (int j = 0; j < NQF; ++j)
            kwd[j] = BITS ? keep_bits[KeepBits::word0(bh, NQF, NKF, j) + KeepBits::dword(i, l31)] >> (4 * hi) : 0u;
#pragma unroll
        for (int j = 0; j < NQF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi0 = j * 32 + acc_row(r, 0);                             // the low half-wave's query; the high one's is qi0 + 4
                const int qi = qi0 + 4 * hi;
                const uint64_t ok = half_masks((qbits >> qi0) & 1ull, (qbits >> (qi0 + 4)) & 1ull) & kmask2;
                const float pv = lanes_of(__expf(s2[j][r] * scale - s_lse[qi]), ok);
                float msk = 1.f;
                if (DROP) {
                    if (BITS) msk = ((kwd[j] >> acc_row(r, 0)) & 1u) ? inv_keep : 0.f;
                    else msk = dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)key, p_drop, inv_keep);
                }
                const float dp = dp2[j][r] * msk;
                dp2[j][r] = pv * (dp - s_delta[qi]) * scale;      // dS[q][key]
                s2[j][r] = pv * msk;                              // P~[q][key]
            }
        if (cs_ws != nullptr) {
            float ksum = 0.f;
#pragma unroll
            for (int j = 0; j < NQF; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ksum += dp2[j][r];
            ksum += __shfl_xor(ksum, 32, 64);
            if (hi == 0) s_ds_k[key] = ksum;
        }
        // dV^T[d][key] = sum_q dO^T[d][q] P~[q][key] ;  dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]
#pragma unroll
        for (int id = 0; id < ND; ++id) {
            f32x16_t va = zero16(), ka = zero16();
#pragma unroll
            for (int j = 0; j < NQF; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    va = mfma32(tfrag<DH, TR, SW>(tdo, id * 32, j * 32 + u * 16, lane), acc_to_frag(s2[j], u), va);
                    ka = mfma32(tfrag<DH, TR, SW>(tq, id * 32, j * 32 + u * 16, lane), acc_to_frag(dp2[j], u), ka);
                }
            store_rows<DH>(va, dv + (size_t)k0 * lddv, lddv, i * 32 + l31, nk, h * DH, id * 32, lane, 1.0f);
            store_rows<DH>(ka, dk + (size_t)k0 * lddk, lddk, i * 32 + l31, nk, h * DH, id * 32, lane, 1.0f);
        }
    }
    if (cs_ws != nullptr) {
        // slab b of the workspace: [q | k | v] x [H*DH] partial bias gradients of this batch element (plain stores: every
        // element has exactly one writer; the second stage sums the B slabs)
        __syncthreads();
        for (int w = tid; w < 3 * DH; w += NW * 64) {
            const int which = w / DH, d = w % DH;
            float acc = 0.f;
            if (which == 0) {
                for (int key = 0; key < NKF * 32; ++key)
                    acc += bf2f(*reinterpret_cast<const bf16_t*>(tk + TL::off(key, d))) * s_ds_k[key];
            } else {
                const uint8_t* tile = which == 1 ? tq : tdo;
                const float* sc = which == 1 ? s_ds_q : s_p_q;
                for (int qi = 0; qi < NQF * 32; ++qi)
                    acc += bf2f(*reinterpret_cast<const bf16_t*>(tile + TL::off(qi, d))) * sc[qi];
            }
            cs_ws[((size_t)b * 3 + which) * (H * DH) + h * DH + d] = acc;
        }
    }
}

// ================================================================== long sequences on the matrix cores (nq or nk in 65..512, bf16)
// --max_text_length above 64 (ref param.py:140): the same 32x32x16 MFMA machinery as the on-chip kernels, walked over 64-key (resp.
// 64-query) blocks.  Forward: a workgroup per (batch, head, 64-query block), two waves of 32 queries; per key block the scores
// S^T = K Q^T (a lane owns a query), an ONLINE softmax -- running maximum and sum per lane, the output accumulators O^T[d][q] rescaled
// by the lane's exp(m_old - m_new), dropout applied to the un-normalised terms -- and O^T += V^T P^T with V staged in LDS; the division
// by the sum and the log-sum-exp at the end.  Backward in two launches like the plain fallback above (sdpa_bwd_long_*): per query
// block, two passes over the key blocks (delta = sum_j P~ dP, then dQ^T += K^T dS^T; delta also goes to the caller's scratch), and per
// key block one pass over the query blocks (dV^T += dO^T P~, dK^T += Q^T dS) with the saved log-sum-exp and delta.  No atomics; same
// dropout counters, packed rows and zeroed pad rows as everywhere else.
template <int DH, bool TR, bool DROP>
__global__ __launch_bounds__(128) void sdpa_fwd_flash(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                      const bf16_t* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                      bf16_t* __restrict__ o, float* __restrict__ lse, int H, int nq, int nk,
                                                      int ldq, int ldk, int ldv, int ldo, float scale, float p_drop, float inv_keep,
                                                      uint64_t seed, const uint64_t* __restrict__ step_seed, VarLen vl) {
    if (DROP) seed = with_step_seed(seed, step_seed);
    __shared__ __attribute__((aligned(16))) uint8_t vt[64 * Tile<DH>::PITCH];
    const int bh = blockIdx.x, b = bh / H, h = bh % H, qb0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    if (blockIdx.y == 0) zero_pad_rows<bf16_t>(vl.q_off, vl.q_pad, b, (int)gridDim.x / H, o, ldo, h * DH, DH, tid, 128);
    if (qb0 >= nq) return;
    const bf16_t* qb = q + (size_t)q0 * ldq;
    const bf16_t* kb = k + (size_t)k0 * ldk;
    const bf16_t* vb = v + (size_t)k0 * ldv;
    const int qi = qb0 + j * 32 + l31;
    constexpr int ND = (DH + 31) / 32;
    bf16x8_t fq[DH / 16];
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) fq[s] = gfrag(qb, ldq, qi, nq, h * DH + s * 16 + hi * 8);
    f32x16_t oa[ND];
#pragma unroll
    for (int id = 0; id < ND; ++id) oa[id] = zero16();
    float mx = -INFINITY, sum = 0.f;
    for (int kb0 = 0; kb0 < nk; kb0 += 64) {
        const int nkb = nk - kb0;                      // keys left from this block on (rows >= nkb of the block do not exist)
        TileStage<DH, 64, 128> sv;
        sv.load(vb + (size_t)kb0 * ldv, ldv, nkb, h * DH, tid);
        bf16x8_t fk[2][DH / 16];
#pragma unroll
        for (int s = 0; s < DH / 16; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) fk[i][s] = gfrag(kb + (size_t)kb0 * ldk, ldk, i * 32 + l31, nkb, h * DH + s * 16 + hi * 8);
        f32x16_t st[2];
        st[0] = zero16(); st[1] = zero16();
#pragma unroll
        for (int s = 0; s < DH / 16; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) st[i] = mfma32(fk[i][s], fq[s], st[i]);
        if (kb0 > 0) __syncthreads();                  // both waves are done with the previous block's V tile
        sv.store(vt, tid);
        const uint64_t kbits = key_bits(key_mask, b * nk_cap + kb0, min(64, nk_cap - kb0), lane);
        float bm = -INFINITY;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = i * 32 + acc_row(r, hi);
                const bool ok = key < nkb && ((kbits >> key) & 1ull) != 0;
                const float sc = ok ? st[i][r] * scale : -INFINITY;
                st[i][r] = sc;
                bm = fmaxf(bm, sc);
            }
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float nm = fmaxf(mx, bm);
        const float corr = (mx == -INFINITY) ? 0.f : __expf(mx - nm);      // (nm = -inf only while every key so far is masked: sum = 0)
        float bs = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = st[i][r] == -INFINITY ? 0.f : __expf(st[i][r] - nm);
                bs += e;
                float pv = e;
                if (DROP) pv *= dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)(kb0 + i * 32 + acc_row(r, hi)), p_drop, inv_keep);
                st[i][r] = pv;
            }
        bs += __shfl_xor(bs, 32, 64);
        sum = sum * corr + bs;
        mx = nm;
#pragma unroll
        for (int id = 0; id < ND; ++id)
#pragma unroll
            for (int r = 0; r < 16; ++r) oa[id][r] *= corr;
        __syncthreads();                               // V tile staged by both waves
#pragma unroll
        for (int id = 0; id < ND; ++id)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) oa[id] = mfma32(tfrag<DH, TR>(vt, id * 32, i * 32 + u * 16, lane), acc_to_frag(st[i], u), oa[id]);
    }
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    if (hi == 0 && qi < nq) lse[(size_t)bh * nq_cap + qi] = (mx == -INFINITY ? 0.f : mx) + __logf(sum);
#pragma unroll
    for (int id = 0; id < ND; ++id) store_rows<DH>(oa[id], o + (size_t)q0 * ldo, ldo, qi, nq, h * DH, id * 32, lane, inv);
}

// backward, query side: delta and dQ of a 64-query block (two waves of 32 queries), two passes over the key blocks
template <int DH, bool TR, bool DROP>
__global__ __launch_bounds__(128) void sdpa_bwd_flash_q(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                        const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                        bf16_t* __restrict__ dq, float* __restrict__ delta_out, int H, int nq, int nk,
                                                        int ldq, int ldk, int ldv, int ldo, int lddq, float scale, float p_drop,
                                                        float inv_keep, uint64_t seed, const uint64_t* __restrict__ step_seed, VarLen vl) {
    if (DROP) seed = with_step_seed(seed, step_seed);
    constexpr bool SW = DH == 64;
    using TL = Tile<DH, SW>;
    __shared__ __attribute__((aligned(16))) uint8_t tk[64 * TL::PITCH];
    const int bh = blockIdx.x, b = bh / H, h = bh % H, qb0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    if (blockIdx.y == 0) zero_pad_rows<bf16_t>(vl.q_off, vl.q_pad, b, (int)gridDim.x / H, dq, lddq, h * DH, DH, tid, 128);
    if (qb0 >= nq) return;
    const bf16_t* qb = q + (size_t)q0 * ldq;
    const bf16_t* kb = k + (size_t)k0 * ldk;
    const bf16_t* vb = v + (size_t)k0 * ldv;
    const bf16_t* dob = dout + (size_t)q0 * ldo;
    const int qi = qb0 + j * 32 + l31;
    constexpr int ND = (DH + 31) / 32;
    bf16x8_t fq[DH / 16], fdo[DH / 16];
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
        fq[s] = gfrag(qb, ldq, qi, nq, h * DH + s * 16 + hi * 8);
        fdo[s] = gfrag(dob, ldo, qi, nq, h * DH + s * 16 + hi * 8);
    }
    const float l = qi < nq ? lse[(size_t)bh * nq_cap + qi] : 0.f;
    float delta = 0.f;
    f32x16_t qa[ND];
#pragma unroll
    for (int id = 0; id < ND; ++id) qa[id] = zero16();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll 1
        for (int kb0 = 0; kb0 < nk; kb0 += 64) {
            const int nkb = nk - kb0;
            TileStage<DH, 64, 128, SW> sk;
            sk.load(kb + (size_t)kb0 * ldk, ldk, nkb, h * DH, tid);
            bf16x8_t fv[2][DH / 16];
#pragma unroll
            for (int s = 0; s < DH / 16; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) fv[i][s] = gfrag(vb + (size_t)kb0 * ldv, ldv, i * 32 + l31, nkb, h * DH + s * 16 + hi * 8);
            __syncthreads();                           // both waves are done with the previous K tile
            sk.store(tk, tid);
            const uint64_t kbits = key_bits(key_mask, b * nk_cap + kb0, min(64, nk_cap - kb0), lane);
            __syncthreads();
            f32x16_t st[2], dpt[2];
            st[0] = zero16(); st[1] = zero16(); dpt[0] = zero16(); dpt[1] = zero16();
#pragma unroll
            for (int s = 0; s < DH / 16; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    st[i] = mfma32(lfrag<DH, SW>(tk, i * 32 + l31, s, lane), fq[s], st[i]);
                    dpt[i] = mfma32(fv[i][s], fdo[s], dpt[i]);
                }
            float dacc = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = i * 32 + acc_row(r, hi);
                    const bool ok = key < nkb && ((kbits >> key) & 1ull) != 0 && qi < nq;
                    const float pv = ok ? __expf(st[i][r] * scale - l) : 0.f;
                    float dp = dpt[i][r];
                    if (DROP) dp *= dropout_scale(seed, (uint32_t)(bh * nq_cap + qi), (uint32_t)(kb0 + key), p_drop, inv_keep);
                    dacc += pv * dp;
                    st[i][r] = pv * (dp - delta) * scale;          // dS^T (second pass: delta is final)
                }
            if (pass == 0) delta += dacc;
            else {
#pragma unroll
                for (int id = 0; id < ND; ++id)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            qa[id] = mfma32(tfrag<DH, TR, SW>(tk, id * 32, i * 32 + u * 16, lane), acc_to_frag(st[i], u), qa[id]);
            }
        }
        if (pass == 0) delta += __shfl_xor(delta, 32, 64);
    }
    if (hi == 0 && qi < nq) delta_out[(size_t)bh * nq_cap + qi] = delta;
#pragma unroll
    for (int id = 0; id < ND; ++id) store_rows<DH>(qa[id], dq + (size_t)q0 * lddq, lddq, qi, nq, h * DH, id * 32, lane, 1.0f);
}

// backward, key side: dK and dV of a 64-key block (two waves of 32 keys), one pass over the query blocks
template <int DH, bool TR, bool DROP>
__global__ __launch_bounds__(128) void sdpa_bwd_flash_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, const uint8_t* __restrict__ key_mask,
                                                        const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                        const float* __restrict__ delta_in, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv,
                                                        int H, int nq, int nk, int ldq, int ldk, int ldv, int ldo, int lddk, int lddv,
                                                        float scale, float p_drop, float inv_keep, uint64_t seed,
                                                        const uint64_t* __restrict__ step_seed, VarLen vl) {
    if (DROP) seed = with_step_seed(seed, step_seed);
    constexpr bool SW = DH == 64;
    using TL = Tile<DH, SW>;
    __shared__ __attribute__((aligned(16))) uint8_t tq[64 * TL::PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t tdo[64 * TL::PITCH];
    __shared__ float s_lse[64], s_delta[64];
    const int bh = blockIdx.x, b = bh / H, h = bh % H, kb0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, i = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int nq_cap = nq, nk_cap = nk;
    const int q0 = vl.row0_q(b, nq), k0 = vl.row0_k(b, nk);
    nq = vl.len_q(b, nq); nk = vl.len_k(b, nk);
    if (blockIdx.y == 0) {
        zero_pad_rows<bf16_t>(vl.k_off, vl.k_pad, b, (int)gridDim.x / H, dk, lddk, h * DH, DH, tid, 128);
        zero_pad_rows<bf16_t>(vl.k_off, vl.k_pad, b, (int)gridDim.x / H, dv, lddv, h * DH, DH, tid, 128);
    }
    if (kb0 >= nk) return;
    const bf16_t* qb = q + (size_t)q0 * ldq;
    const bf16_t* kb = k + (size_t)k0 * ldk;
    const bf16_t* vb = v + (size_t)k0 * ldv;
    const bf16_t* dob = dout + (size_t)q0 * ldo;
    const int key = kb0 + i * 32 + l31;
    constexpr int ND = (DH + 31) / 32;
    bf16x8_t fk[DH / 16], fv[DH / 16];
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) {
        fk[s] = gfrag(kb, ldk, key, nk, h * DH + s * 16 + hi * 8);
        fv[s] = gfrag(vb, ldv, key, nk, h * DH + s * 16 + hi * 8);
    }
    const uint64_t kbits = key_bits(key_mask, b * nk_cap + kb0, min(64, nk_cap - kb0), lane);
    const bool kok = key < nk && ((kbits >> (i * 32 + l31)) & 1ull) != 0;
    f32x16_t va[ND], ka[ND];
#pragma unroll
    for (int id = 0; id < ND; ++id) { va[id] = zero16(); ka[id] = zero16(); }
#pragma unroll 1
    for (int qb0 = 0; qb0 < nq; qb0 += 64) {
        const int nqb = nq - qb0;
        TileStage<DH, 64, 128, SW> sq, sd;
        sq.load(qb + (size_t)qb0 * ldq, ldq, nqb, h * DH, tid);
        sd.load(dob + (size_t)qb0 * ldo, ldo, nqb, h * DH, tid);
        const float lq = (tid < 64 && tid < nqb) ? lse[(size_t)bh * nq_cap + qb0 + tid] : 0.f;
        const float dq_ = (tid < 64 && tid < nqb) ? delta_in[(size_t)bh * nq_cap + qb0 + tid] : 0.f;
        __syncthreads();                               // both waves are done with the previous block's tiles
        sq.store(tq, tid);
        sd.store(tdo, tid);
        if (tid < 64) { s_lse[tid] = lq; s_delta[tid] = dq_; }
        __syncthreads();
        f32x16_t s2[2], dp2[2];
        s2[0] = zero16(); s2[1] = zero16(); dp2[0] = zero16(); dp2[1] = zero16();
#pragma unroll
        for (int s = 0; s < DH / 16; ++s)
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                s2[jq] = mfma32(lfrag<DH, SW>(tq, jq * 32 + l31, s, lane), fk[s], s2[jq]);
                dp2[jq] = mfma32(lfrag<DH, SW>(tdo, jq * 32 + l31, s, lane), fv[s], dp2[jq]);
            }
#pragma unroll
        for (int jq = 0; jq < 2; ++jq)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = jq * 32 + acc_row(r, hi);            // query inside the block
                const float e = __expf(s2[jq][r] * scale - s_lse[ql]);
                const float pv = (kok && ql < nqb) ? e : 0.f;
                float msk = 1.f;
                if (DROP) msk = dropout_scale(seed, (uint32_t)(bh * nq_cap + qb0 + ql), (uint32_t)key, p_drop, inv_keep);
                const float dp = dp2[jq][r] * msk;
                dp2[jq][r] = pv * (dp - s_delta[ql]) * scale;      // dS[q][key]
                s2[jq][r] = pv * msk;                              // P~[q][key]
            }
#pragma unroll
        for (int id = 0; id < ND; ++id)
#pragma unroll
            for (int jq = 0; jq < 2; ++jq)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    va[id] = mfma32(tfrag<DH, TR, SW>(tdo, id * 32, jq * 32 + u * 16, lane), acc_to_frag(s2[jq], u), va[id]);
                    ka[id] = mfma32(tfrag<DH, TR, SW>(tq, id * 32, jq * 32 + u * 16, lane), acc_to_frag(dp2[jq], u), ka[id]);
                }
    }
#pragma unroll
    for (int id = 0; id < ND; ++id) {
        store_rows<DH>(va[id], dv + (size_t)k0 * lddv, lddv, key, nk, h * DH, id * 32, lane, 1.0f);
        store_rows<DH>(ka[id], dk + (size_t)k0 * lddk, lddk, key, nk, h * DH, id * 32, lane, 1.0f);
    }
}

// XL_SDPA_LONG_MFMA=0: the plain long-sequence kernels (any dtype) for every launch
static bool long_mfma_enabled() {
    static const bool on = [] { const char* e = getenv("XL_SDPA_LONG_MFMA"); return e == nullptr || atoi(e) != 0; }();
    return on;
}

struct SdpaArgs {
    const void *q, *k, *v; const uint8_t* key_mask; const void* dout; void* o; float* lse;
    void *dq, *dk, *dv;
    int B, H, nq, nk, dh, ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    float scale, p_drop, inv_keep; uint64_t seed;
    float* cs_ws;
    VarLen vl;
    uint32_t* keep_bits;
};

template <int DH, int NQF, int NKF, bool TR, bool DROP>
static void launch_fwd2(const SdpaArgs& a, hipStream_t st) {
    hipLaunchKernelGGL((sdpa_fwd_mfma<DH, NQF, NKF, TR, DROP>), dim3(a.B * a.H), dim3(NQF * 64), 0, st, (const bf16_t*)a.q,
                       (const bf16_t*)a.k, (const bf16_t*)a.v, a.key_mask, (bf16_t*)a.o, a.lse, a.H, a.nq, a.nk, a.ldq, a.ldk,
                       a.ldv, a.ldo, a.scale, a.p_drop, a.inv_keep, a.seed, ctx().step_seed, a.vl, a.keep_bits);
}
template <int DH, int NQF, int NKF, bool TR, bool DROP, bool BITS = false>
static void launch_bwd2(const SdpaArgs& a, hipStream_t st) {
    constexpr int NW = NQF > NKF ? NQF : NKF;
    hipLaunchKernelGGL((sdpa_bwd_mfma<DH, NQF, NKF, TR, DROP, NW, BITS>), dim3(a.B * a.H), dim3(NW * 64), 0, st, (const bf16_t*)a.q,
                       (const bf16_t*)a.k, (const bf16_t*)a.v, a.key_mask, (const bf16_t*)a.dout, a.lse, (bf16_t*)a.dq,
                       (bf16_t*)a.dk, (bf16_t*)a.dv, a.H, a.nq, a.nk, a.ldq, a.ldk, a.ldv, a.ldo, a.lddq, a.lddk, a.lddv,
                       a.scale, a.p_drop, a.inv_keep, a.seed, a.cs_ws, ctx().step_seed, a.vl, (const uint32_t*)a.keep_bits);
}
template <int DH, int NQF, int NKF>
static void launch_fwd(const SdpaArgs& a, hipStream_t st) {
    const bool drop = a.p_drop > 0.f;
    if (ctx().use_tr_read) { if (drop) launch_fwd2<DH, NQF, NKF, true, true>(a, st); else launch_fwd2<DH, NQF, NKF, true, false>(a, st); }
    else { if (drop) launch_fwd2<DH, NQF, NKF, false, true>(a, st); else launch_fwd2<DH, NQF, NKF, false, false>(a, st); }
}
template <int DH, int NQF, int NKF>
static void launch_bwd(const SdpaArgs& a, hipStream_t st) {
    const bool drop = a.p_drop > 0.f, bits = drop && a.keep_bits != nullptr;
    if (ctx().use_tr_read) {
        if (bits) launch_bwd2<DH, NQF, NKF, true, true, true>(a, st);
        else if (drop) launch_bwd2<DH, NQF, NKF, true, true>(a, st);
        else launch_bwd2<DH, NQF, NKF, true, false>(a, st);
    } else {
        if (bits) launch_bwd2<DH, NQF, NKF, false, true, true>(a, st);
        else if (drop) launch_bwd2<DH, NQF, NKF, false, true>(a, st);
        else launch_bwd2<DH, NQF, NKF, false, false>(a, st);
    }
}

template <bool FWD, int DH>
static void dispatch_frags(const SdpaArgs& a, hipStream_t st) {
    const int nqf = (a.nq + 31) / 32, nkf = (a.nk + 31) / 32;
#define XL_CASE(QF, KF)                                                       \
    if (nqf == QF && nkf == KF) {                                             \
        if (FWD) launch_fwd<DH, QF, KF>(a, st); else launch_bwd<DH, QF, KF>(a, st); \
        return;                                                               \
    }
    XL_CASE(1, 1) XL_CASE(1, 2) XL_CASE(2, 1) XL_CASE(2, 2)
#undef XL_CASE
}

static bool mfma_eligible(const SdpaArgs& a, bool bwd) {
    if (!(a.dh == 16 || a.dh == 32 || a.dh == 64)) return false;
    if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 8) return false;
    if (!aligned16(a.q) || !aligned16(a.k) || !aligned16(a.v)) return false;
    if (bwd) {
        if (a.lddq % 8 || a.lddk % 8 || a.lddv % 8) return false;          // 16-byte output stores
        if (!aligned16(a.dout) || !aligned16(a.dq) || !aligned16(a.dk) || !aligned16(a.dv)) return false;
    } else if (!aligned16(a.o)) return false;
    return true;
}

static int check_common(const SdpaArgs& a, int dtype, const char* fn) {
    XL_CHECK_ARG(dtype == XL_F32 || dtype == XL_BF16, XL_ERR_BAD_DTYPE, "%s: bad dtype %d", fn, dtype);
    XL_CHECK_ARG(a.B > 0 && a.H > 0 && a.nq > 0 && a.nk > 0 && a.nq <= MAXLONG && a.nk <= MAXLONG, XL_ERR_BAD_SHAPE,
                 "%s: need 1 <= nq,nk <= %d (got %d,%d)", fn, MAXLONG, a.nq, a.nk);
    XL_CHECK_ARG(a.dh > 0 && a.dh <= MAXN, XL_ERR_BAD_SHAPE, "%s: head size %d not in 1..%d", fn, a.dh, MAXN);
    XL_CHECK_ARG(a.p_drop >= 0.f && a.p_drop < 1.f, XL_ERR_BAD_ARG, "%s: p_drop %f", fn, a.p_drop);
    return XL_OK;
}

}  // namespace xl

using namespace xl;

extern "C" int64_t xl_sdpa_keep_bits_bytes(int B, int H, int nq, int nk, int dh, int dtype) {
    if (B <= 0 || H <= 0 || nq <= 0 || nk <= 0 || nq > MAXN || nk > MAXN || dtype != XL_BF16 || !(dh == 16 || dh == 32 || dh == 64)) return 0;
    return (int64_t)KeepBits::word0(B * H, (nq + 31) / 32, (nk + 31) / 32, 0) * 4;
}

extern "C" int xl_sdpa_fwd(const void* q, const void* k, const void* v, const uint8_t* key_mask,
                           void* o, float* lse, int B, int H, int nq, int nk, int dh,
                           int ldq, int ldk, int ldv, int ldo, float scale,
                           float p_drop, uint64_t seed, const int* q_rowoff, const int* k_rowoff, int q_rows_padded,
                           int k_rows_padded, uint32_t* keep_bits, int dtype, void* stream) {
    SdpaArgs a = {};
    a.vl = VarLen{q_rowoff, k_rowoff, q_rows_padded, k_rows_padded};
    a.keep_bits = keep_bits;
    a.q = q; a.k = k; a.v = v; a.key_mask = key_mask; a.o = o; a.lse = lse;
    a.B = B; a.H = H; a.nq = nq; a.nk = nk; a.dh = dh; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.scale = scale; a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    int rc = check_common(a, dtype, "xl_sdpa_fwd");
    if (rc) return rc;
    XL_CHECK_ARG(q && k && v && o && lse, XL_ERR_BAD_ARG, "xl_sdpa_fwd: null pointer");
    XL_CHECK_ARG(keep_bits == nullptr || (xl_sdpa_keep_bits_bytes(B, H, nq, nk, dh, dtype) > 0 && mfma_eligible(a, false) && aligned16(keep_bits)),
                 XL_ERR_BAD_ARG, "xl_sdpa_fwd: keep_bits only exists on the on-chip bf16 path (nq, nk <= %d, dh 16/32/64, 16-byte "
                 "aligned operands and row strides)", MAXN);
    hipStream_t st = (hipStream_t)stream;
    if ((nq > MAXN || nk > MAXN) && dtype == XL_BF16 && mfma_eligible(a, false) && ctx().use_tr_read && long_mfma_enabled()) {
        const dim3 grid(B * H, (nq + 63) / 64);     // long sequences on the matrix cores
        const bool drop = p_drop > 0.f;
#define XL_FLASH_FWD(DH_)                                                                                                        \
        if (drop) hipLaunchKernelGGL((sdpa_fwd_flash<DH_, true, true>), grid, dim3(128), 0, st, (const bf16_t*)q, (const bf16_t*)k,  \
                                     (const bf16_t*)v, key_mask, (bf16_t*)o, lse, H, nq, nk, ldq, ldk, ldv, ldo, scale, p_drop,  \
                                     a.inv_keep, seed, ctx().step_seed, a.vl);                                                   \
        else hipLaunchKernelGGL((sdpa_fwd_flash<DH_, true, false>), grid, dim3(128), 0, st, (const bf16_t*)q, (const bf16_t*)k,     \
                                (const bf16_t*)v, key_mask, (bf16_t*)o, lse, H, nq, nk, ldq, ldk, ldv, ldo, scale, p_drop,       \
                                a.inv_keep, seed, ctx().step_seed, a.vl);
        if (dh == 64) { XL_FLASH_FWD(64) } else if (dh == 32) { XL_FLASH_FWD(32) } else { XL_FLASH_FWD(16) }
#undef XL_FLASH_FWD
        XL_CHECK_LAUNCH();
        return XL_OK;
    }
    if (nq > MAXN || nk > MAXN) {                 // long sequences: the plain kernels (any dtype)
        const dim3 grid(B * H, (nq + 63) / 64);
        if (dtype == XL_BF16)
            hipLaunchKernelGGL((sdpa_fwd_long<bf16_t>), grid, dim3(64), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask,
                               (bf16_t*)o, lse, H, nq, nk, dh, ldq, ldk, ldv, ldo, scale, p_drop, a.inv_keep, seed, ctx().step_seed, a.vl);
        else
            hipLaunchKernelGGL((sdpa_fwd_long<float>), grid, dim3(64), 0, st, (const float*)q, (const float*)k, (const float*)v, key_mask,
                               (float*)o, lse, H, nq, nk, dh, ldq, ldk, ldv, ldo, scale, p_drop, a.inv_keep, seed, ctx().step_seed, a.vl);
        XL_CHECK_LAUNCH();
        return XL_OK;
    }
    if (dtype == XL_BF16 && mfma_eligible(a, false)) {
        if (dh == 64) dispatch_frags<true, 64>(a, st);
        else if (dh == 32) dispatch_frags<true, 32>(a, st);
        else dispatch_frags<true, 16>(a, st);
    } else if (dtype == XL_BF16) {
        hipLaunchKernelGGL((sdpa_fwd_generic<bf16_t>), dim3(B * H), dim3(64), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, key_mask, (bf16_t*)o, lse, H, nq, nk, dh, ldq, ldk, ldv, ldo, scale, p_drop,
                           a.inv_keep, seed, ctx().step_seed, a.vl);
    } else {
        hipLaunchKernelGGL((sdpa_fwd_generic<float>), dim3(B * H), dim3(64), 0, st, (const float*)q, (const float*)k,
                           (const float*)v, key_mask, (float*)o, lse, H, nq, nk, dh, ldq, ldk, ldv, ldo, scale, p_drop,
                           a.inv_keep, seed, ctx().step_seed, a.vl);
    }
    XL_CHECK_LAUNCH();
    return XL_OK;
}

extern "C" int xl_sdpa_bwd(const void* q, const void* k, const void* v, const uint8_t* key_mask,
                           const void* dout, const float* lse,
                           void* dq, void* dk, void* dv, int B, int H, int nq, int nk, int dh,
                           int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv, float scale,
                           float p_drop, uint64_t seed, float* bias_grad, float* workspace,
                           const int* q_rowoff, const int* k_rowoff, int q_rows_padded, int k_rows_padded,
                           const uint32_t* keep_bits, int dtype, void* stream) {
    SdpaArgs a = {};
    a.vl = VarLen{q_rowoff, k_rowoff, q_rows_padded, k_rows_padded};
    a.keep_bits = const_cast<uint32_t*>(keep_bits);
    a.q = q; a.k = k; a.v = v; a.key_mask = key_mask; a.dout = dout; a.lse = const_cast<float*>(lse);
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.B = B; a.H = H; a.nq = nq; a.nk = nk; a.dh = dh; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.scale = scale; a.p_drop = p_drop; a.inv_keep = 1.0f / (1.0f - p_drop); a.seed = seed;
    int rc = check_common(a, dtype, "xl_sdpa_bwd");
    if (rc) return rc;
    XL_CHECK_ARG(q && k && v && dout && lse && dq && dk && dv, XL_ERR_BAD_ARG, "xl_sdpa_bwd: null pointer");
    XL_CHECK_ARG(keep_bits == nullptr || (xl_sdpa_keep_bits_bytes(B, H, nq, nk, dh, dtype) > 0 && mfma_eligible(a, true) && aligned16(keep_bits)),
                 XL_ERR_BAD_ARG, "xl_sdpa_bwd: keep_bits only exists on the on-chip bf16 path (nq, nk <= %d, dh 16/32/64, 16-byte "
                 "aligned operands, gradients and row strides)", MAXN);
    hipStream_t st = (hipStream_t)stream;
    const int HD = H * dh;
    const bool is_long = nq > MAXN || nk > MAXN;
    const bool fused_bias = !is_long && bias_grad != nullptr && workspace != nullptr && dtype == XL_BF16 && mfma_eligible(a, true) &&
                            (int64_t)B * 3 * HD <= xl_workspace_floats(HD);
    if (is_long) {
        // delta [B, H, nq] goes through the caller's workspace (free again before the bias column sums below use it)
        XL_CHECK_ARG(workspace != nullptr && (int64_t)B * H * nq <= xl_workspace_floats(HD), XL_ERR_BAD_ARG,
                     "xl_sdpa_bwd: sequences longer than %d need the workspace (xl_workspace_floats(H * dh) floats)", MAXN);
        const dim3 gq(B * H, (nq + 63) / 64), gk(B * H, (nk + 63) / 64);
        if (dtype == XL_BF16 && mfma_eligible(a, true) && ctx().use_tr_read && long_mfma_enabled()) {
            const bool drop = p_drop > 0.f;
#define XL_FLASH_BWD(DH_, DR_)                                                                                                   \
            hipLaunchKernelGGL((sdpa_bwd_flash_q<DH_, true, DR_>), gq, dim3(128), 0, st, (const bf16_t*)q, (const bf16_t*)k,        \
                               (const bf16_t*)v, key_mask, (const bf16_t*)dout, lse, (bf16_t*)dq, workspace, H, nq, nk, ldq, ldk,  \
                               ldv, ldo, lddq, scale, p_drop, a.inv_keep, seed, ctx().step_seed, a.vl);                            \
            hipLaunchKernelGGL((sdpa_bwd_flash_k<DH_, true, DR_>), gk, dim3(128), 0, st, (const bf16_t*)q, (const bf16_t*)k,        \
                               (const bf16_t*)v, key_mask, (const bf16_t*)dout, lse, workspace, (bf16_t*)dk, (bf16_t*)dv, H, nq, nk, \
                               ldq, ldk, ldv, ldo, lddk, lddv, scale, p_drop, a.inv_keep, seed, ctx().step_seed, a.vl);
            if (dh == 64) { if (drop) { XL_FLASH_BWD(64, true) } else { XL_FLASH_BWD(64, false) } }
            else if (dh == 32) { if (drop) { XL_FLASH_BWD(32, true) } else { XL_FLASH_BWD(32, false) } }
            else { if (drop) { XL_FLASH_BWD(16, true) } else { XL_FLASH_BWD(16, false) } }
#undef XL_FLASH_BWD
        } else if (dtype == XL_BF16) {
            hipLaunchKernelGGL((sdpa_bwd_long_q<bf16_t>), gq, dim3(64), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask,
                               (const bf16_t*)dout, lse, (bf16_t*)dq, workspace, H, nq, nk, dh, ldq, ldk, ldv, ldo, lddq, scale, p_drop,
                               a.inv_keep, seed, ctx().step_seed, a.vl);
            hipLaunchKernelGGL((sdpa_bwd_long_k<bf16_t>), gk, dim3(64), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, key_mask,
                               (const bf16_t*)dout, lse, workspace, (bf16_t*)dk, (bf16_t*)dv, H, nq, nk, dh, ldq, ldk, ldv, ldo, lddk, lddv,
                               scale, p_drop, a.inv_keep, seed, ctx().step_seed, a.vl);
        } else {
            hipLaunchKernelGGL((sdpa_bwd_long_q<float>), gq, dim3(64), 0, st, (const float*)q, (const float*)k, (const float*)v, key_mask,
                               (const float*)dout, lse, (float*)dq, workspace, H, nq, nk, dh, ldq, ldk, ldv, ldo, lddq, scale, p_drop,
                               a.inv_keep, seed, ctx().step_seed, a.vl);
            hipLaunchKernelGGL((sdpa_bwd_long_k<float>), gk, dim3(64), 0, st, (const float*)q, (const float*)k, (const float*)v, key_mask,
                               (const float*)dout, lse, workspace, (float*)dk, (float*)dv, H, nq, nk, dh, ldq, ldk, ldv, ldo, lddk, lddv,
                               scale, p_drop, a.inv_keep, seed, ctx().step_seed, a.vl);
        }
    } else if (dtype == XL_BF16 && mfma_eligible(a, true)) {
        a.cs_ws = fused_bias ? workspace : nullptr;
        if (dh == 64) dispatch_frags<false, 64>(a, st);
        else if (dh == 32) dispatch_frags<false, 32>(a, st);
        else dispatch_frags<false, 16>(a, st);
        XL_CHECK_LAUNCH();
        if (fused_bias) {
            xl::launch_colsum_reduce(workspace, B, 3 * HD, bias_grad, st);        // bias_grad[q | k | v] += sum of the B slabs
            XL_CHECK_LAUNCH();
            return XL_OK;
        }
    } else if (dtype == XL_BF16) {
        hipLaunchKernelGGL((sdpa_bwd_generic<bf16_t>), dim3(B * H), dim3(64), 0, st, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, key_mask, (const bf16_t*)dout, lse, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, H, nq,
                           nk, dh, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, p_drop, a.inv_keep, seed, ctx().step_seed, a.vl);
    } else {
        hipLaunchKernelGGL((sdpa_bwd_generic<float>), dim3(B * H), dim3(64), 0, st, (const float*)q, (const float*)k,
                           (const float*)v, key_mask, (const float*)dout, lse, (float*)dq, (float*)dk, (float*)dv, H, nq, nk,
                           dh, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, p_drop, a.inv_keep, seed, ctx().step_seed, a.vl);
    }
    XL_CHECK_LAUNCH();
    if (bias_grad != nullptr) {          // no fused partials on this path: column sums of the stored gradients
        const int rq = q_rowoff ? q_rows_padded : B * nq, rk = k_rowoff ? k_rows_padded : B * nk;      // (pad rows are zero)
        // three column sums in flight at once (deferred combines): a third of the caller's xl_workspace_floats(HD) = 4096 * HD floats each
        constexpr size_t kWsThird = 4096 / 3;
        int rc2 = xl_colsum(dq, bias_grad, rq, HD, lddq, workspace, dtype, stream);
        if (rc2 == XL_OK) rc2 = xl_colsum(dk, bias_grad + HD, rk, HD, lddk, workspace ? workspace + kWsThird * (size_t)HD : nullptr, dtype, stream);
        if (rc2 == XL_OK) rc2 = xl_colsum(dv, bias_grad + 2 * HD, rk, HD, lddv, workspace ? workspace + 2 * kWsThird * (size_t)HD : nullptr, dtype, stream);
        return rc2;
    }
    return XL_OK;
}

extern "C" int xl_attn_probs(const void* q, const void* k, const uint8_t* key_mask, const float* lse, float* probs,
                             int B, int H, int nq, int nk, int dh, int ldq, int ldk, float scale, float p_drop, uint64_t seed,
                             const int* q_rowoff, const int* k_rowoff, int dtype, void* stream) {
    SdpaArgs a = {};
    a.B = B; a.H = H; a.nq = nq; a.nk = nk; a.dh = dh; a.p_drop = p_drop;
    int rc = check_common(a, dtype, "xl_attn_probs");
    if (rc) return rc;
    XL_CHECK_ARG(q && k && lse && probs, XL_ERR_BAD_ARG, "xl_attn_probs: null pointer");
    const VarLen vl{q_rowoff, k_rowoff, 0, 0};
    const float inv_keep = 1.0f / (1.0f - p_drop);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == XL_BF16)
        hipLaunchKernelGGL((attn_probs_kernel<bf16_t>), dim3(B * H, (nq + 63) / 64), dim3(64), 0, st, (const bf16_t*)q, (const bf16_t*)k, key_mask, lse,
                           probs, H, nq, nk, dh, ldq, ldk, scale, p_drop, inv_keep, seed, ctx().step_seed, vl);
    else
        hipLaunchKernelGGL((attn_probs_kernel<float>), dim3(B * H, (nq + 63) / 64), dim3(64), 0, st, (const float*)q, (const float*)k, key_mask, lse,
                           probs, H, nq, nk, dh, ldq, ldk, scale, p_drop, inv_keep, seed, ctx().step_seed, vl);
    XL_CHECK_LAUNCH();
    return XL_OK;
}
