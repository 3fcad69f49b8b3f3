// "Quad-occupancy" tiles: 128 x 192 output tiles by EIGHT waves of 32 x 96 (three 32x32x16 accumulators: 48 registers), at most 128
// registers per wave and 80 KiB of LDS per workgroup -- TWO workgroups per CU, sixteen waves, four per SIMD.
//
// Why: a K = 768 tile of the 256 x 256 ping-pong kernel spends 30-40 % of its CU time outside the K loop
// (prologue, epilogue, hand-over to the next workgroup) and nothing overlaps it, because one workgroup owns the CU.  The "duo" tiles
// (two four-wave workgroups per CU) did not hide it either: that K loop needs TWO waves per SIMD (one reads fragments while the other's
// MFMAs run) and each duo workgroup has one, so a workgroup whose partner is in its epilogue computes at half rate.  Here every
// workgroup brings two waves per SIMD of its own, and there is no choreography to disturb: a wave reads the four fragments of a k
// step, issues three MFMAs, and the other three waves of its SIMD cover the latencies -- occupancy instead of a ping-pong schedule.
// While one workgroup runs its prologue / epilogue, the other's K loop still has two waves per SIMD.
//
// K loop: a two-slot ring of 40 KiB stages [A 128 x 64 | B as three 64-row parts], filled by LDS-DMA (5 pieces per wave and stage,
// swizzle on the source address: OpTile), ONE barrier per K tile:
//     compute(slot t) ; lgkmcnt(0) ; vmcnt(0) [stage t+1, requested a whole stage ago] ; barrier ; request stage t+2 into slot t
// Same MFMA, same k order per output element as the other bf16 kernels: bit-identical results (tests/test_hip_kernels.py).
#include "gemm_pp_kernel.h"

namespace xl {

template <bool BKM, int EPIK>
__global__ __launch_bounds__(512, 4) void gemm_bf16_q_kernel(GemmParams p) {
    constexpr int BM = 128, BN = 192, WTN = 96;
    using TA = OpTile<true, 128>;
    using TB = OpTile<BKM, 64>;
    constexpr int AB = TA::BYTES, BPB = TB::BYTES, BUF = AB + 3 * BPB;      // 16 KiB + 3 x 8 KiB
    static_assert(BUF == 40960, "stage geometry");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];          // [2 slots][A | B0 | B1 | B2]
    int tm, tn, z;
    tile_coords(p, tm, tn, z);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    f32x16_t acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // per-lane sources of this wave's 1 KiB pieces of a stage (byte offsets, k0 excluded): two of A, one of each B part
    uint32_t srca[2], srcb[3];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        int rs, c;
        TA::decode((wave * 2 + pt) * 1024 + lane * 16, rs, c);
        srca[pt] = 2u * ((uint32_t)(m0 + rs) * (uint32_t)p.lda + c * 8);
    }
    {
        int rs, c;
        TB::decode(wave * 1024 + lane * 16, rs, c);
        const int lr = BKM ? rs : c * 8;                                    // local row of the part (first of 8 when M-major)
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int gn = n0 + (lr >> 5) * WTN + h * 32 + (lr & 31);
            srcb[h] = 2u * (BKM ? (uint32_t)gn * (uint32_t)p.ldb + c * 8 : (uint32_t)rs * (uint32_t)p.ldb + gn);
        }
    }
    const auto rsrc_of = [](const void* ptr, uint32_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(p.A, (uint32_t)(((size_t)(p.M - 1) * p.lda + p.K) * 2));
    const __amdgpu_buffer_rsrc_t rb = rsrc_of(p.B, (uint32_t)(((size_t)((BKM ? p.N : p.K) - 1) * p.ldb + (BKM ? p.K : p.N)) * 2));
    auto stage = [&](int kt) {
        const int k0 = kt * BK;
        uint8_t* dst = smem + (kt & 1) * BUF;
        const uint32_t sa = (uint32_t)k0 * 2u, sb = (uint32_t)(BKM ? k0 : k0 * p.ldb) * 2u;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dst + (wave * 2 + pt) * 1024), 16,
                                                     (int)srca[pt], (int)sa, 0, 0);
#pragma unroll
        for (int h = 0; h < 3; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dst + AB + h * BPB + wave * 1024), 16,
                                                     (int)srcb[h], (int)sb, 0, 0);
    };
    [[maybe_unused]] uint32_t b_tr = 0;
    if constexpr (!BKM) b_tr = tr_lane_off<64>(wc * 32, lane);
    const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem;
    auto compute = [&](int kt) {
        const uint8_t* buf = smem + (kt & 1) * BUF;
        bf16x8_t fa[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) fa[s] = TA::template frag<true>(buf, wr * 32, s, lane);
        if constexpr (BKM) {
            bf16x8_t fb[3][4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int h = 0; h < 3; ++h) fb[h][s] = TB::template frag<true>(buf + AB + h * BPB, wc * 32, s, lane);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int h = 0; h < 3; ++h)
                    acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf16_t, fa[s]), __builtin_bit_cast(v8bf16_t, fb[h][s]),
                                                                     acc[h], 0, 0, 0);
        } else {
            // M-major B: transpose reads through inline assembly (gemm_common.h TrFrag), two k steps at a time -- all twelve fragments
            // at once would not fit the 128 registers
            const uint32_t ad = smem_lds + (uint32_t)(buf - smem) + b_tr;
            using R0 = TrFrag<TB::RP, AB>;
            using R1 = TrFrag<TB::RP, AB + BPB>;
            using R2 = TrFrag<TB::RP, AB + 2 * BPB>;
            bf16x8_t fb[3][2];
            auto mma2 = [&](int s0) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int h = 0; h < 3; ++h)
                        acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf16_t, fa[s0 + s]),
                                                                         __builtin_bit_cast(v8bf16_t, fb[h][s]), acc[h], 0, 0, 0);
            };
            fb[0][0] = R0::template get<0>(ad); fb[0][1] = R0::template get<1>(ad);
            fb[1][0] = R1::template get<0>(ad); fb[1][1] = R1::template get<1>(ad);
            fb[2][0] = R2::template get<0>(ad); fb[2][1] = R2::template get<1>(ad);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma2(0);
            __builtin_amdgcn_sched_barrier(0);
            fb[0][0] = R0::template get<2>(ad); fb[0][1] = R0::template get<3>(ad);
            fb[1][0] = R1::template get<2>(ad); fb[1][1] = R1::template get<3>(ad);
            fb[2][0] = R2::template get<2>(ad); fb[2][1] = R2::template get<3>(ad);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma2(2);
        }
    };
    auto stamp = [&](int i) {
        if (p.trace != nullptr && tid == 0) p.trace[(size_t)blockIdx.x * 4 + i] = wall_clock64();
    };
    stamp(0);
    [[maybe_unused]] const uint64_t dseed = dropout_seed_of<EPIK>(p);
    float bias8[8], bias8b[8];
    load_bias8(p, lane, true, n0 + wc * WTN, bias8);
    sub_load_bias8<32>(p, lane, true, n0 + wc * WTN + 64, bias8b);

    const int nkt = p.K / BK;                               // (host: K % 64 == 0)
    stage(0);
    if (nkt > 1) { stage(1); wait_vmcnt<5>(); } else { wait_vmcnt<0>(); }
    hard_barrier();
    stamp(1);
    for (int kt = 0; kt < nkt; ++kt) {
        compute(kt);
        if (kt + 1 < nkt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vmcnt<0>();                                // stage kt + 1: requested a whole stage ago
            hard_barrier();                                 // slot kt & 1 is free, stage kt + 1 is visible to every wave
            if (kt + 2 < nkt) stage(kt + 2);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    hard_barrier();                                         // every fragment read of the ring is done: the epilogue reuses it
    stamp(2);
    if (p.ablate & 4) return;

    // ---- epilogue: the 32 x 64 half quad, then the 32 x 32 third accumulator, through 8 KiB of wave-private LDS
    const int mw = m0 + wr * 32, nw = n0 + wc * WTN;
    float* wbuf = reinterpret_cast<float*>(smem + wave * 8192);
    QuadOperand op0, op1;
    quad_operand_load<EPIK, 4>(p, lane, mw, nw, op0);
    __builtin_amdgcn_sched_barrier(0);
    hquad_to_lds(wbuf, lane, acc[0], acc[1]);
    __builtin_amdgcn_sched_barrier(0);
    sub_operand_load<EPIK, 32, 32>(p, lane, mw, nw + 64, op1);
    __builtin_amdgcn_sched_barrier(0);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    epilogue_rows_fast<EPIK, 4>(p, wbuf, lane, true, mw, nw, op0, cs, bias8, dseed);
    __builtin_amdgcn_sched_barrier(0);
    acc32_to_lds(wbuf, lane, acc[2]);
    __builtin_amdgcn_sched_barrier(0);
    sub_rows_fast<EPIK, 32, false, 32>(p, wbuf, lane, mw, nw + 64, op1, bias8b, dseed, cs);
    if (p.trace != nullptr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(3); }
}

template <bool BKM, int EPIK>
static hipError_t launch_q_one(const GemmParams& p, int nblk, hipStream_t st) {
    constexpr int lds = 81920;
    hipError_t e = hipSuccess;
    static bool attr = false;
    auto k = gemm_bf16_q_kernel<BKM, EPIK>;
    if (!attr) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL(k, dim3(nblk), dim3(512), lds, st, p);
    return e;
}

bool q_has_instance(int b_kmajor, int epik) {
    if (epik == XL_EPI_NONE || epik == XL_EPI_RESIDUAL) return true;
    return b_kmajor ? epik == XL_EPI_GELU_DG : epik == XL_EPI_MULAUX;
}

hipError_t launch_q(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st) {
    if (b_kmajor) {
        switch (epik) {
            case XL_EPI_NONE: return launch_q_one<true, XL_EPI_NONE>(p, nblk, st);
            case XL_EPI_RESIDUAL: return launch_q_one<true, XL_EPI_RESIDUAL>(p, nblk, st);
            case XL_EPI_GELU_DG: return launch_q_one<true, XL_EPI_GELU_DG>(p, nblk, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (epik) {
        case XL_EPI_NONE: return launch_q_one<false, XL_EPI_NONE>(p, nblk, st);
        case XL_EPI_RESIDUAL: return launch_q_one<false, XL_EPI_RESIDUAL>(p, nblk, st);
        case XL_EPI_MULAUX: return launch_q_one<false, XL_EPI_MULAUX>(p, nblk, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace xl
