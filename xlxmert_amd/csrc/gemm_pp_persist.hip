// Persistent variant of the 256x256 ping-pong kernel for launches of SEVERAL rounds of tiles with a short contraction
// (N = 3072 / K = 768: the FFN's first Linear forward and the gradient through its GELU -- three rounds of 256 tiles).
//
// Per tile the plain kernel pays, besides its K loop, a prologue (descriptor set-up + the first LDS-DMA round trip, ~1.4 us),
// an epilogue (2.5-7 us) and the hand-over of the CU to the next workgroup (~3.5 us until that one's first barrier):
// at K = 768 (12 K tiles, ~15 us) a third of the tile's time.  Here ONE workgroup per CU walks its share of the tiles and
// requests the first K tile of the NEXT output tile right after the last barrier of its K loop -- before the epilogue --
// so the epilogue's transposition, operand loads and stores run under that round trip, and no workgroup is torn down or
// dispatched between tiles.  What makes that possible: the epilogue transposes through ONE half of the staging ring
// (8 KiB per wave: 32 x 64 half quads instead of 64 x 64 quads), leaving the other half to the incoming K tile; the second
// K tile is requested when the epilogue's last LDS read is behind a barrier.  The wait in front of the next K loop is the
// ordinary counted one: loads return in order, so "at most 8 requests outstanding" implies that the first K tile has
// landed however many of the epilogue's stores are still in flight.
// (A persistent loop WITHOUT this prefetch was measured in round 2 and lost to the hardware dispatcher.)
// MEASURED (round 3, tools/gemm_bench.py and bench.py A/B on one box): bit-identical output; 16384 x 3072 x 768 + GELU_DG
// 103-104 -> 97-99 us alone, the MULAUX launch of the same shape 86 -> 86 us; the whole training step 18.57-18.64 -> 18.70 ms.
// The step runs four streams whose kernels interleave at tile boundaries; a workgroup that keeps its CU for three tiles takes
// those boundaries away from the other streams, which costs more than the hand-over it saves.  The kernel is therefore
// OPT-IN (xl_set_gemm_persistent / XL_GEMM_PERSIST=1): the right choice for a single-stream caller, not for this step.
// Tiles are dealt statically, XCD-aware: block b (XCD b % 8) takes the tiles b, b + grid, b + 2 grid ... of the same
// XCD-contiguous order the plain kernel uses, so the workgroups co-resident on an XCD still share operand panels.
// Restrictions (the host only sends such launches here): A K-major, bf16 in / out, M % 256 == N % 256 == K % 64 == 0,
// K >= 128, a fast-epilogue kind, no K split.
#include "gemm_pp_kernel.h"

namespace xl {

template <bool BKM, int EPIK, bool CS>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp_persist_kernel(GemmParams p) {
    constexpr bool AK = true;
    constexpr int WC = 4, AF = 2, NB = 2;                               // PPGeo<256>
    constexpr int WTM = 128, WTN = 64, HR = 64, BROWS = 128;
    using TA = OpTile<AK, 128>;
    using TB = OpTile<BKM, BROWS>;
    constexpr int HT = 16384, BPB = TB::BYTES, PB = BPB / 8192;         // 16 KiB parts, 2 pieces per wave and part
    constexpr int BUF = 2 * HT + NB * BPB;                              // 64 KiB: [A0 | B0 | B1 | A1]
    constexpr int NA = 2, NBP = NB * PB;
    constexpr int WAIT = 2 * NA + NBP;
    static_assert(BUF == 65536 && PB == 2, "256x256 geometry");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];      // [2 buffers][A0 | B parts | A1]
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int grp = wave >> 2;

    const auto rsrc_of = [](const void* ptr, uint32_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(A, (uint32_t)(((size_t)(p.M - 1) * p.lda + p.K) * 2));
    const __amdgpu_buffer_rsrc_t rb = rsrc_of(B, (uint32_t)(((size_t)((BKM ? p.N : p.K) - 1) * p.ldb + (BKM ? p.K : p.N)) * 2));

    // tile-independent part of the per-lane source offsets (bytes): the lane's 16 bytes inside a 1 KiB piece
    uint32_t la[2][NA], lb[NB][PB];
#pragma unroll
    for (int pt = 0; pt < NA; ++pt) {
        int rs, c;
        TA::decode((wave * NA + pt) * 1024 + lane * 16, rs, c);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int lr = (rs / HR) * WTM + h * HR + (rs % HR);
            la[h][pt] = 2u * ((uint32_t)lr * (uint32_t)p.lda + c * 8);
        }
    }
#pragma unroll
    for (int pt = 0; pt < PB; ++pt) {
        int rs, c;
        TB::decode((wave * PB + pt) * 1024 + lane * 16, rs, c);
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int lr = BKM ? rs : c * 8;
            const int ln = (lr >> 5) * WTN + h * 32 + (lr & 31);
            lb[h][pt] = 2u * (BKM ? (uint32_t)ln * (uint32_t)p.ldb + c * 8 : (uint32_t)rs * (uint32_t)p.ldb + ln);
        }
    }
    // tile-dependent part: a scalar byte offset of the tile's first row / column
    auto stage_a = [&](uint32_t abase, auto H, int kt) {
        constexpr int h = decltype(H)::value;
        uint8_t* dst = smem + (kt & 1) * BUF + (h == 0 ? 0 : HT + NB * BPB) + wave * (NA * 1024);
        const uint32_t soff = abase + (uint32_t)(kt * BK) * 2u;
#pragma unroll
        for (int pt = 0; pt < NA; ++pt)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dst + pt * 1024), 16,
                                                     (int)la[h][pt], (int)soff, 0, 0);
    };
    auto stage_b = [&](uint32_t bbase, int kt) {
        const uint32_t soff = bbase + (uint32_t)(BKM ? kt * BK : kt * BK * p.ldb) * 2u;
        uint8_t* dst0 = smem + (kt & 1) * BUF + HT + wave * (PB * 1024);
#pragma unroll
        for (int h = 0; h < NB; ++h)
#pragma unroll
            for (int pt = 0; pt < PB; ++pt)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dst0 + h * BPB + pt * 1024),
                                                         16, (int)lb[h][pt], (int)soff, 0, 0);
    };
    const int total = p.tiles_m * p.tiles_n;
    const int q8 = total >> 3, r8 = total & 7;
    auto tile_at = [&](int vb, int& tm, int& tn) {          // virtual block id -> tile (XCD-contiguous grouped order)
        const int xcd = vb & 7, pos = vb >> 3;
        tile_of(p, (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + pos, tm, tn);
    };
    auto bases = [&](int tm, int tn, uint32_t& ab, uint32_t& bb) {
        ab = __builtin_amdgcn_readfirstlane((uint32_t)(tm * 256) * (uint32_t)p.lda * 2u);
        bb = __builtin_amdgcn_readfirstlane(BKM ? (uint32_t)(tn * 256) * (uint32_t)p.ldb * 2u : (uint32_t)(tn * 256) * 2u);
    };

    f32x16_t acc[2 * AF][NB];
    bf16x8_t fa[AF][4], fb[NB][4];
    auto read_a = [&](const uint8_t* buf, auto H) {
        const uint8_t* t = buf + (decltype(H)::value == 0 ? 0 : HT + NB * BPB);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < AF; ++i) fa[i][s] = TA::template frag<true>(t, wr * HR + i * 32, s, lane);
    };
    auto read_b = [&](const uint8_t* buf) {
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const uint8_t* t = buf + HT + h * BPB;
#pragma unroll
            for (int s = 0; s < 4; ++s) fb[h][s] = TB::template frag<true>(t, wc * 32, s, lane);
        }
    };
    auto mma2 = [&](auto AH) {
        constexpr int ah = decltype(AH)::value;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int bh = 0; bh < NB; ++bh)
#pragma unroll
                for (int i = 0; i < AF; ++i)
                    acc[ah * AF + i][bh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(v8bf16_t, fa[i][s]), __builtin_bit_cast(v8bf16_t, fb[bh][s]), acc[ah * AF + i][bh], 0, 0, 0);
    };
    uint32_t ab = 0, bb = 0;                                 // current tile's operand bases
    auto phase = [&](auto X, auto WAITC, auto ISSUE, int kt) {
        constexpr int x = decltype(X)::value;
        const uint8_t* buf = smem + (kt & 1) * BUF;
        if constexpr (x == 0) {
            read_b(buf);
            read_a(buf, ic<0>{});
        } else {
            read_a(buf, ic<1>{});
        }
        if constexpr (decltype(ISSUE)::value != 0) {
            if constexpr (x == 0) {
                stage_a(ab, ic<1>{}, kt + 1);
            } else {
                stage_a(ab, ic<0>{}, kt + 2); stage_b(bb, kt + 2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<decltype(WAITC)::value>();
        hard_barrier();
        __builtin_amdgcn_s_setprio(1);
        if constexpr (x == 0) mma2(ic<0>{});
        else mma2(ic<1>{});
        __builtin_amdgcn_s_setprio(0);
        hard_barrier();
    };

    const int nkt = p.K / BK;                                // >= 2
    const uint64_t dseed = dropout_seed_of<EPIK>(p);
    int vb = blockIdx.x;
    int tm, tn;
    tile_at(vb, tm, tn);
    bases(tm, tn, ab, bb);
    // prologue of the first tile: K tile 0 complete, plus A0 | B of K tile 1
    stage_a(ab, ic<0>{}, 0); stage_b(bb, 0); stage_a(ab, ic<1>{}, 0);
    stage_a(ab, ic<0>{}, 1); stage_b(bb, 1);
    wait_vmcnt<WAIT>();
    hard_barrier();
    for (;;) {
#pragma unroll
        for (int i = 0; i < 2 * AF; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (grp == 1) hard_barrier();                 // waves 4-7 run one barrier behind waves 0-3
        for (int kt = 0; kt < nkt - 2; ++kt) {
            phase(ic<0>{}, ic<WAIT>{}, ic<1>{}, kt);
            phase(ic<1>{}, ic<WAIT>{}, ic<1>{}, kt);
        }
        phase(ic<0>{}, ic<WAIT>{}, ic<1>{}, nkt - 2);
        phase(ic<1>{}, ic<NA>{}, ic<0>{}, nkt - 2);
        phase(ic<0>{}, ic<0>{}, ic<0>{}, nkt - 1);
        phase(ic<1>{}, ic<0>{}, ic<0>{}, nkt - 1);
        if (grp == 0) hard_barrier();                 // both groups: every fragment read of the ring is behind a barrier
        // ---- the next tile's first K tile goes into buffer 0 now; the epilogue works in buffer 1
        const int vn = vb + (int)gridDim.x;
        const bool has_next = vn < total;
        int tm2 = 0, tn2 = 0;
        uint32_t ab2 = 0, bb2 = 0;
        if (has_next) {
            tile_at(vn, tm2, tn2);
            bases(tm2, tn2, ab2, bb2);
            stage_a(ab2, ic<0>{}, 0); stage_b(bb2, 0); stage_a(ab2, ic<1>{}, 0);
        }
        // ---- epilogue: four 32 x 64 half quads of the wave's 128 x 64 tile through 8 KiB of wave-private LDS (whole 128-byte
        // row segments of C / aux per 8 lanes, as in the plain kernel's quads)
        {
            const int mw = tm * 256 + wr * WTM, nw = tn * 256 + wc * WTN;
            float* wbuf = reinterpret_cast<float*>(smem + BUF + wave * 8192);
            float bv[8];
            load_bias8(p, lane, true, nw, bv);
            float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2 * AF; ++i) {
                QuadOperand op;
                quad_operand_load<EPIK, 4>(p, lane, mw + i * 32, nw, op);
                __builtin_amdgcn_sched_barrier(0);
                hquad_to_lds(wbuf, lane, acc[i][0], acc[i][1]);
                __builtin_amdgcn_sched_barrier(0);
                epilogue_rows_fast<EPIK, 4>(p, wbuf, lane, true, mw + i * 32, nw, op, cs, bv, dseed);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (p.colsum_ws != nullptr) colsum_flush(p, lane, mw >> 7, nw, cs);      // one slab per 128 rows
        }
        if (!has_next) break;
        hard_barrier();                               // every wave is done with buffer 1
        stage_a(ab2, ic<0>{}, 1); stage_b(bb2, 1);
        // loads return in order: at most WAIT requests outstanding => K tile 0 (A0, B) of the new tile has landed, whatever
        // the epilogue's stores are doing
        wait_vmcnt<WAIT>();
        hard_barrier();
        vb = vn; tm = tm2; tn = tn2; ab = ab2; bb = bb2;
    }
}

template <bool BKM, int EPIK, bool CS = false>
static hipError_t launch_persist_one(const GemmParams& p, int nblk, hipStream_t st) {
    constexpr int lds = 131072;
    hipError_t e = hipSuccess;
    auto k = gemm_bf16_pp_persist_kernel<BKM, EPIK, CS>;
    static bool attr = false;
    if (!attr) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL(k, dim3(nblk), dim3(512), lds, st, p);
    return e;
}

// -> hipErrorInvalidValue when there is no persistent instance for this (layout, epilogue kind)
hipError_t launch_pp_persist(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st) {
    if (b_kmajor) {
        if (p.colsum_ws != nullptr) return hipErrorInvalidValue;
        switch (epik) {
            case XL_EPI_NONE: return launch_persist_one<true, XL_EPI_NONE>(p, nblk, st);
            case XL_EPI_GELU: return launch_persist_one<true, XL_EPI_GELU>(p, nblk, st);
            case XL_EPI_GELU_DG: return launch_persist_one<true, XL_EPI_GELU_DG>(p, nblk, st);
            default: return hipErrorInvalidValue;
        }
    }
    const bool cs = p.colsum_ws != nullptr;
    switch (epik) {
        case XL_EPI_NONE: return cs ? hipErrorInvalidValue : launch_persist_one<false, XL_EPI_NONE>(p, nblk, st);
        case XL_EPI_MULAUX: return cs ? launch_persist_one<false, XL_EPI_MULAUX, true>(p, nblk, st) : launch_persist_one<false, XL_EPI_MULAUX>(p, nblk, st);
        case XL_EPI_DGELU: return cs ? hipErrorInvalidValue : launch_persist_one<false, XL_EPI_DGELU>(p, nblk, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace xl
