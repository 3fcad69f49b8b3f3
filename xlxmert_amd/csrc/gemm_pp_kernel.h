// The 256x256 / 256x192 ping-pong kernel templates (see gemm.hip for the entry point and the 128x128 kernel).  Included by the
// translation units that instantiate them -- gemm_pp.hip (forward layout, 256 wide + grouped weight gradients), gemm_pp_nn.hip (dX and
// the remaining layouts, 256 wide), gemm_pp_192.hip (256x192 tiles): three compiles side by side instead of one of six minutes.
#pragma once
#include "gemm_common.h"

namespace xl {

// ================================================================== 256x256 ping-pong kernel
// 8 waves (2 x 4), wave tile 128 x 64, one workgroup per CU (128 KiB LDS), two waves per SIMD.  A K tile (64) is staged by
// LDS-DMA as four 16 KiB half-tiles (A0 | B0 | B1 | A1: 64 of every wave's 128 rows / 32 of its 64 columns) into a
// two-deep ring and consumed in two phases of 16 MFMAs (rows A0, then rows A1, against both column halves).
// Phase = { fragment reads + LDS-DMA issue + lgkmcnt(0) + counted vmcnt | barrier | 16 MFMAs | barrier }.  Waves 4-7 run
// one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section while its partner reads fragments and
// issues loads.  vmcnt is never 0 in the steady state: four half-tiles (8 DMA instructions per wave) stay in flight
// across the barriers.  (A four-phase variant with 8 MFMAs per section measured 1.9 us per K tile against this one's
// figure in DESIGN.md: the barrier round trip, ~200 cycles, is the overhead to amortise.)
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void hard_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
template <int V> using ic = std::integral_constant<int, V>;

// Geometry of the two tile shapes (8 waves each):
//   BN = 256: waves 2 (M) x 4 (N), wave tile 128 x 64: A half = 2 row fragments, B in 2 parts of one column fragment
//   BN = 192: waves 4 (M) x 2 (N), wave tile  64 x 96: A half = 1 row fragment,  B in 3 parts of one column fragment
// (N = 768 = 3 x 256 = 4 x 192: 64 row tiles give 192 tiles of 256x256 -- a quarter of the 256 CUs idle for the whole launch --
// or 256 tiles of 256x192; N = 2304 likewise 576 -> 768 = 3 full rounds.)  The A half-tiles are 128 rows x 64 k (16 KiB,
// 2 DMA pieces per wave) in both; a B part is (waves in N) x 32 columns: 128 rows / 16 KiB / 2 pieces or 64 rows / 8 KiB / 1 piece.
// BM = 128 ("duo", BN = 192 only): FOUR waves (2 x 2) with the 64 x 96 wave tile of the 256x192 shape, 80 KiB of LDS and 256
// registers per wave -- TWO workgroups per CU.  Nothing synchronises the two; each is its partner's filler: while one runs its
// prologue / epilogue / hand-over to the next workgroup, the other's K loop has the MFMA pipes (the 8-wave shapes leave them idle
// for ~1/3 of a K = 768 tile's life), and workgroups of different launches (streams) can share a CU.
template <int BN, int BM = 256> struct PPGeo;
template <> struct PPGeo<256, 256> { static constexpr int WR = 2, WC = 4, AF = 2, NB = 2; };
template <> struct PPGeo<192, 256> { static constexpr int WR = 4, WC = 2, AF = 1, NB = 3; };
template <> struct PPGeo<192, 128> { static constexpr int WR = 2, WC = 2, AF = 1, NB = 3; };

template <bool AK, bool BKM, int EPIK, int BN = 256, int BM = 256>
__device__ __forceinline__ void pp_tile(const GemmParams& p, int tm, int tn, int kbeg, int kend, bool first,
                                        int slab_tile = -1, int z = 0, int nz = 1) {
    using G = PPGeo<BN, BM>;
    constexpr int WR = G::WR, WC = G::WC, AF = G::AF, NB = G::NB, NW = WR * WC;
    constexpr int WTM = BM / WR, WTN = BN / WC, HR = AF * 32;          // wave tile; rows of a wave in one A half
    constexpr int BROWS = WC * 32;                                     // columns (tile rows) of one B part
    using TA = OpTile<AK, BM / 2>;
    using TB = OpTile<BKM, BROWS>;
    constexpr int HT = TA::BYTES, BPB = TB::BYTES, PB = BPB / (NW * 1024);   // bytes of an A half / a B part, DMA pieces per wave and part
    constexpr int BUF = 2 * HT + NB * BPB;                             // [A0 | B parts | A1]
    constexpr int NA = HT / (NW * 1024), NBP = NB * PB;                // DMA instructions per wave: A half, all of B
    static_assert(NA == 2 && PB >= 1 && 2 * BUF <= (BM == 256 ? 131072 : 81920) && NW * 16384 <= (BM == 256 ? 131072 : 81920),
                  "ring geometry / epilogue transposition space");
    constexpr int WAIT = 2 * NA + NBP;                                 // steady-state vmcnt (see the phase comment)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];     // [2 buffers][A0 | B parts | A1]
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int grp = NW == 8 ? wave >> 2 : 0;       // waves 4-7 run one barrier behind waves 0-3 (one wave of each group per SIMD)

    f32x16_t acc[2 * AF][NB];
#pragma unroll
    for (int i = 0; i < 2 * AF; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane sources of this wave's 1 KiB pieces of each half-tile / part (BYTE offsets, k0 excluded), and the k
    // coordinate of the lane's 16 bytes inside the K tile (ragged last tile: lanes past kend read zeros instead)
    uint32_t srca[2][NA], srcb[NB][PB];
    int kka[NA], kkb[PB];
#pragma unroll
    for (int pt = 0; pt < NA; ++pt) {
        const int o = (wave * NA + pt) * 1024 + lane * 16;
        int rs, c;
        TA::decode(o, rs, c);
        kka[pt] = AK ? c * 8 : rs;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int lr = AK ? rs : c * 8;                                   // local row (first of 8 when M-major)
            const int gr = m0 + (lr / HR) * WTM + h * HR + (lr % HR);
            srca[h][pt] = 2u * (AK ? (uint32_t)min(gr, p.M - 1) * (uint32_t)p.lda + c * 8
                                   : (uint32_t)rs * (uint32_t)p.lda + min(gr, p.lda - 8));
        }
    }
#pragma unroll
    for (int pt = 0; pt < PB; ++pt) {
        const int o = (wave * PB + pt) * 1024 + lane * 16;
        int rs, c;
        TB::decode(o, rs, c);
        kkb[pt] = BKM ? c * 8 : rs;
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int lr = BKM ? rs : c * 8;
            const int gn = n0 + (lr >> 5) * WTN + h * 32 + (lr & 31);
            srcb[h][pt] = 2u * (BKM ? (uint32_t)min(gn, p.N - 1) * (uint32_t)p.ldb + c * 8
                                    : (uint32_t)rs * (uint32_t)p.ldb + min(gn, p.ldb - 8));
        }
    }
    // LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): the per-lane part of the address is a constant
    // 32-bit byte offset, the K position goes into the scalar offset, and a lane past the end of K is pointed beyond the
    // descriptor's range, where the hardware returns zeros -- two VALU instructions per DMA instead of a 64-bit address
    // computation and a two-register select against a zero page.
    const auto rsrc_of = [](const void* ptr, uint32_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(A, (uint32_t)(((size_t)((AK ? p.M : p.K) - 1) * p.lda + (AK ? p.K : p.M)) * 2));
    const __amdgpu_buffer_rsrc_t rb = rsrc_of(B, (uint32_t)(((size_t)((BKM ? p.N : p.K) - 1) * p.ldb + (BKM ? p.K : p.N)) * 2));
    auto stage_a = [&](auto H, int kt) {
        constexpr int h = decltype(H)::value;
        const int k0 = kbeg + kt * BK;
        const int krem = kend - k0;
        uint8_t* dst = smem + (kt & 1) * BUF + (h == 0 ? 0 : HT + NB * BPB) + wave * (NA * 1024);
        const uint32_t soff = (uint32_t)(AK ? k0 : k0 * p.lda) * 2u;
        if (krem >= BK) {                                 // whole K tile in range (wave-uniform): no per-lane work at all
#pragma unroll
            for (int pt = 0; pt < NA; ++pt)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dst + pt * 1024), 16,
                                                         (int)srca[h][pt], (int)soff, 0, 0);
        } else {
#pragma unroll
            for (int pt = 0; pt < NA; ++pt) {
                uint32_t voff = srca[h][pt];
                if (kka[pt] >= krem) voff = 0x7FFFFFF0u;                      // out of range -> zeros
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dst + pt * 1024), 16,
                                                         (int)voff, (int)soff, 0, 0);
            }
        }
    };
    auto stage_b = [&](int kt) {
        const int k0 = kbeg + kt * BK;
        const int krem = kend - k0;
        const uint32_t soff = (uint32_t)(BKM ? k0 : k0 * p.ldb) * 2u;
        uint8_t* dst0 = smem + (kt & 1) * BUF + HT + wave * (PB * 1024);
        if (krem >= BK) {
#pragma unroll
            for (int h = 0; h < NB; ++h)
#pragma unroll
                for (int pt = 0; pt < PB; ++pt)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dst0 + h * BPB + pt * 1024),
                                                             16, (int)srcb[h][pt], (int)soff, 0, 0);
        } else {
#pragma unroll
            for (int h = 0; h < NB; ++h)
#pragma unroll
                for (int pt = 0; pt < PB; ++pt) {
                    uint32_t voff = srcb[h][pt];
                    if (kkb[pt] >= krem) voff = 0x7FFFFFF0u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dst0 + h * BPB + pt * 1024),
                                                             16, (int)voff, (int)soff, 0, 0);
                }
        }
    };
    bf16x8_t fa[AF][4], fb[NB][4];        // A: row fragments x 4 k-steps of the current A half; B: all parts
    // M-major operands: transpose reads through inline assembly (gemm_common.h TrFrag: no compiler-made vmcnt(0) in front of them)
    [[maybe_unused]] uint32_t a_tr[AF], b_tr = 0;
    if constexpr (!AK) {
#pragma unroll
        for (int i = 0; i < AF; ++i) a_tr[i] = tr_lane_off<BM / 2>(wr * HR + i * 32, lane);
    }
    if constexpr (!BKM) b_tr = tr_lane_off<BROWS>(wc * 32, lane);
    const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem;
    auto read_a = [&](const uint8_t* buf, auto H) {
        constexpr int toff = decltype(H)::value == 0 ? 0 : HT + NB * BPB;
        if constexpr (!AK) {
            const uint32_t base = smem_lds + (uint32_t)(buf - smem);
#pragma unroll
            for (int i = 0; i < AF; ++i) {
                using R = TrFrag<TA::RP, toff>;
                const uint32_t ad = base + a_tr[i];
                fa[i][0] = R::template get<0>(ad); fa[i][1] = R::template get<1>(ad);
                fa[i][2] = R::template get<2>(ad); fa[i][3] = R::template get<3>(ad);
            }
        } else {
            const uint8_t* t = buf + toff;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < AF; ++i) fa[i][s] = TA::template frag<true>(t, wr * HR + i * 32, s, lane);
        }
    };
    auto read_b_part = [&](const uint8_t* buf, auto HH) {
        constexpr int h = decltype(HH)::value;
        if constexpr (!BKM) {
            using R = TrFrag<TB::RP, HT + h * BPB>;
            const uint32_t ad = smem_lds + (uint32_t)(buf - smem) + b_tr;
            fb[h][0] = R::template get<0>(ad); fb[h][1] = R::template get<1>(ad);
            fb[h][2] = R::template get<2>(ad); fb[h][3] = R::template get<3>(ad);
        } else {
            const uint8_t* t = buf + HT + h * BPB;
#pragma unroll
            for (int s = 0; s < 4; ++s) fb[h][s] = TB::template frag<true>(t, wc * 32, s, lane);
        }
    };
    auto read_b = [&](const uint8_t* buf) {
        read_b_part(buf, ic<0>{});
        read_b_part(buf, ic<1>{});
        if constexpr (NB == 3) read_b_part(buf, ic<2>{});
    };
    auto mma2 = [&](auto AH) {            // 16 (12) MFMAs over 4 (3) independent accumulators
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
    // Two phases per K tile: P0 = rows A0 x all of B, P1 = rows A1 x all of B.
    //   P0(t) reads B, A0 of tile t and issues A1(t+1);  P1(t) reads A1 of tile t and issues A0, B of tile t+2.
    // Every half-tile is issued two phases before the phase that waits for it (vmcnt) and three before its first read;
    // a slot is rewritten one phase after its last read, which is safe because the fragment reads are retired
    // (lgkmcnt(0)) BEFORE the barrier that ends the reading section.  After either phase's issue the DMAs that may still
    // be in flight are one A half and one (A half + B): WAIT = 2 NA + NBP instructions.
#ifdef XL_PP_PROFILE      // debug build: cycles of the sections of wave 0 / wave 4 of every workgroup (tools/gemm_trace.py --sections)
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};       // L(P0), L(P1), barrier-1 wait, M, barrier-2 wait, phases
#define XL_T() __builtin_readcyclecounter()
#else
#define XL_T() 0ull
#endif
    auto phase = [&](auto X, auto WAITC, auto ISSUE, int kt) {
        constexpr int x = decltype(X)::value;
        const uint8_t* buf = smem + (kt & 1) * BUF;
        [[maybe_unused]] const unsigned long long t0 = XL_T();
        if constexpr (x == 0) {
            read_b(buf);
            read_a(buf, ic<0>{});
        } else {
            read_a(buf, ic<1>{});
        }
        if constexpr (decltype(ISSUE)::value != 0) {
            if constexpr (x == 0) {
                stage_a(ic<1>{}, kt + 1);
            } else {
                stage_a(ic<0>{}, kt + 2); stage_b(kt + 2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        [[maybe_unused]] const unsigned long long t1 = XL_T();
        wait_vmcnt<decltype(WAITC)::value>();
        [[maybe_unused]] const unsigned long long t1b = XL_T();
        hard_barrier();
        [[maybe_unused]] const unsigned long long t2 = XL_T();
        __builtin_amdgcn_s_setprio(1);
        if constexpr (x == 0) mma2(ic<0>{});
        else mma2(ic<1>{});
        __builtin_amdgcn_s_setprio(0);
        [[maybe_unused]] const unsigned long long t3 = XL_T();
        hard_barrier();
#ifdef XL_PP_PROFILE
        const unsigned long long t4 = XL_T();
        pc[x] += t1 - t0; pc[2] += t2 - t1b; pc[3] += t3 - t2; pc[4] += t4 - t3; pc[5] += t1b - t1;
#endif
    };

    auto stamp = [&](int i) {
        if (p.trace != nullptr && tid == 0) p.trace[(size_t)blockIdx.x * 4 + i] = wall_clock64();
    };
    stamp(0);
    // the epilogue's bias segment is requested before the first DMA: older than every counted load, so the K loop's vmcnt
    // arithmetic is unchanged, and its latency (a full miss after 15 us of streaming operands) is off the epilogue's front
    [[maybe_unused]] const uint64_t dseed = dropout_seed_of<EPIK>(p);
    float bias8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] float bias8b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // BN = 192: third column fragment
    if constexpr (EPIK >= 0) {
        if (n0 + wc * WTN + WTN <= p.N) {
            const bool with_bias = first || slab_tile >= 0;        // slabs: whichever slice arrives last adds the bias
            load_bias8(p, lane, with_bias, n0 + wc * WTN, bias8);
            if constexpr (BN == 192) sub_load_bias8<32>(p, lane, with_bias, n0 + wc * WTN + 64, bias8b);
        }
    }
    const int nkt = (kend - kbeg + BK - 1) / BK;
    // prologue: tile 0 complete, plus A0 | B of tile 1; A0, B of tile 0 must have landed before P0(0)
    stage_a(ic<0>{}, 0); stage_b(0); stage_a(ic<1>{}, 0);
    if (nkt >= 2) { stage_a(ic<0>{}, 1); stage_b(1); wait_vmcnt<WAIT>(); } else { wait_vmcnt<NA>(); }
    hard_barrier();
    stamp(1);
    if (grp == 1) hard_barrier();                 // waves 4-7 run one barrier behind waves 0-3
    for (int kt = 0; kt < nkt - 2; ++kt) {
        phase(ic<0>{}, ic<WAIT>{}, ic<1>{}, kt);
        phase(ic<1>{}, ic<WAIT>{}, ic<1>{}, kt);
    }
    if (nkt >= 2) {
        phase(ic<0>{}, ic<WAIT>{}, ic<1>{}, nkt - 2);
        phase(ic<1>{}, ic<NA>{}, ic<0>{}, nkt - 2);
    }
    phase(ic<0>{}, ic<0>{}, ic<0>{}, nkt - 1);
    phase(ic<1>{}, ic<0>{}, ic<0>{}, nkt - 1);
    if (grp == 0) hard_barrier();
    stamp(2);
#ifdef XL_PP_PROFILE
    if (p.trace != nullptr && lane == 0 && (wave == 0 || wave == 4)) {
        unsigned long long* o = p.trace + 4 * 8192 + ((size_t)blockIdx.x * 2 + (wave >> 2)) * 6;
        for (int i = 0; i < 6; ++i) o[i] = pc[i];
    }
#endif
    if (p.ablate & 4) return;
    // ---- split-K through slabs: only the last arriver of this output tile goes on, with the whole sum in its registers
    const bool slabbed = slab_tile >= 0 && nz > 1;
    if (slabbed) {
        if constexpr (BM == 256) {
            if (!slab_exchange(p, smem, tid, slab_tile, z, nz, acc)) return;
        }
        first = true;
    }
    // ---- epilogue (all fragment reads of the staging LDS are behind the last barrier)
    const int mw = m0 + wr * WTM, nw = n0 + wc * WTN;
    if (p.atomic_out) {
        if constexpr (BN == 256) {
            // one writer per tile (no split, or the slab path's last arriver), interior tile, aligned rows: vector accumulate
            if ((slab_tile >= 0) && p.vec_epi && mw + 128 <= p.M && nw + 64 <= p.N) {
                float* wb = reinterpret_cast<float*>(smem + wave * 16384);
                epilogue_quad_accum(p, wb, lane, mw, nw, !p.overwrite, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
                epilogue_quad_accum(p, wb, lane, mw + 64, nw, !p.overwrite, acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * AF; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) epilogue_atomic_frag(p, lane, first, mw + i * 32, nw + j * 32, acc[i][j]);
        return;
    }
    float* wbuf = reinterpret_cast<float*>(smem + wave * 16384);
    if constexpr (BN == 192) {
        // 64 x 96 wave tile = one 64x64 quad + one 64x32 half quad; the host sends only launches here whose tiles are all
        // interior and take the fast epilogue
        static_assert(EPIK >= 0, "the 256x192 tile has the fast epilogue only");
        QuadOperand op0, op1;
        quad_operand_load<EPIK>(p, lane, mw, nw, op0);
        __builtin_amdgcn_sched_barrier(0);
        quad_to_lds(wbuf, lane, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        sub_operand_load<EPIK, 32>(p, lane, mw, nw + 64, op1);
        __builtin_amdgcn_sched_barrier(0);
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        epilogue_rows_fast<EPIK>(p, wbuf, lane, first, mw, nw, op0, cs, bias8, dseed);
        __builtin_amdgcn_sched_barrier(0);
        half_to_lds(wbuf, lane, acc[0][2], acc[1][2]);
        __builtin_amdgcn_sched_barrier(0);
        sub_rows_fast<EPIK, 32>(p, wbuf, lane, mw, nw + 64, op1, bias8b, dseed);
        if (p.trace != nullptr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(3); }
        return;
    } else {
        if constexpr (EPIK >= 0) {
            if (mw + 128 <= p.M && nw + 64 <= p.N) {
                // the first quad's operand rows are requested before its transpose, the second quad's as soon as the first
                // quad's accumulators are in LDS (their registers are free from then on)
                QuadOperand op0, op1;
                quad_operand_load<EPIK>(p, lane, mw, nw, op0);
                __builtin_amdgcn_sched_barrier(0);
                quad_to_lds(wbuf, lane, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
                __builtin_amdgcn_sched_barrier(0);
                quad_operand_load<EPIK>(p, lane, mw + 64, nw, op1);
                __builtin_amdgcn_sched_barrier(0);
                float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                epilogue_rows_fast<EPIK>(p, wbuf, lane, first, mw, nw, op0, cs, bias8, dseed);
                __builtin_amdgcn_sched_barrier(0);
                quad_to_lds(wbuf, lane, acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
                __builtin_amdgcn_sched_barrier(0);
                epilogue_rows_fast<EPIK>(p, wbuf, lane, first, mw + 64, nw, op1, cs, bias8, dseed);
                if (p.colsum_ws != nullptr) colsum_flush(p, lane, mw >> 7, nw, cs);      // one slab per 128 rows
                if (p.trace != nullptr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(3); }
                return;
            }
        }
        epilogue_quad(p, wbuf, lane, first, mw, nw, acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
        epilogue_quad(p, wbuf, lane, first, mw + 64, nw, acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
        if (p.trace != nullptr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(3); }
    }
}

template <bool AK, bool BKM, int EPIK, int BN>
__global__ __launch_bounds__(256, 2) void gemm_bf16_pp_duo_kernel(GemmParams p) {      // 128 x BN tiles, two workgroups per CU
    int tm, tn, z;
    tile_coords(p, tm, tn, z);
    pp_tile<AK, BKM, EPIK, BN, 128>(p, tm, tn, 0, p.K, true);
}

template <bool AK, bool BKM, int EPIK, int BN>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp_kernel(GemmParams p) {
    int tm, tn, z;
    if (p.tail_tiles > 0) {
        // Tail split: a launch whose last round would hold only a few tiles (264 tiles on 256 CUs: a second round of 8) runs
        // those tiles as `splitk` K slices each, combined through slabs by the last arriver, which also runs the epilogue --
        // they fill the CUs as the first round drains instead of keeping 8 of them busy for another full tile.
        // (workgroups are dispatched in blockIdx order: the whole tiles take the first `full` indices, XCD-aware among
        // themselves, the slices the rest)
        const int full = p.tiles_m * p.tiles_n - p.tail_tiles;
        if ((int)blockIdx.x < full) {
            tile_of(p, linear_block(full), tm, tn);
            pp_tile<AK, BKM, EPIK, BN>(p, tm, tn, 0, p.K, true);
        } else {
            const int r = (int)blockIdx.x - full, tt = r / p.splitk;
            z = r - tt * p.splitk;
            tile_of(p, full + tt, tm, tn);
            const int kbeg = z * p.tail_kper;
            pp_tile<AK, BKM, EPIK, BN>(p, tm, tn, kbeg, min(p.K, kbeg + p.tail_kper), z == 0, tt, z,
                                       (p.K + p.tail_kper - 1) / p.tail_kper);
        }
        return;
    }
    tile_coords(p, tm, tn, z);
    const int kbeg = z * p.kper;
    pp_tile<AK, BKM, EPIK, BN>(p, tm, tn, kbeg, min(p.K, kbeg + p.kper), z == 0,
                               p.slab != nullptr ? tn * p.tiles_m + tm : -1, z, p.splitk);
}

// Two problems in ONE launch ("pair"): C_i = epilogue(A_i B_i^T) for i = 0, 1 with the same N, K, layouts, leading dimensions and
// epilogue kind but their own operands, bias / residual / aux, row count and dropout seed -- the visual (B x 64 rows) and the
// language (packed B x ~13 rows) side of a cross-modality layer's self-attention / FFN sub-blocks, or a visual and a language
// layer of the two stacks (HF:417-449, 516-529: same shapes, different weights).  The language side alone is 39 row tiles of 256:
// as its own launch it occupies 39-156 of the 256 CUs at 0.1 of the MFMA peak next to the other streams' launches; as the tail of
// the visual side's tile list it fills CUs that launch leaves idle anyway (N = 768: 192 tiles + 39 on 256 CUs).  Linear tile
// order: problem 0's tiles, then problem 1's, each in its own XCD-aware (8 row tiles x all columns) order.
template <bool AK, bool BKM, int EPIK>
__global__ __launch_bounds__(512, 1) void gemm_bf16_pp_pair_kernel(PairParams pp) {
    const int L = linear_block();
    const int which = L >= pp.tiles0 ? 1 : 0;
    const GemmParams& p = pp.p[which];
    int tm, tn;
    tile_of(p, L - which * pp.tiles0, tm, tn);
    pp_tile<AK, BKM, EPIK, 256>(p, tm, tn, 0, p.K, true);
}

template <bool AK, bool BKM, int EPIK>
static hipError_t launch_pp_pair_one(const PairParams& pp, int nblk, hipStream_t st) {
    constexpr int lds = 131072;
    hipError_t e = hipSuccess;
    static bool attr = false;
    auto k = gemm_bf16_pp_pair_kernel<AK, BKM, EPIK>;
    if (!attr) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL(k, dim3(nblk), dim3(512), lds, st, pp);
    return e;
}

template <bool AK, bool BKM, int EPIK, int BN = 256, int BM = 256>
static hipError_t launch_pp_one(const GemmParams& p, int nblk, hipStream_t st) {
    hipError_t e = hipSuccess;
    static bool attr = false;
    if constexpr (BM == 128) {
        constexpr int lds = 81920;              // two ring slots of 40 KiB; 2 x 80 KiB = the CU's 160 KiB
        auto k = gemm_bf16_pp_duo_kernel<AK, BKM, EPIK, BN>;
        if (!attr) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
        hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, st, p);
    } else {
        constexpr int lds = 131072;
        auto k = gemm_bf16_pp_kernel<AK, BKM, EPIK, BN>;
        if (!attr) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
        hipLaunchKernelGGL(k, dim3(nblk), dim3(512), lds, st, p);
    }
    return e;
}


// fast-epilogue instances of one layout and tile width (a kind without an instance: hipErrorInvalidValue at 192, the generic
// epilogue at 256)
template <bool AK, bool BKM, int BN, int BM = 256>
static hipError_t launch_pp_layout(const GemmParams& p, int epik, int nblk, hipStream_t st) {
    if constexpr (AK) {
        switch (epik) {
            case XL_EPI_NONE: return launch_pp_one<AK, BKM, XL_EPI_NONE, BN, BM>(p, nblk, st);
            case XL_EPI_GELU: return launch_pp_one<AK, BKM, XL_EPI_GELU, BN, BM>(p, nblk, st);
            case XL_EPI_RESIDUAL: return launch_pp_one<AK, BKM, XL_EPI_RESIDUAL, BN, BM>(p, nblk, st);
            case XL_EPI_DGELU: return launch_pp_one<AK, BKM, XL_EPI_DGELU, BN, BM>(p, nblk, st);
            case XL_EPI_GELU_DG:
                if constexpr (BKM) return launch_pp_one<AK, BKM, XL_EPI_GELU_DG, BN, BM>(p, nblk, st);
                break;
            case XL_EPI_MULAUX:
                if constexpr (!BKM) return launch_pp_one<AK, BKM, XL_EPI_MULAUX, BN, BM>(p, nblk, st);
                break;
            case XL_EPI_ROWMAX:
                if constexpr (BKM && BN == 256) return launch_pp_one<AK, BKM, XL_EPI_ROWMAX, BN>(p, nblk, st);
                break;
            default: break;
        }
    }
    if constexpr (BN == 192) return hipErrorInvalidValue;
    else return launch_pp_one<AK, BKM, -1, BN>(p, nblk, st);
}

// one per translation unit
hipError_t launch_pp_nt256(const GemmParams& p, int epik, int nblk, hipStream_t st);                 // gemm_pp.hip
hipError_t launch_pp_other256(const GemmParams& p, int a_kmajor, int b_kmajor, int epik, int nblk, hipStream_t st);   // gemm_pp_nn.hip
hipError_t launch_pp_192(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st);   // gemm_pp_192.hip
hipError_t launch_pp_duo(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st);   // gemm_pp_duo.hip (128x192, two per CU)

}  // namespace xl
