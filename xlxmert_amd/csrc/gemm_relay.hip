// "Relay" kernel: the K = 768-class contractions with their epilogue hidden INSIDE the workgroup.
//
// The 256x256 ping-pong tile (gemm_pp_kernel.h) spends a K = 768 tile's life as prologue 1.5 us + K loop 17.5 us + epilogue
// 3.4-8.6 us + hand-over 3.5 us (profiles/r05c/trace_epi.txt): a quarter to a third of a CU's time with the matrix pipes idle, and
// every earlier attempt to fill it needed a SECOND workgroup (duo, q tiles) or left the SIMD without a K-loop wave (persistent).
// Here ONE persistent workgroup of eight waves is two groups of four (one wave of each group per SIMD) that trade roles every
// 256 x 128 output tile:
//   * the COMPUTE group runs the tile's whole K loop as one self-pipelined stream per wave -- wave tile 128 x 64 = 8 accumulators
//     (128 registers), the fragments of k-step s+1 (6 x ds_read_b128, 24 registers, double-buffered) requested under the 8 MFMAs of
//     k-step s, no VALU, no LDS-DMA, one barrier per K tile of 32 MFMAs;
//   * the SUPPORT group, whose accumulators hold the tile it computed one slot earlier, (i) issues every LDS-DMA of the compute
//     group's operand stream (12 pieces of 1 KiB per wave and K tile, two K tiles ahead, through a 3 x 48 KiB ring) and (ii) runs its
//     own epilogue in eight chunks of one 32 x 32 accumulator, one chunk per K tile (bias / dropout + residual / GELU + saved
//     derivative / multiply-by-saved-derivative, fp32 transposition through 4 KiB of wave-private LDS, 16-byte stores), then clears
//     its accumulators.  A DMA issue costs the issuing wave 60-185 cycles (MI355X_MICROARCH.md): in a one-wave-per-SIMD design they
//     would sit between that wave's own MFMAs (the 4-wave / 512-register kernel of round 2 lost 14 % to exactly this); here they sit
//     in the partner wave's stream, beside the MFMAs.
// Two accumulator sets per SIMD -- one per wave -- so tile i's epilogue runs under tile i+1's K loop; the matrix pipe of a SIMD is
// always owned by exactly one self-sufficient stream; nothing is handed between workgroups.  The price is the 256 x 128 tile's
// operand traffic: 48 KiB of LDS-DMA per 32 MFMAs and SIMD against 64 KiB per 64 for the 256 x 256 tile (1.5x).
//
// Protocol (g = stage = K tile counted over the workgroup's whole tile list, ring slot g % 3; B_g = the one barrier of stage g):
//   support, between B_{g-1} and B_g:  issue DMA(g+2) | request the operand rows of epilogue chunk c+1 | chunk c's accumulator ->
//       LDS | s_waitcnt vmcnt(12 + younger loads): DMA(g+1) has landed and chunk c's operand rows are here (loads return in
//       order; stores are never counted as allowance, they complete out of order with loads) | chunk c's rows -> math -> stores
//   compute, stage g:  k-steps 0..2 | lgkmcnt(0): every read of slot g % 3 is retired | B_g | first fragments of stage g+1 |
//       k-step 3.   After B_g slot g % 3 may be refilled (DMA(g+3)) and slot (g+1) % 3 may be read.
//   At a role swap the new compute group still owns DMA(g+1) from its last support interval: its first stage waits vmcnt(0)
//   (its epilogue's stores are long done by then) before B_g.
// Every LDS access of this kernel is inline assembly and the epilogue's operand rows arrive by LDS-DMA: next to outstanding LDS-DMA hipcc drains
// vmcnt(0) in front of plain LDS reads and of any ordinary load's first use (cdna_hip_programming.md, glds traps b).
// Results are bit-identical to the ping-pong kernel's: same MFMA, same k order per accumulator, same epilogue arithmetic.
// Restrictions (host: xl_gemm): A K-major, bf16 in / out, M % 256 == N % 256 == K % 64 == 0, K >= 768, fast-epilogue kinds
// NONE / RESIDUAL / GELU_DG (B K-major) and NONE / RESIDUAL / MULAUX (B M-major), no K split, no fused column sums (CS = false).
#include "gemm_pp_kernel.h"

namespace xl {

constexpr int RL_SLOT = 49152;                 // [A rows 0..127 | A rows 128..255 | B 128 columns], 16 KiB each
constexpr int RL_EPI = 3 * RL_SLOT;            // 4 KiB of transposition space per support wave behind the ring
constexpr int RL_LDS = RL_EPI + 4 * 4096;      // = 160 KiB

template <int OFF>
__device__ __forceinline__ bf16x8_t rl_read128(uint32_t addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ f32x4_t rl_read128f(uint32_t addr) {
    f32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void rl_write32(uint32_t addr, float v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t rl_gload16(const void* ptr) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr));
    return v;
}
template <int N>
__device__ __forceinline__ void rl_vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct RlFrags { bf16x8_t a[4], b[2]; };
__device__ __forceinline__ void rl_wait_frags(RlFrags& f) {      // the MFMAs that consume f depend on THIS statement
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]));
}

#ifdef XL_RELAY_PROFILE       // debug build: section cycle counters of waves 0 and 4 of every workgroup (tools/relay_trace.py)
#define RL_T() __builtin_readcyclecounter()
#define RL_ADD(i, d) pc[i] += (d)
#else
#define RL_T() 0ull
#define RL_ADD(i, d) (void)0
#endif

template <int EPI>
constexpr int rl_oploads() { return (EPI == XL_EPI_RESIDUAL || EPI == XL_EPI_MULAUX) ? 2 : 0; }

template <bool BKM, int EPIK>
__global__ __launch_bounds__(512, 1) void gemm_bf16_relay_kernel(GemmParams p) {
    using TA = OpTile<true, 128>;
    using TB = OpTile<BKM, 128>;
    constexpr int LC = rl_oploads<EPIK>();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wr = w4 >> 1, wc = w4 & 1;
    const int nkt = p.K / BK;

    // this workgroup's tiles: a contiguous range of the linear 256 x 128 tile order (two N halves of one 256 x 256 tile of the
    // XCD-aware order are neighbours: the second reads the A panel the first just pulled through this XCD's L2)
    const int ntiles = p.tiles_m * p.tiles_n * 2;
    const int nwg = (int)gridDim.x;
    const int vb = linear_block(nwg);
    const int tq = ntiles / nwg, tr = ntiles - tq * nwg;
    const int t_beg = vb * tq + min(vb, tr);
    const int nslots = tq + (vb < tr ? 1 : 0);
    if (nslots <= 0) return;
    auto tile_xy = [&](int L, int& m0, int& n0) {
        int tm, tn;
        tile_of(p, L >> 1, tm, tn);
        m0 = __builtin_amdgcn_readfirstlane(tm * 256);
        n0 = __builtin_amdgcn_readfirstlane(tn * 256 + (L & 1) * 128);
    };

    // ------------------------------------------------------------------ loader side
    const auto rsrc_of = [](const void* ptr, uint32_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(p.A, (uint32_t)(((size_t)(p.M - 1) * p.lda + p.K) * 2));
    const __amdgpu_buffer_rsrc_t rb = rsrc_of(p.B, (uint32_t)(((size_t)((BKM ? p.N : p.K) - 1) * p.ldb + (BKM ? p.K : p.N)) * 2));
    // per-lane byte offsets of the lane's 16 bytes inside a 1 KiB piece: K-major pieces are 8 rows x 128 B whose chunk swizzle
    // depends on the piece's parity; M-major pieces are 4 k-rows x 256 B (one pattern).  The piece's first row goes into the
    // scalar offset.
    uint32_t voa[2], vob[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        int rs, c;
        TA::decode(par * 1024 + lane * 16, rs, c);
        voa[par] = 2u * ((uint32_t)(rs - par * 8) * (uint32_t)p.lda + c * 8);
        TB::decode(par * 1024 + lane * 16, rs, c);
        vob[par] = BKM ? 2u * ((uint32_t)(rs - par * 8) * (uint32_t)p.ldb + c * 8)
                       : 2u * ((uint32_t)(rs - par * 4) * (uint32_t)p.ldb + c * 8);
    }
    const uint32_t a_wave = __builtin_amdgcn_readfirstlane((uint32_t)((w4 >> 1) * 128 + (w4 & 1) * 64) * (uint32_t)p.lda * 2u);
    const uint32_t a_step = __builtin_amdgcn_readfirstlane(8u * (uint32_t)p.lda * 2u);
    const uint32_t b_wave = __builtin_amdgcn_readfirstlane((uint32_t)(w4 * (BKM ? 32 : 16)) * (uint32_t)p.ldb * 2u);
    const uint32_t b_step = __builtin_amdgcn_readfirstlane((BKM ? 8u : 4u) * (uint32_t)p.ldb * 2u);
    const uint32_t b_kstep = __builtin_amdgcn_readfirstlane(BKM ? 128u : 64u * (uint32_t)p.ldb * 2u);     // bytes per K tile
    // loader cursor: next stage to issue
    int ld_left = 0, ld_kt = 0;
    uint32_t ld_ring = 0, ld_a = 0, ld_b = 0, ld_a2 = 0, ld_b2 = 0;
    auto bases = [&](int L, uint32_t& ab, uint32_t& bb) {
        int m0, n0;
        tile_xy(L, m0, n0);
        ab = __builtin_amdgcn_readfirstlane((uint32_t)m0 * (uint32_t)p.lda * 2u);
        bb = __builtin_amdgcn_readfirstlane(BKM ? (uint32_t)n0 * (uint32_t)p.ldb * 2u : (uint32_t)n0 * 2u);
    };
    [[maybe_unused]] unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // compute: total, barrier wait, stages | support: total, DMA issue, vmcnt wait, chunk, barrier wait
    const int abl = p.ablate;             // timing probes only (XL_GEMM_ABLATE): 16 no epilogue chunks, 32 no support waits, 64 no DMA, 128 no MFMA
    // one stage = 12 pieces of 1 KiB per wave (8 of A, 4 of B), issued a few at a time between the epilogue's LDS round trips: a
    // DMA issue occupies the wave for ~55-70 cycles (the CU's address path takes 16 per KiB and four waves share it) and an LDS
    // write -> read -> use chain of the epilogue waits ~130 cycles per hop -- each hides in the other
    auto pieces = [&](auto LO, auto HI, bool issued) {
        constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
        if (!issued || (abl & 64)) return;
        const uint32_t sa = ld_a + a_wave + (uint32_t)ld_kt * 128u;
        const uint32_t sb = ld_b + b_wave + (uint32_t)ld_kt * b_kstep;
        uint8_t* dst = smem + ld_ring + w4 * 8192;
        uint8_t* dstb = smem + ld_ring + 32768 + w4 * 4096;
#pragma unroll
        for (int i = lo; i < hi; ++i) {
            if (i < 8)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                         (int)voa[i & 1], (int)(sa + (uint32_t)i * a_step), 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(dstb + (i - 8) * 1024), 16,
                                                         (int)vob[BKM ? (i & 1) : 0], (int)(sb + (uint32_t)(i - 8) * b_step), 0, 0);
        }
    };
    auto advance = [&](bool issued) {
        if (!issued) return;
        ld_ring = ld_ring == 2u * RL_SLOT ? 0u : ld_ring + RL_SLOT;
        --ld_left;
        if (++ld_kt == nkt) { ld_kt = 0; ld_a = ld_a2; ld_b = ld_b2; }
    };
    auto issue_stage = [&]() -> bool {
        const bool issued = ld_left > 0;
        pieces(ic<0>{}, ic<12>{}, issued);
        advance(issued);
        return issued;
    };

    // ------------------------------------------------------------------ compute side
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // per-lane LDS byte offsets (ring slot excluded) of the s = 0..3 fragments: rows i*32 / j*32 go into the immediate
    uint32_t fa_off[4], fb_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        fa_off[s] = smem_lds + (uint32_t)(wr * 16384 + TA::encode(lane & 31, s * 2 + (lane >> 5)));
        if constexpr (BKM) fb_off[s] = smem_lds + (uint32_t)(32768 + TB::encode(wc * 64 + (lane & 31), s * 2 + (lane >> 5)));
        else fb_off[s] = 0;
    }
    [[maybe_unused]] uint32_t btr[2] = {0, 0};
    if constexpr (!BKM) {
        btr[0] = smem_lds + tr_lane_off<128>(wc * 64, lane);
        btr[1] = smem_lds + tr_lane_off<128>(wc * 64 + 32, lane);
    }
    auto read_k = [&](auto S, uint32_t ring, RlFrags& f) {
        constexpr int s = decltype(S)::value;
        const uint32_t aa = fa_off[s] + ring;
        f.a[0] = rl_read128<0>(aa); f.a[1] = rl_read128<4096>(aa); f.a[2] = rl_read128<8192>(aa); f.a[3] = rl_read128<12288>(aa);
        if constexpr (BKM) {
            const uint32_t bb = fb_off[s] + ring;
            f.b[0] = rl_read128<0>(bb); f.b[1] = rl_read128<4096>(bb);
        } else {
            using R = TrFrag<TB::RP, 32768>;
            f.b[0] = R::template get<s>(btr[0] + ring);
            f.b[1] = R::template get<s>(btr[1] + ring);
        }
    };
    auto mfma8 = [&](const RlFrags& f) {
        if (abl & 128) return;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf16_t, f.a[i]), __builtin_bit_cast(v8bf16_t, f.b[j]),
                                                                    acc[i][j], 0, 0, 0);
    };
    auto compute_slot = [&](int j) {
        uint32_t ring = (uint32_t)((j * nkt) % 3) * RL_SLOT;
        RlFrags f0, f1;
        if (!(abl & 256)) __builtin_amdgcn_s_setprio(2);
        [[maybe_unused]] const unsigned long long tc0 = RL_T();
        read_k(ic<0>{}, ring, f0);
        for (int kt = 0; kt < nkt; ++kt) {
            const uint32_t nring = ring == 2u * RL_SLOT ? 0u : ring + RL_SLOT;
            rl_wait_frags(f0); read_k(ic<1>{}, ring, f1); mfma8(f0);
            __builtin_amdgcn_sched_barrier(0);            // (the MFMAs of a k-step stay in front of the next k-step's wait)
            rl_wait_frags(f1); read_k(ic<2>{}, ring, f0); mfma8(f1);
            __builtin_amdgcn_sched_barrier(0);
            rl_wait_frags(f0); read_k(ic<3>{}, ring, f1); mfma8(f0);
            __builtin_amdgcn_sched_barrier(0);
            rl_wait_frags(f1);                            // lgkmcnt(0): every read of this slot is retired
            if (kt == 0) rl_vmwait<0>();                  // DMA(g+1), issued in this wave's last support interval
            [[maybe_unused]] const unsigned long long tb0 = RL_T();
            hard_barrier();                               // B_g
            RL_ADD(1, RL_T() - tb0); RL_ADD(2, 1);
            if (kt + 1 < nkt) read_k(ic<0>{}, nring, f0);
            mfma8(f1);
            __builtin_amdgcn_sched_barrier(0);
            ring = nring;
        }
        __builtin_amdgcn_s_setprio(0);
        RL_ADD(0, RL_T() - tc0);
    };

    // ------------------------------------------------------------------ support side: epilogue of the previous tile + loader
    // Wave-private 4 KiB behind the ring: [0, 2 KiB) half a 32 x 32 accumulator as fp32 rows (16 rows x 128 B), [2 KiB, 4 KiB) the
    // chunk's operand rows (residual / saved derivative: 32 rows x 64 B), which arrive by LDS-DMA like the operand stream -- a
    // VGPR-destination load next to outstanding LDS-DMA is either drained by hipcc (plain load: vmcnt(0) at its first use) or, as
    // inline assembly, exposed to register copies the compiler places between the load and the wait it cannot see.  The two bias
    // values of a lane (accumulator layout: one column per lane and column fragment) are the only such loads left.
    const uint64_t dseed = dropout_seed_of<EPIK>(p);
    const bool drop = p.p_drop > 0.0f;
    const uint32_t wb = smem_lds + RL_EPI + w4 * 4096;
    const uint32_t wb_w = wb + (uint32_t)((4 * (lane >> 5)) * 128 + (lane & 31) * 4);            // accumulator layout -> image
    const uint32_t wb_r = wb + (uint32_t)((lane >> 2) * 128 + (lane & 3) * 32);                  // image -> row (lane>>2), 8 columns
    const uint32_t wb_o = wb + 2048u + (uint32_t)lane * 16u;                                     // operand rows, lane-linear
    uint8_t* const op_lds = smem + RL_EPI + w4 * 4096 + 2048;
    uint32_t braw[2] = {0, 0};               // bias of this lane's column in the two column fragments (raw bits until waited for)
    float bcol[2] = {0.f, 0.f};
    int em0 = 0, en0 = 0;                    // origin of the tile whose accumulators this group holds
    auto load_bias = [&]() {
        // unconditional (a conditional definition makes the compiler merge it with a zero through a register copy -- of a value
        // that has not arrived yet); without a bias the loads read operand A and the values are dropped after the wait
        const float* b = p.bias != nullptr ? p.bias + en0 + wc * 64 + (lane & 31) : reinterpret_cast<const float*>(p.A) + lane;
        asm volatile("global_load_dword %0, %1, off" : "=v"(braw[0]) : "v"(b));
        asm volatile("global_load_dword %0, %1, off offset:128" : "=v"(braw[1]) : "v"(b));
    };
    auto bias_arrived = [&]() {
        asm volatile("" : "+v"(braw[0]), "+v"(braw[1]));
        bcol[0] = p.bias != nullptr ? __uint_as_float(braw[0]) : 0.f;
        bcol[1] = p.bias != nullptr ? __uint_as_float(braw[1]) : 0.f;
    };
    // Epilogue UNIT u = 0..15: half h = u & 1 (rows 16h .. 16h+15 = accumulator registers 8h .. 8h+7) of chunk c = u >> 1 (accumulator
    // acc[c >> 1][c & 1]).  Its operand rows (16 rows x 64 B = one LDS-DMA piece) land in the upper 2 KiB at h * 1 KiB.
    auto load_op = [&](auto U) {
        constexpr int u = decltype(U)::value, c = u >> 1, h = u & 1;
        if constexpr (LC > 0) {
            const bf16_t* src = reinterpret_cast<const bf16_t*>(EPIK == XL_EPI_RESIDUAL ? p.residual : p.aux);
            const int ld = EPIK == XL_EPI_RESIDUAL ? p.ldr : p.ldx;
            const bf16_t* s0 = src + (size_t)(em0 + wr * 128 + (c >> 1) * 32 + h * 16 + (lane >> 2)) * ld + en0 + wc * 64 + (c & 1) * 32 + (lane & 3) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s0,
                                             (__attribute__((address_space(3))) void*)(op_lds + h * 1024), 16, 0, 0);
        }
    };
    // rows of unit u in registers -> the epilogue kind's arithmetic on 8 consecutive columns per lane -> one 16-byte store (two with a
    // saved derivative); lane -> row lane / 4 of the unit, columns (lane & 3) * 8 ..
    auto unit_out = [&](auto U, const f32x4_t& q0, const f32x4_t& q1, [[maybe_unused]] const u32x4_t& o) {
        constexpr int u = decltype(U)::value, c = u >> 1, h = u & 1, j = c & 1;
        const size_t m = (size_t)(em0 + wr * 128 + (c >> 1) * 32 + h * 16 + (lane >> 2));
        const int n = en0 + wc * 64 + j * 32 + (lane & 3) * 8;
        float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
        if constexpr (EPIK == XL_EPI_RESIDUAL) {
            float rv[8];
            unpack8(make_uint4(o.x, o.y, o.z, o.w), rv);
            if (drop) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dropout_scale(dseed, (uint32_t)m, (uint32_t)(n + e), p.p_drop, p.inv_keep);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rv[e];
        } else if constexpr (EPIK == XL_EPI_GELU_DG) {
            float gv[8];
            gelu_fast8_dg(v, gv);
            stvec(reinterpret_cast<bf16_t*>(p.aux) + m * p.ldx + n, gv);
        } else if constexpr (EPIK == XL_EPI_MULAUX) {
            float av[8];
            unpack8(make_uint4(o.x, o.y, o.z, o.w), av);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= av[e];
        }
        uint4 t;
        t.x = pack2bf(v[0], v[1]); t.y = pack2bf(v[2], v[3]); t.z = pack2bf(v[4], v[5]); t.w = pack2bf(v[6], v[7]);
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + m * p.ldc + n) = t;
    };
    // alpha * acc + bias in the accumulator layout -> fp32 rows in LDS at byte offset `base` of the wave's image
    auto unit_to_lds = [&](auto U, auto BASE) {
        constexpr int u = decltype(U)::value, c = u >> 1, h = u & 1, j = c & 1, base = decltype(BASE)::value;
        const f32x16_t& a = acc[c >> 1][j];
        const float bj = bcol[j];
        rl_write32<base + 0>(wb_w, a[8 * h + 0] * p.alpha + bj); rl_write32<base + 128>(wb_w, a[8 * h + 1] * p.alpha + bj);
        rl_write32<base + 256>(wb_w, a[8 * h + 2] * p.alpha + bj); rl_write32<base + 384>(wb_w, a[8 * h + 3] * p.alpha + bj);
        rl_write32<base + 1024>(wb_w, a[8 * h + 4] * p.alpha + bj); rl_write32<base + 1152>(wb_w, a[8 * h + 5] * p.alpha + bj);
        rl_write32<base + 1280>(wb_w, a[8 * h + 6] * p.alpha + bj); rl_write32<base + 1408>(wb_w, a[8 * h + 7] * p.alpha + bj);
    };
    // One or two units with the stage's LDS-DMA pieces [6, 12) dealt into their LDS round trips.  Kinds with operand rows: one unit at
    // a time through the lower 2 KiB (operand rows in the upper 2 KiB); kinds without: two units side by side in all 4 KiB.  The caller
    // has issued pieces [0, 6) and waited vmcnt(6): DMA(g+1), these units' operand rows and the bias have arrived.
    auto units = [&](auto U0, auto NU, bool issued) {
        constexpr int u0 = decltype(U0)::value, nu = decltype(NU)::value, u1 = u0 + (nu == 2 ? 1 : 0);
        const u32x4_t none = {0, 0, 0, 0};
        unit_to_lds(U0, ic<0>{});
        if constexpr (nu == 2 && LC == 0) unit_to_lds(ic<u1>{}, ic<2048>{});
        pieces(ic<6>{}, ic<9>{}, issued);
        f32x4_t q0 = rl_read128f<0>(wb_r), q1 = rl_read128f<16>(wb_r);
        [[maybe_unused]] f32x4_t r0, r1;
        [[maybe_unused]] u32x4_t o = none, o1 = none;
        if constexpr (LC > 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o) : "v"(wb_o), "n"((u0 & 1) * 1024));
        if constexpr (nu == 2 && LC == 0) { r0 = rl_read128f<2048>(wb_r); r1 = rl_read128f<2064>(wb_r); }
        pieces(ic<9>{}, ic<12>{}, issued);
        if constexpr (LC > 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(o));
        else if constexpr (nu == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(r0), "+v"(r1));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1));
        if constexpr (nu == 2 && LC > 0) {
            unit_to_lds(ic<u1>{}, ic<0>{});            // (the first unit's reads of the image are retired)
            r0 = rl_read128f<0>(wb_r); r1 = rl_read128f<16>(wb_r);
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(o1) : "v"(wb_o), "n"((u1 & 1) * 1024));
        }
        unit_out(U0, q0, q1, o);
        if constexpr (nu == 2) {
            if constexpr (LC > 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(o1));
            unit_out(ic<u1>{}, r0, r1, o1);
        }
    };
    // Schedule of the 16 units over a slot's intervals (K tiles): interval 0 requests the bias and the first operand rows; intervals
    // 1-5 take two units each, 6-11 one each -- a support wave's VMEM issue (12 KiB of LDS-DMA per interval + the units' stores and
    // operand rows, ~64 cycles per KiB and wave) and its VALU / LDS work then stay below the compute group's K tile in EVERY interval;
    // bunched as one chunk per interval over eight intervals the support group was the slower of the two.
    // VMEM order of a wave (oldest first): ... DMA(g+1) x 12 | stores of the previous interval's units | their successors' operand DMA ||
    // this interval: DMA(g+2) pieces 0-5 | wait vmcnt(6) = everything older than those six is complete (loads return in order; stores
    // only ever make the wait stricter, and the youngest stores are a whole interval old here) -> DMA(g+1) has landed, this interval's
    // operand rows and the bias are here | pieces 6-11 between the units' LDS hops | the units' stores | next operand DMA.
    auto support_iv = [&](auto KT, auto HP) {
        constexpr int kt = decltype(KT)::value;          // 0..11; 12 = any later interval
        constexpr bool have_prev = decltype(HP)::value != 0;
        constexpr int nu = !have_prev ? 0 : (kt >= 1 && kt <= 5) ? 2 : (kt >= 6 && kt <= 11) ? 1 : 0;
        constexpr int u0 = nu == 2 ? 2 * (kt - 1) : nu == 1 ? 10 + (kt - 6) : 0;
        constexpr int nnext = !have_prev ? 0 : (kt <= 4) ? 2 : (kt >= 5 && kt <= 10) ? 1 : 0;       // units of interval kt + 1
        constexpr int unext = kt <= 4 ? 2 * kt : 10 + (kt - 5);
        [[maybe_unused]] const unsigned long long ts0 = RL_T();
        const bool issued = ld_left > 0;
        if constexpr (nu > 0) pieces(ic<0>{}, ic<6>{}, issued);
        else pieces(ic<0>{}, ic<12>{}, issued);
        [[maybe_unused]] const unsigned long long ts1 = RL_T();
        if constexpr (have_prev && kt == 0) {
            load_bias();         // nothing to confirm (DMA(g+1) belongs to the other group), nothing to use yet
        } else if (!(abl & 32)) {
            if constexpr (nu > 0) { if (issued) rl_vmwait<6>(); else rl_vmwait<0>(); }
            else { if (issued) rl_vmwait<12>(); else rl_vmwait<0>(); }
        }
        [[maybe_unused]] const unsigned long long ts2 = RL_T();
        if constexpr (have_prev) {
            if constexpr (kt == 1) bias_arrived();
            if constexpr (nu > 0) {
                if (!(abl & 16)) units(ic<u0>{}, ic<nu>{}, issued);
                else pieces(ic<6>{}, ic<12>{}, issued);
            }
            if constexpr (nnext >= 1) load_op(ic<(nnext >= 1 ? unext : 0)>{});
            if constexpr (nnext == 2) load_op(ic<(nnext == 2 ? unext + 1 : 0)>{});
            if constexpr (kt == 11) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
        }
        advance(issued);
        [[maybe_unused]] const unsigned long long ts3 = RL_T();
        hard_barrier();                                  // B_g
        RL_ADD(4, ts1 - ts0); RL_ADD(5, ts2 - ts1); RL_ADD(6, ts3 - ts2); RL_ADD(7, RL_T() - ts3); RL_ADD(3, RL_T() - ts0);
    };
    auto support_slot = [&](int j, auto HP) {
        constexpr bool have_prev = decltype(HP)::value != 0;
        if constexpr (have_prev) tile_xy(t_beg + j - 1, em0, en0);
        // cursor: stage G0 + 2 = (tile j, K tile 2)
        ld_kt = 2;
        ld_ring = (uint32_t)((j * nkt + 2) % 3) * RL_SLOT;
        ld_left = (nslots - j) * nkt - 2;
        bases(t_beg + j, ld_a, ld_b);
        if (j + 1 < nslots) bases(t_beg + j + 1, ld_a2, ld_b2);
        support_iv(ic<0>{}, HP); support_iv(ic<1>{}, HP); support_iv(ic<2>{}, HP);
        support_iv(ic<3>{}, HP); support_iv(ic<4>{}, HP); support_iv(ic<5>{}, HP);
        support_iv(ic<6>{}, HP); support_iv(ic<7>{}, HP); support_iv(ic<8>{}, HP);
        support_iv(ic<9>{}, HP); support_iv(ic<10>{}, HP); support_iv(ic<11>{}, HP);
        for (int kt = 12; kt < nkt; ++kt) support_iv(ic<12>{}, HP);
    };

    // ------------------------------------------------------------------ prologue: stages 0 and 1 by slot 0's support group
    if (grp == 1) {
        ld_kt = 0; ld_ring = 0; ld_left = nslots * nkt;
        bases(t_beg, ld_a, ld_b);
        ld_a2 = ld_a; ld_b2 = ld_b;
        issue_stage(); issue_stage();
        rl_vmwait<12>();                                 // stage 0 has landed
    }
    hard_barrier();
    if (grp == 0) compute_slot(0);
    else support_slot(0, ic<0>{});
    for (int j = 1; j < nslots; ++j) {
        if ((j & 1) == grp) compute_slot(j);
        else support_slot(j, ic<1>{});
    }
#ifdef XL_RELAY_PROFILE
    if (p.trace != nullptr && lane == 0 && w4 == 0) {
        unsigned long long* o = p.trace + ((size_t)blockIdx.x * 2 + grp) * 8;
        for (int i = 0; i < 8; ++i) o[i] = pc[i];
    }
#endif
    // ------------------------------------------------------------------ the last tile's epilogue (nobody to hide it under)
    if (((nslots - 1) & 1) == grp) {
        tile_xy(t_beg + nslots - 1, em0, en0);
        load_bias();
        load_op(ic<0>{});
        rl_vmwait<0>();
        bias_arrived();
        auto drain = [&](auto U) {
            constexpr int u = decltype(U)::value;
            if constexpr (u < 15) load_op(ic<(u < 15 ? u + 1 : 0)>{});
            units(U, ic<1>{}, false);
            rl_vmwait<0>();
        };
        drain(ic<0>{}); drain(ic<1>{}); drain(ic<2>{}); drain(ic<3>{}); drain(ic<4>{}); drain(ic<5>{}); drain(ic<6>{}); drain(ic<7>{});
        drain(ic<8>{}); drain(ic<9>{}); drain(ic<10>{}); drain(ic<11>{}); drain(ic<12>{}); drain(ic<13>{}); drain(ic<14>{}); drain(ic<15>{});
    }
}

template <bool BKM, int EPIK>
static hipError_t launch_relay_one(const GemmParams& p, int nblk, hipStream_t st) {
    hipError_t e = hipSuccess;
    auto k = gemm_bf16_relay_kernel<BKM, EPIK>;
    static bool attr = false;
    if (!attr) { e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, RL_LDS); attr = true; }
    hipLaunchKernelGGL(k, dim3(nblk), dim3(512), RL_LDS, st, p);
    return e;
}

bool relay_has_instance(int b_kmajor, int epik) {
    if (b_kmajor) return epik == XL_EPI_NONE || epik == XL_EPI_RESIDUAL || epik == XL_EPI_GELU_DG;
    return epik == XL_EPI_NONE || epik == XL_EPI_RESIDUAL || epik == XL_EPI_MULAUX;
}

// p.tiles_m x p.tiles_n = the grid of 256 x 256 tiles; nblk persistent workgroups share its 2 x as many 256 x 128 tiles
hipError_t launch_relay(const GemmParams& p, int b_kmajor, int epik, int nblk, hipStream_t st) {
    if (b_kmajor) {
        switch (epik) {
            case XL_EPI_NONE: return launch_relay_one<true, XL_EPI_NONE>(p, nblk, st);
            case XL_EPI_RESIDUAL: return launch_relay_one<true, XL_EPI_RESIDUAL>(p, nblk, st);
            case XL_EPI_GELU_DG: return launch_relay_one<true, XL_EPI_GELU_DG>(p, nblk, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (epik) {
        case XL_EPI_NONE: return launch_relay_one<false, XL_EPI_NONE>(p, nblk, st);
        case XL_EPI_RESIDUAL: return launch_relay_one<false, XL_EPI_RESIDUAL>(p, nblk, st);
        case XL_EPI_MULAUX: return launch_relay_one<false, XL_EPI_MULAUX>(p, nblk, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace xl
