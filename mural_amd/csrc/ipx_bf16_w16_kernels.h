// bf16 scan, 16-wave variant of the 256 x 256 tile (large batches). Why it exists: measured on gfx950, a wave
// gets about one 1-KiB LDS-DMA copy piece through per ~230 ns however deep its ring is, so the fill rate of a
// CU grows with the number of waves that copy. scan_bf16_kernel<4> has 8 waves with 4-6 pieces each per
// 32-48 KiB stage and spends more time on its copies than on its MFMAs. Here 16 waves (1024 threads, 4 per
// SIMD, <= 128 registers each) take a 64 x 64 corner of the tile each (acc 2 x 2) and 2-3 pieces per stage;
// with four waves per SIMD the LDS read latency is hidden by the other waves, so the loop is plain: read an
// operand set, wait, multiply. Same packed layouts, same streaming top-k machinery, same results.
// Included by ipx.hip after ipx_bf16_kernels.h and ipx_f32_small_kernels.h.
#pragma once

namespace ipx {

// Entries of a selection's side list per selecting wave (compact_query_select; a wave's scratch is 4 bytes per queue
// slot + this list).
constexpr int W16_SEL_AUX = SEL_AUX;

template <int QP, int NB = 0> struct WCfg {
    static constexpr int THREADS = 1024;
    static constexpr int PART_B = 256 * 64;                  // one part of the query stage: 16 KiB
    static constexpr int STAGE = HSTAGE_A + PART_B * QP;     // 32 / 48 KiB
    static constexpr int NBUF = NB ? NB : (QP == 1 ? 4 : 3); // ring: 128 / 144 KiB unless a shallower one is asked for
    static constexpr int LOADS = 1 + QP;                     // copy pieces per wave per stage
    static constexpr int RING = NBUF * STAGE;
};

// The copy pieces of one stage, issued from asm in the scalar-base form: M0 = the piece's LDS address, base in scalar
// registers, the lane's 16 bytes as a 32-bit offset — no per-lane 64-bit address arithmetic in the stage loop (round 3:
// on the fp32 scan the same change was worth 1.5 %; a VALU instruction in an MFMA-bound loop is not free).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 on a clobber list (the copy pieces set it themselves)
template <int QP, bool NT = false>
struct WPieces {
    const char* ga;
    const char* gb;
    uint32_t la;
    uint32_t lane16;
    static __device__ __forceinline__ const char* uniform_ptr(const char* ptr) {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    __device__ __forceinline__ WPieces(const ScanParams& p, char* buf, int64_t tile, int qb, int ks, int wave, int lane) {
        ga = uniform_ptr((const char*)p.Xp + ((size_t)tile * p.nks + ks) * HSTAGE_A + wave * 1024);
        gb = uniform_ptr((const char*)p.Qp + ((size_t)qb * p.nks + ks) * (WCfg<QP>::PART_B * QP) + wave * 1024);
        la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)buf + wave * 1024);
        lane16 = (uint32_t)lane * 16;
    }
    // from the wave's piece addresses of the stage (kept as running pointers by the scan loop)
    __device__ __forceinline__ WPieces(const char* ga_, const char* gb_, char* buf, int wave, int lane) {
        ga = ga_;
        gb = gb_;
        la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)buf + wave * 1024);
        lane16 = (uint32_t)lane * 16;
    }
    template <bool NT_>
    static __device__ __forceinline__ void piece(const char* g, uint32_t l, uint32_t voff) {
        if constexpr (NT_)
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 nt" :: "s"(l), "v"(voff), "s"(g) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(voff), "s"(g) : "memory", "m0");
    }
    __device__ __forceinline__ void issue_a() const { piece<NT>(ga, la, lane16); }
    __device__ __forceinline__ void issue_b() const {
        piece<false>(gb, la + HSTAGE_A, lane16);
        if constexpr (QP == 2) piece<false>(gb + WCfg<QP>::PART_B, la + HSTAGE_A + WCfg<QP>::PART_B, lane16);
    }
    __device__ __forceinline__ void issue_all() const { issue_a(); issue_b(); }
};
#pragma clang diagnostic pop

__device__ __forceinline__ void w16_wait4(f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) :: "memory");
}
__device__ __forceinline__ void w16_wait2(f32x4& b0, f32x4& b1) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1) :: "memory");
}
#define IPX_W16_MFMA4(A0_, A1_, B0_, B1_)                                                                             \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A0_), as_bf16x8(B0_), acc[0][0], 0, 0, 0);          \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A0_), as_bf16x8(B1_), acc[0][1], 0, 0, 0);          \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A1_), as_bf16x8(B0_), acc[1][0], 0, 0, 0);          \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A1_), as_bf16x8(B1_), acc[1][1], 0, 0, 0)

// ORD: issue order behind the per-stage barrier: 0 = copy pieces, then the four MFMAs; 1 = MFMAs, then pieces;
//      2 = two MFMAs, corpus piece, two MFMAs, query piece(s); 3 = as 1 with s_setprio 1 around the MFMAs.
// NT : corpus pieces carry the non-temporal hint (they do not displace the query blocks from L2).
// NB : stage-ring depth (0 = the default of WCfg).
template <int QP, bool S16, int ORD = 0, bool NT = false, int NB = 0>
__global__ void __launch_bounds__(1024)
scan_bf16_w16_kernel(const ScanParams p, const int nsortwaves) {
    static_assert(!S16 || QP == 1, "the 16x16x32 variant exists for one query part");
    static_assert(!S16 || ORD == 0, "issue-order variants exist for the 32x32x16 form");
    using C = WCfg<QP, NB>;
    constexpr int BQ_ = 256, NBUF = C::NBUF, STAGE = C::STAGE, LOADS = C::LOADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the state words sit behind whichever is largest: the ring, the final sorts' scratch, the selections' scratch
    // (the host sizes the allocation by the same rule)
    // (the queue capacity is a multiple of 512 here, not a power of two: the final sort rounds it up)
    int sort_cap = 512;
    while (sort_cap < p.cap) sort_cap <<= 1;
    int stage_region = C::RING;
    if (nsortwaves * sort_cap * 8 > stage_region) stage_region = nsortwaves * sort_cap * 8;
    if (p.nselect * (p.cap + W16_SEL_AUX) * 4 > stage_region) stage_region = p.nselect * (p.cap + W16_SEL_AUX) * 4;
    int* cnts = (int*)(smem + stage_region);
    float* taus = (float*)(smem + stage_region + BQ_ * 4);
    int* flags = (int*)(smem + stage_region + 2 * BQ_ * 4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wave & 3, wq = wave >> 2;             // 4 x 4 waves: 64 docs x 64 queries each
    const int h = lane >> 5, l31 = lane & 31;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int nqb_ = (int)gridDim.x / p.S;
    int qb, split;
    decode_block(p.map, p.S, nqb_, logical, qb, split);

    if (tid < BQ_) {
        cnts[tid] = 0;
        taus[tid] = ((int64_t)qb * BQ_ + tid < p.nq) ? -FLT_MAX : INFINITY;
    }
    if (tid < 2) flags[tid] = 0;
    uint32_t* mypub = p.S > 1 ? p.pub + ((size_t)qb * p.S + split) * BQ_ : nullptr;
    const int fill_limit = p.cap - 2 * HBD;

    uint64_t* myqueues = p.cand + (size_t)blockIdx.x * BQ_ * p.cap;
    uint64_t* scratch = (uint64_t*)smem + (size_t)wave * sort_cap;
    const int nselect = p.nselect > 0 ? p.nselect : nsortwaves;

    const int sw = (lane >> 2) & 3;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_off = (wd * 64 + l31) * 64;
    const uint32_t b_off = HSTAGE_A + (wq * 64 + l31) * 64;
    const uint32_t pc0 = ((0 + h) ^ sw) * 16, pc1 = ((2 + h) ^ sw) * 16;

    // runs of R tiles dealt round-robin (split s: runs s, s + S, ...); a walker is a position in that
    // sequence of stages: w0 = the stage being computed (ring slot `slot`), wi = the next stage to copy
    const int R = p.run_tiles;
    struct Walker { int t, ks, end, run; bool ok; };
    auto advance = [&](Walker& w) {
        if (!w.ok) return;
        if (w.ks + 1 < p.nks) { ++w.ks; return; }
        if (w.t + 1 < w.end) { ++w.t; w.ks = 0; return; }
        const int64_t next = (int64_t)(w.run + p.S) * R;
        if (next < p.ntiles) {
            w.run += p.S;
            w.t = (int)next;
            w.end = w.t + R < p.ntiles ? w.t + R : p.ntiles;
            w.ks = 0;
            return;
        }
        w.ok = false;          // stays on its last stage: the dummy copies go to a slot nobody reads
    };
    Walker w0{split * R, 0, (split + 1) * R < p.ntiles ? (split + 1) * R : p.ntiles, split, split * R < p.ntiles};
    if (!w0.ok) w0.t = 0;
    Walker wi = w0;
#pragma unroll 1
    for (int i = 0; i < NBUF; ++i) {
        WPieces<QP, NT>(p, smem + i * STAGE, wi.t, qb, wi.ks, wave, lane).issue_all();
        advance(wi);
    }
    // The stage to copy next (wi), as the wave's piece addresses: inside a tile a stage is one constant step further in
    // both operands, so the loop adds instead of multiplying, and the walker's branches run once per tile, not per stage
    // (round 3: what a wave pays per stage beside MFMA-streaming waves is its NUMBER of instructions).
    const char* ga_i;
    const char* gb_i;
    auto wi_ptrs = [&]() {
        ga_i = WPieces<QP, NT>::uniform_ptr((const char*)p.Xp + ((size_t)wi.t * p.nks + wi.ks) * HSTAGE_A + wave * 1024);
        gb_i = WPieces<QP, NT>::uniform_ptr((const char*)p.Qp + ((size_t)qb * p.nks + wi.ks) * (C::PART_B * QP) + wave * 1024);
    };
    auto step_wi = [&]() {
        if (wi.ok && wi.ks + 1 < p.nks) {
            ++wi.ks;
            ga_i += HSTAGE_A;
            gb_i += C::PART_B * QP;
        } else {
            advance(wi);
            wi_ptrs();
        }
    };
    wi_ptrs();
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * (NBUF - 1)) : "memory");
    __syncthreads();

    int slot = 0, tiles_done = 0;
    f32x4 a0, a1, b0, b1, c0, c1;
    // optional trace (ipx_set_param "trace"), the record layout of scan_bf16_kernel: wave 0's cycles in the filter and
    // in episodes, appends of all lanes, episodes
    uint64_t tr_begin = 0, tr_epi = 0, tr_sort = 0, tr_e0 = 0, tr_e1 = 0;
    uint32_t tr_appends = 0, tr_episodes = 0;
    if (p.trace) tr_begin = __builtin_readcyclecounter();
    while (w0.ok) {
        const int t = w0.t;
        bool episode;
        if constexpr (!S16) {
            f32x16 acc[2][2];
    #pragma unroll
            for (int i = 0; i < 2; ++i)
    #pragma unroll
                for (int j = 0; j < 2; ++j)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

            for (int ks = 0; ks < p.nks; ++ks) {
                const uint32_t buf = smem_base + slot * STAGE;
                const int nslot = slot + 1 == NBUF ? 0 : slot + 1;
                // (Reading the next operand set under the current MFMAs — the 8-wave kernel's in-wave software
                // pipeline — was measured here too: 97 ms against 93 ms for this plain order on the 8.8M-row
                // workload; with four waves per SIMD the other waves already cover the read latency.)
                // first 16 features of the stage
                IPX_DS_READ_B128(a0, buf + a_off + pc0, 0); IPX_DS_READ_B128(a1, buf + a_off + pc0, 2048);
                IPX_DS_READ_B128(b0, buf + b_off + pc0, 0); IPX_DS_READ_B128(b1, buf + b_off + pc0, 2048);
                if constexpr (QP == 2) {
                    IPX_DS_READ_B128(c0, buf + b_off + C::PART_B + pc0, 0); IPX_DS_READ_B128(c1, buf + b_off + C::PART_B + pc0, 2048);
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) :: "memory");
                    IPX_W16_MFMA4(a0, a1, b0, b1);
                    w16_wait2(c0, c1);
                    IPX_W16_MFMA4(a0, a1, c0, c1);
                } else {
                    w16_wait4(a0, a1, b0, b1);
                    IPX_W16_MFMA4(a0, a1, b0, b1);
                }
                asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
                // last 16 features
                IPX_DS_READ_B128(a0, buf + a_off + pc1, 0); IPX_DS_READ_B128(a1, buf + a_off + pc1, 2048);
                IPX_DS_READ_B128(b0, buf + b_off + pc1, 0); IPX_DS_READ_B128(b1, buf + b_off + pc1, 2048);
                if constexpr (QP == 2) {
                    IPX_DS_READ_B128(c0, buf + b_off + C::PART_B + pc1, 0); IPX_DS_READ_B128(c1, buf + b_off + C::PART_B + pc1, 2048);
                }
                const WPieces<QP, NT> sp(ga_i, gb_i, smem + slot * STAGE, wave, lane);
                if constexpr (QP == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1) :: "memory");
                else w16_wait4(a0, a1, b0, b1);
                // this wave's last reads of the slot are in registers; its pieces of the next stage have landed
                // (NBUF - 2 younger stages may be in flight); after the barrier both hold for all 16 waves
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * (NBUF - 2)) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ORD == 0) {
                    sp.issue_all();
                    IPX_W16_MFMA4(a0, a1, b0, b1);
                    if constexpr (QP == 2) { IPX_W16_MFMA4(a0, a1, c0, c1); }
                } else if constexpr (ORD == 2) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(b0), acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(b1), acc[0][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue_a();
                    __builtin_amdgcn_sched_barrier(0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(b0), acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(b1), acc[1][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue_b();
                    if constexpr (QP == 2) { IPX_W16_MFMA4(a0, a1, c0, c1); }
                } else {
                    if constexpr (ORD == 3) __builtin_amdgcn_s_setprio(1);
                    IPX_W16_MFMA4(a0, a1, b0, b1);
                    if constexpr (QP == 2) { IPX_W16_MFMA4(a0, a1, c0, c1); }
                    if constexpr (ORD == 3) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue_all();
                }
                asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
                step_wi();
                slot = nslot;
            }
            w0.ks = p.nks - 1;          // the tile is done: on to the walker's next tile
            advance(w0);

            // ---- epilogue: threshold filter, rare append to the per-query queues ----
            ++tiles_done;
            if (p.trace) tr_e0 = __builtin_readcyclecounter();
            const int docbase = t * HBD + wd * 64 + 4 * h;
            episode = lds_read_i32_nodma(&flags[(tiles_done + 1) & 1]) != 0;
            int* myflag = &flags[tiles_done & 1];
    #pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ql = wq * 64 + j * 32 + l31;
                const float tau = lds_read_f32_nodma(&taus[ql]);
                float m = acc[0][j][0];
    #pragma unroll
                for (int i = 0; i < 2; ++i)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
                if (__any(m > tau)) {
                    uint64_t* qu = myqueues + (size_t)ql * p.cap;
    #pragma unroll
                    for (int i = 0; i < 2; ++i)
    #pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float sc = acc[i][j][r];
                            if (__any(sc > tau)) {
                                const int64_t doc = docbase + i * 32 + (r & 3) + 8 * (r >> 2);
                                if (sc > tau && doc < p.N) {
                                    const int pos = lds_atomic_inc_nodma(&cnts[ql]);
                                    if (pos < p.cap) qu[pos] = make_key(sc, (uint32_t)doc);
                                    else *p.err = 1;
                                    if (pos >= fill_limit) lds_write_i32_nodma(myflag, 1);
                                    ++tr_appends;
                                }
                            }
                        }
                }
            }
        } else {
            // v_mfma_f32_16x16x32_bf16: one MFMA covers a whole K-stage (32 features) of a 16 x 16 tile; the
            // wave's 64 x 64 corner is 4 x 4 of them. Same flops per cycle as the 32x32x16 form on paper, but
            // the chip sustains a higher clock on it (MI355X_MICROARCH.md: ~1.15x the FLOP/s in bare loops).
            // Lane (r = l & 15, g = l >> 4) holds features 8g .. 8g+7 of row r: 16-byte chunk g of the row's
            // 64-byte stage line, swizzled as packed.
            f32x4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int r15 = lane & 15, g4 = lane >> 4;
            // Which 8 features a lane group multiplies is free as long as the corpus and the query operand agree
            // (the MFMA sums over all 32): group g takes 16-byte chunk sigma(g) = (0, 3, 1, 2)[g] of the stage line.
            // With sigma = identity every ds_read_b128 here was 2-way bank-conflicted (SQ_LDS_BANK_CONFLICT = one extra
            // LDS cycle per 16 lanes, 512 cycles per stage and CU): the instruction is served in lane groups
            // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), i.e. rows 0-3 and 12-15 of group g with rows 4-11 of
            // group g + 1, and under the packed swizzle (chunk ^ (row >> 2 & 3)) rows 0-3 / 4-7 and 8-11 / 12-15 then
            // landed on the same banks. sigma makes the four quads of rows of every lane group hit four different
            // chunks.
            const int sigma = (0x9C >> (2 * g4)) & 3;                 // 0, 3, 1, 2
            const uint32_t chunk = (uint32_t)((sigma ^ ((r15 >> 2) & 3)) * 16);
            const uint32_t a16 = (wd * 64 + r15) * 64 + chunk;
            const uint32_t b16 = HSTAGE_A + (wq * 64 + r15) * 64 + chunk;
            f32x4 x0, x1, x2, x3;
#define IPX_W16_ROW(J_, B_)                                                                                              \
            acc[0][J_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x0), as_bf16x8(B_), acc[0][J_], 0, 0, 0);     \
            acc[1][J_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x1), as_bf16x8(B_), acc[1][J_], 0, 0, 0);     \
            acc[2][J_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x2), as_bf16x8(B_), acc[2][J_], 0, 0, 0);     \
            acc[3][J_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x3), as_bf16x8(B_), acc[3][J_], 0, 0, 0)
            for (int ks = 0; ks < p.nks; ++ks) {
                const uint32_t buf = smem_base + slot * STAGE;
                const int nslot = slot + 1 == NBUF ? 0 : slot + 1;
                IPX_DS_READ_B128(x0, buf + a16, 0); IPX_DS_READ_B128(x1, buf + a16, 1024);
                IPX_DS_READ_B128(x2, buf + a16, 2048); IPX_DS_READ_B128(x3, buf + a16, 3072);
                IPX_DS_READ_B128(b0, buf + b16, 0); IPX_DS_READ_B128(b1, buf + b16, 1024);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(b0), "+v"(b1) :: "memory");
                IPX_W16_ROW(0, b0);
                IPX_W16_ROW(1, b1);
                __builtin_amdgcn_sched_barrier(0);
                IPX_DS_READ_B128(b0, buf + b16, 2048); IPX_DS_READ_B128(b1, buf + b16, 3072);
                const WPieces<QP, NT> sp(ga_i, gb_i, smem + slot * STAGE, wave, lane);
                w16_wait2(b0, b1);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * (NBUF - 2)) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                sp.issue_all();
                IPX_W16_ROW(2, b0);
                IPX_W16_ROW(3, b1);
                __builtin_amdgcn_sched_barrier(0);
                step_wi();
                slot = nslot;
            }
            w0.ks = p.nks - 1;          // the tile is done: on to the walker's next tile
            advance(w0);
#undef IPX_W16_ROW
            ++tiles_done;
            if (p.trace) tr_e0 = __builtin_readcyclecounter();
            const int docbase = t * HBD + wd * 64 + 4 * g4;
            episode = lds_read_i32_nodma(&flags[(tiles_done + 1) & 1]) != 0;
            int* myflag = &flags[tiles_done & 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ql = wq * 64 + j * 16 + r15;
                const float tau = lds_read_f32_nodma(&taus[ql]);
                float m = acc[0][j][0];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[i][j][r]);
                if (__any(m > tau)) {
                    // (Counting a lane's hits and taking all its queue slots with one LDS atomic, then a predicated
                    // store per register, was measured: 89.5 against 84 ms — in the steady state a group has 0 or 1
                    // hit, and sixteen exec-masked store blocks cost more than sixteen wave-uniform branches.)
                    uint64_t* qu = myqueues + (size_t)ql * p.cap;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sc = acc[i][j][r];
                            if (__any(sc > tau)) {
                                const int64_t doc = docbase + i * 16 + r;
                                if (sc > tau && doc < p.N) {
                                    const int pos = lds_atomic_inc_nodma(&cnts[ql]);
                                    if (pos < p.cap) qu[pos] = make_key(sc, (uint32_t)doc);
                                    else *p.err = 1;
                                    if (pos >= fill_limit) lds_write_i32_nodma(myflag, 1);
                                    ++tr_appends;
                                }
                            }
                        }
                }
            }
        }
        if (p.trace) { tr_e1 = __builtin_readcyclecounter(); tr_epi += tr_e1 - tr_e0; }
        episode = episode || is_pow4(tiles_done);
        if (episode) {
            ++tr_episodes;
            // the selection scratch aliases the stage ring: let every copy land, compact, fetch the NBUF stages again
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < 2) flags[tid] = 0;
            if (wave < nselect)
                for (int ql = wave; ql < BQ_; ql += nselect)
                    compact_query_select(p, (uint64_t*)((uint32_t*)smem + (size_t)wave * (p.cap + W16_SEL_AUX)), cnts, taus,
                                         myqueues + (size_t)ql * p.cap, ql, lane, mypub ? mypub + ql : nullptr,
                                         (uint32_t*)smem + (size_t)wave * (p.cap + W16_SEL_AUX) + p.cap, W16_SEL_AUX);
            __syncthreads();
            if (tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
            Walker wr = w0;
#pragma unroll 1
            for (int i = 0; i < NBUF; ++i) {
                const int sl = slot + i < NBUF ? slot + i : slot + i - NBUF;
                WPieces<QP, NT>(p, smem + sl * STAGE, wr.t, qb, wr.ks, wave, lane).issue_all();
                advance(wr);
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * (NBUF - 1)) : "memory");
            __syncthreads();
            if (p.trace) tr_sort += __builtin_readcyclecounter() - tr_e1;
        } else if ((tiles_done & 1) == 0 && is_pow4(tiles_done >> 1)) {
            if (tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
        }
    }
    if (p.trace) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tr_appends += __shfl_xor(tr_appends, off);
        if (lane == 0) atomicAdd((unsigned long long*)(p.trace + (size_t)blockIdx.x * 4 + 3), ((unsigned long long)tr_appends << 16));
        if (tid == 0) {
            uint64_t* tr = p.trace + (size_t)blockIdx.x * 4;
            tr[0] = __builtin_readcyclecounter() - tr_begin;
            tr[1] = tr_epi;
            tr[2] = tr_sort;
            atomicAdd((unsigned long long*)(tr + 3), (unsigned long long)tr_episodes);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.deepK) {           // deep candidate lists: unsorted, deep_finish_kernel selects and sorts across the splits
        if (wave < nselect) {
            uint32_t* sh = (uint32_t*)smem + (size_t)wave * (p.cap + W16_SEL_AUX);
            for (int ql = wave; ql < BQ_; ql += nselect) {
                const int64_t qglobal = (int64_t)qb * BQ_ + ql;
                if (qglobal < p.nq)
                    compact_query_deep_final(p, sh, sh + p.cap, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane, qglobal, split);
            }
        }
        return;
    }
    if (wave < nsortwaves)
        for (int ql = wave; ql < BQ_; ql += nsortwaves) {
            const int64_t qglobal = (int64_t)qb * BQ_ + ql;
            if (qglobal < p.nq)
                compact_query(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane, true, qglobal, split, nullptr);
        }
}
#undef IPX_W16_MFMA4

}  // namespace ipx
