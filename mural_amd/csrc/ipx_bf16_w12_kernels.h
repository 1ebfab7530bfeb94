// bf16 scan, large batches, one query part: a 384-row x 256-query tile on 12 waves. Why: the 16-wave 256 x 256 kernel
// (ipx_bf16_w16_kernels.h) is held back by the bytes it moves per flop — per 32-feature stage it copies
// (256 + 256) x 64 B from L2 into LDS and every wave reads (64 + 64) x 64 B of it back for its 64 x 64 corner. Here the
// stage is (384 + 256) x 64 B for 1.5x the flops (-17 % fill per flop), the twelve waves (3 per SIMD, <= 168 registers)
// hold 128 rows x 64 queries each (acc 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 = 128 registers; -25 % LDS reads per
// flop) and there is one barrier per 1.5x the work. Same packed layouts (a 384-row tile is three 128-row halves of the
// 256-row packed tiles), same streaming top-k machinery, same results.
// Included by ipx.hip after ipx_bf16_w16_kernels.h.
#pragma once
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 on a clobber list (the copy pieces set it themselves)

namespace ipx {

template <int NB = 2> struct XCfg;
struct XCfgBase {
    static constexpr int THREADS = 768;
    static constexpr int TD = 384;                       // corpus rows per tile
    static constexpr int A_BYTES = TD * 64;              // 24 KiB
    static constexpr int B_BYTES = 256 * 64;             // 16 KiB
    static constexpr int STAGE = A_BYTES + B_BYTES;      // 40 KiB = 40 copy pieces: 4 for waves 0-3, 3 for the others
};
template <int NB> struct XCfg : XCfgBase {
    static constexpr int NBUF = NB;                      // 2: one stage in flight (deeper rings clock lower)
    static constexpr int RING = NBUF * STAGE;            // 80 / 120 KiB
};

// The copy pieces of one stage. Corpus: 24 pieces, wave w takes pieces w and w + 12 (piece i = 1 KiB i & 7 of the
// 128-row half i >> 3; half h of tile T is half (3T + h) & 1 of packed tile (3T + h) >> 1, clamped to the last packed
// half — rows behind N are dropped by the filter whatever their bytes). Queries: 16 pieces, wave w takes piece w and,
// if w < 4, piece w + 12.
// Issued from asm in the scalar-base form (base in scalar registers, the lane's 16 bytes as a 32-bit offset; M0 = LDS
// address of the piece, lane l lands at + 16 l): two registers per lane less than a 64-bit address per piece, and hipcc
// then tracks no vector-memory op in the loop and adds no vmcnt(0) of its own.
struct XPieces {
    const char* ga0;
    const char* ga1;
    const char* gb;
    uint32_t la;
    bool extra;
    static __device__ __forceinline__ const char* uniform_ptr(const char* ptr) {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    __device__ __forceinline__ XPieces(const ScanParams& p, uint32_t buf, int tile, int qb, int ks, int wave, int hmax) {
        int h0 = 3 * tile + (wave >> 3), h1 = 3 * tile + ((wave + 12) >> 3);
        h0 = h0 < hmax ? h0 : hmax;
        h1 = h1 < hmax ? h1 : hmax;
        ga0 = uniform_ptr((const char*)p.Xp + ((size_t)(h0 >> 1) * p.nks + ks) * HSTAGE_A + (h0 & 1) * 8192 + (wave & 7) * 1024);
        ga1 = uniform_ptr((const char*)p.Xp + ((size_t)(h1 >> 1) * p.nks + ks) * HSTAGE_A + (h1 & 1) * 8192 + ((wave + 12) & 7) * 1024);
        gb = uniform_ptr((const char*)p.Qp + ((size_t)qb * p.nks + ks) * XCfgBase::B_BYTES + wave * 1024);
        la = __builtin_amdgcn_readfirstlane(buf + wave * 1024);
        extra = wave < 4;
    }
    static __device__ __forceinline__ void piece(const char* g, uint32_t l, uint32_t lane16) {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(lane16), "s"(g) : "memory", "m0");
    }
    __device__ __forceinline__ void issue_all(uint32_t lane16) const {
        piece(ga0, la, lane16);
        piece(ga1, la + 12 * 1024, lane16);
        piece(gb, la + XCfgBase::A_BYTES, lane16);
        if (extra) piece(gb + 12 * 1024, la + XCfgBase::A_BYTES + 12 * 1024, lane16);
    }
};

template <int NB>
__global__ void __launch_bounds__(768)
scan_bf16_w12_kernel(const ScanParams p, const int nsortwaves) {
    using C = XCfg<NB>;
    constexpr int BQ_ = 256, NBUF = C::NBUF, STAGE = C::STAGE, TD = C::TD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the state words sit behind whichever is largest: the ring, the final sorts' scratch, the selections' scratch
    // (the host sizes the allocation by the same rule; the queue capacity is a multiple of 512)
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
    const int wq = wave / 3, wd = wave - 3 * wq;          // 3 x 4 waves: 128 rows x 64 queries each

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
    const int fill_limit = p.cap - 2 * TD;
    const int hmax = 2 * (int)((p.N + HBD - 1) / HBD) - 1;         // the last 128-row half of the packed corpus

    uint64_t* myqueues = p.cand + (size_t)blockIdx.x * BQ_ * p.cap;
    uint64_t* scratch = (uint64_t*)smem + (size_t)wave * sort_cap;
    const int nselect = p.nselect;

    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // operand addressing of the 16x16x32 form as in scan_bf16_w16_kernel<1, true>: lane (r = l & 15, g = l >> 4) reads
    // 16-byte chunk sigma(g) = (0, 3, 1, 2)[g] of row r's 64-byte stage line (packed swizzle: chunk ^ (row >> 2 & 3))
    const int r15 = lane & 15, g4 = lane >> 4;
    const int sigma = (0x9C >> (2 * g4)) & 3;
    const uint32_t chunk = (uint32_t)((sigma ^ ((r15 >> 2) & 3)) * 16);
    const uint32_t a16 = (wd * 128 + r15) * 64 + chunk;
    const int b_delta = C::A_BYTES + (wq * 64 - wd * 128) * 64;        // wave-uniform: query operand = row operand + this
    const uint32_t lane16 = (uint32_t)lane * 16;

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
    // this wave's pieces of the oldest stage in flight have landed: at most those of N younger stages are still out
    auto wait_but = [&](auto n_tag) {
        constexpr int N = decltype(n_tag)::value;
        if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * N) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * N) : "memory");
    };
    auto wait_first = [&]() { wait_but(std::integral_constant<int, NBUF - 1>{}); };
    Walker w0{split * R, 0, (split + 1) * R < p.ntiles ? (split + 1) * R : p.ntiles, split, split * R < p.ntiles};
    if (!w0.ok) w0.t = 0;
    Walker wi = w0;
#pragma unroll 1
    for (int i = 0; i < NBUF; ++i) {
        XPieces(p, smem_base + i * STAGE, wi.t, qb, wi.ks, wave, hmax).issue_all(lane16);
        advance(wi);
    }
    wait_first();
    __syncthreads();

    int slot = 0, tiles_done = 0;
    f32x4 b0, b1, b2, b3, x, y, z;
#define IPX_W12_ROW(I_, X_)                                                                                     \
    acc[I_][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(X_), as_bf16x8(b0), acc[I_][0], 0, 0, 0);   \
    acc[I_][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(X_), as_bf16x8(b1), acc[I_][1], 0, 0, 0);   \
    acc[I_][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(X_), as_bf16x8(b2), acc[I_][2], 0, 0, 0);   \
    acc[I_][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(X_), as_bf16x8(b3), acc[I_][3], 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0)
// the oldest outstanding row operand has arrived (one younger read stays in flight); then the next read goes out
#define IPX_W12_NEXT(ARRIVED_, DST_, OFF_)                                  \
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ARRIVED_) :: "memory");      \
    IPX_DS_READ_B128(DST_, aa, OFF_);                                       \
    __builtin_amdgcn_sched_barrier(0)
    while (w0.ok) {
        const int t = w0.t;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < p.nks; ++ks) {
            const uint32_t aa = smem_base + slot * STAGE + a16, bb = aa + b_delta;
            // the stage's four query operands stay in registers; the eight 16-row operands rotate through three
            // registers sets, two reads in flight under the MFMAs of the third
            IPX_DS_READ_B128(b0, bb, 0); IPX_DS_READ_B128(b1, bb, 1024);
            IPX_DS_READ_B128(b2, bb, 2048); IPX_DS_READ_B128(b3, bb, 3072);
            IPX_DS_READ_B128(x, aa, 0); IPX_DS_READ_B128(y, aa, 1024);
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(x) :: "memory");
            IPX_DS_READ_B128(z, aa, 2048);
            __builtin_amdgcn_sched_barrier(0);
            IPX_W12_ROW(0, x);
            IPX_W12_NEXT(y, x, 3072);
            IPX_W12_ROW(1, y);
            IPX_W12_NEXT(z, y, 4096);
            IPX_W12_ROW(2, z);
            IPX_W12_NEXT(x, z, 5120);
            IPX_W12_ROW(3, x);
            IPX_W12_NEXT(y, x, 6144);
            IPX_W12_ROW(4, y);
            IPX_W12_NEXT(z, y, 7168);
            IPX_W12_ROW(5, z);
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(x) :: "memory");
            IPX_W12_ROW(6, x);
            const XPieces sp(p, smem_base + slot * STAGE, wi.t, qb, wi.ks, wave, hmax);
            // this wave's last reads of the slot are in registers and its pieces of the next stage have landed; behind
            // the barrier both hold for all 12 waves: the slot is refilled under the last MFMAs
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y) :: "memory");
            wait_but(std::integral_constant<int, NBUF - 2>{});
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            sp.issue_all(lane16);
            IPX_W12_ROW(7, y);
            advance(wi);
            advance(w0);
            slot = slot + 1 == NBUF ? 0 : slot + 1;
        }

        // ---- epilogue: threshold filter, rare append to the per-query queues ----
        // (the lane-derived addresses below are recomputed per tile from a value the compiler cannot hoist over the K
        // loop, where every register is taken)
        ++tiles_done;
        int le = lane;
        asm volatile("" : "+v"(le));
        const int r15e = le & 15, g4e = le >> 4;
        const int docbase = t * TD + wd * 128 + 4 * g4e;
        bool episode = lds_read_i32_nodma(&flags[(tiles_done + 1) & 1]) != 0;
        int* myflag = &flags[tiles_done & 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ql = wq * 64 + j * 16 + r15e;
            const float tau = lds_read_f32_nodma(&taus[ql]);
            float m = acc[0][j][0];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[i][j][r]);
            if (__any(m > tau)) {
                uint64_t* qu = myqueues + (size_t)ql * p.cap;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sc = acc[i][j][r];
                        if (__any(sc > tau)) {
                            const int64_t doc = (int64_t)docbase + i * 16 + r;
                            if (sc > tau && doc < p.N) {
                                const int pos = lds_atomic_inc_nodma(&cnts[ql]);
                                if (pos < p.cap) qu[pos] = make_key(sc, (uint32_t)doc);
                                else *p.err = 1;
                                if (pos >= fill_limit) lds_write_i32_nodma(myflag, 1);
                            }
                        }
                    }
            }
        }
        episode = episode || is_pow4(tiles_done);
        if (episode) {
            // the selection scratch aliases the stage ring: let every copy land, compact, fetch the two stages again
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            int te = tid;                     // (as in the epilogue: nothing lane-derived is to be kept over the K loop)
            asm volatile("" : "+v"(te));
            if (te < 2) flags[te] = 0;
            if (wave < nselect)
                for (int ql = wave; ql < BQ_; ql += nselect)
                    compact_query_select(p, (uint64_t*)((uint32_t*)smem + (size_t)wave * (p.cap + W16_SEL_AUX)), cnts, taus,
                                         myqueues + (size_t)ql * p.cap, ql, te & 63, mypub ? mypub + ql : nullptr,
                                         (uint32_t*)smem + (size_t)wave * (p.cap + W16_SEL_AUX) + p.cap, W16_SEL_AUX);
            __syncthreads();
            if (te < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, te);
            Walker wr = w0;
#pragma unroll 1
            for (int i = 0; i < NBUF; ++i) {
                XPieces(p, smem_base + (slot + i < NBUF ? slot + i : slot + i - NBUF) * STAGE, wr.t, qb, wr.ks, wave, hmax).issue_all(lane16);
                advance(wr);
            }
            wait_first();
            __syncthreads();
        } else if ((tiles_done & 1) == 0 && is_pow4(tiles_done >> 1)) {
            int te = tid;
            asm volatile("" : "+v"(te));
            if (te < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, te);
        }
    }
#undef IPX_W12_ROW
#undef IPX_W12_NEXT
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave < nsortwaves)
        for (int ql = wave; ql < BQ_; ql += nsortwaves) {
            const int64_t qglobal = (int64_t)qb * BQ_ + ql;
            if (qglobal < p.nq)
                compact_query(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane, true, qglobal, split, nullptr);
        }
}

}  // namespace ipx
#pragma clang diagnostic pop
