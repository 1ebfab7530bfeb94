// fp32 scan for small query batches (<= 64 queries per workgroup): the HBM-bound regime on an fp32
// index (BASELINE.json: "HBM GB/s for small-batch streaming"). Same packed corpus, same MFMA
// (v_mfma_f32_32x32x2_f32), same order of the products inside every score as scan_f32_kernel — a
// score does not depend on the batch it was computed in, bit for bit — but the tile is 256 rows x
// 32 QT queries instead of 128 x 256, so a corpus byte costs 1/8 (QT 1) or 1/4 (QT 2) of the
// matrix-core time: the corpus stream sets the pace.
// Measured on gfx950: one wave sustains about one 1-KiB LDS-DMA piece per 235 ns whatever the ring
// depth, so the stream rate of a CU is set by how many waves copy. Hence 8 waves per workgroup and
// two workgroups per CU (16 waves), each wave copying only the 2 KiB of corpus rows it multiplies
// itself (a pair of 128-row corpus tiles per step: wave = (tile of the pair, 32-row group)); the
// small query stage is shared (copied by the first 2 QT waves).
// Included by ipx.hip after ipx_kernels.h and ipx_bf16_kernels.h (shares their helpers).
#pragma once

namespace ipx {

constexpr int IPX_MAX_SUBSPLITS = 4;    // short-list path of the small-batch scans: sub-splits per split

template <int QT> struct SCfg {
    static constexpr int BQ_ = 32 * QT;                       // queries per workgroup
    static constexpr int THREADS = 512;
    static constexpr int PAIR_A = 2 * STAGE_A_BYTES;          // corpus bytes per stage: 2 tiles x 8 KiB
    static constexpr int STAGE_B = BQ_ * 64;                  // query stage: 2 / 4 KiB
    static constexpr int STAGE = PAIR_A + STAGE_B;            // 18 / 20 KiB
    static constexpr int NBUF = QT == 1 ? 4 : 3;              // ring slots (two workgroups per CU: <= 80 KiB each)
    static constexpr int NBW = 2 * QT;                        // waves that also copy a query piece
    static constexpr int RING = NBUF * STAGE;                 // 72 / 60 KiB
};

// The copy pieces of one stage for one wave: 2 x 1 KiB of its own corpus rows; waves < NBW also
// one 1-KiB piece of the query stage.
// (Round 3: issued from asm in the scalar-base form — M0 = the piece's LDS address, base in scalar registers, the lane's
// 16 bytes as a 32-bit offset: no per-lane 64-bit address arithmetic per stage. At 64 queries this kernel is bound by
// the fp32 matrix work, and a VALU instruction beside MFMA-streaming waves is not free: DESIGN.md 4.1.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 on a clobber list (the copy pieces set it themselves)
template <int QT>
struct SmallPieces {
    const char* ga;
    const char* gb;
    uint32_t la;
    uint32_t lb;
    uint32_t lane16;
    static __device__ __forceinline__ const char* uniform_ptr(const char* ptr) {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    __device__ __forceinline__ SmallPieces(const ScanParams& p, char* slot, int pair, int qb, int ks, int wave, int lane) {
        using C = SCfg<QT>;
        const int th = wave >> 2, wd = wave & 3;
        int64_t tile = (int64_t)pair * 2 + th;
        if (tile >= p.ntiles) tile = p.ntiles - 1;            // odd tile count: the last pair repeats its first tile (rows masked by N)
        ga = uniform_ptr((const char*)p.Xp + ((size_t)tile * p.nks + ks) * STAGE_A_BYTES + wd * 2048);
        gb = uniform_ptr((const char*)p.Qp + ((size_t)qb * p.nks + ks) * C::STAGE_B + (wave & (C::NBW - 1)) * 1024);
        const uint32_t lslot = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)slot;
        la = __builtin_amdgcn_readfirstlane(lslot + th * STAGE_A_BYTES + wd * 2048);
        lb = __builtin_amdgcn_readfirstlane(lslot + C::PAIR_A + (wave & (C::NBW - 1)) * 1024);
        lane16 = (uint32_t)lane * 16;
    }
    // from the wave's piece addresses of the stage (kept as running pointers by the scan loop)
    __device__ __forceinline__ SmallPieces(const char* ga_, const char* gb_, char* slot, int wave, int lane) {
        using C = SCfg<QT>;
        const int th = wave >> 2, wd = wave & 3;
        ga = ga_;
        gb = gb_;
        const uint32_t lslot = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)slot;
        la = __builtin_amdgcn_readfirstlane(lslot + th * STAGE_A_BYTES + wd * 2048);
        lb = __builtin_amdgcn_readfirstlane(lslot + C::PAIR_A + (wave & (C::NBW - 1)) * 1024);
        lane16 = (uint32_t)lane * 16;
    }
    // With one query block the corpus rows are read once per search: non-temporal (+1.5 % measured). With several,
    // the other blocks' workgroups find them in L2: default policy.
    __device__ __forceinline__ void issue_a(int i, bool once) const {
        const char* g = ga + i * 1024;
        const uint32_t l = la + i * 1024;
        if (once)
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 nt" :: "s"(l), "v"(lane16), "s"(g) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(lane16), "s"(g) : "memory", "m0");
    }
    __device__ __forceinline__ void issue_b() const {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lb), "v"(lane16), "s"(gb) : "memory", "m0");
    }
};
#pragma clang diagnostic pop

// s_waitcnt lgkmcnt(N) that the compiler sees as (re)defining the operand registers waited for
template <int QT, int N>
__device__ __forceinline__ void wait_ops(f32x4& a, f32x4 (&b)[QT]) {
    if constexpr (QT == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b[0]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
}
template <int QT>
__device__ __forceinline__ void read_ops(f32x4& a, f32x4 (&b)[QT], uint32_t a_addr, uint32_t b_addr) {
    IPX_DS_READ_B128(a, a_addr, 0);
    IPX_DS_READ_B128(b[0], b_addr, 0);
    if constexpr (QT == 2) IPX_DS_READ_B128(b[1], b_addr, 2048);
}
template <int QT>
__device__ __forceinline__ void pin1(f32x16 (&acc)[QT]) {
    if constexpr (QT == 1) asm volatile("" : "+v"(acc[0]));
    else asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
}

template <int QT>
__global__ void __launch_bounds__(512, 2)
scan_f32_small_kernel(const ScanParams p, const int nsortwaves) {
    using C = SCfg<QT>;
    constexpr int BQ_ = C::BQ_, NBUF = C::NBUF, STAGE = C::STAGE, THREADS = C::THREADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int scratch_bytes = nsortwaves * p.cap * 8;
    const int stage_region = scratch_bytes > C::RING ? scratch_bytes : C::RING;
    int* cnts = (int*)(smem + stage_region);
    float* taus = (float*)(smem + stage_region + BQ_ * 4);
    int* flags = (int*)(smem + stage_region + 2 * BQ_ * 4);
    uint32_t* best = (uint32_t*)(flags + 4);              // [IPX_MAX_SUBSPLITS][BQ_] best scores so far (ordered bits)
    uint32_t* tmin = best + IPX_MAX_SUBSPLITS * BQ_;      // [BQ_]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int th = wave >> 2, wd = wave & 3;
    const int h = lane >> 5, l31 = lane & 31;
    const bool copies_b = wave < C::NBW;
    const bool once = (int)gridDim.x == p.S;       // a single query block

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = logical / p.S, split = logical - qb * p.S;

    if (tid < BQ_) {
        cnts[tid] = 0;
        taus[tid] = ((int64_t)qb * BQ_ + tid < p.nq) ? -FLT_MAX : INFINITY;
#pragma unroll
        for (int u = 0; u < IPX_MAX_SUBSPLITS; ++u) best[u * BQ_ + tid] = 0;
    }
    if (tid < 2) flags[tid] = 0;
    const int fill_limit = p.cap - 4 * BD;      // a set flag is acted on one step (2 tiles) later
    // Short lists (k <= 4 S): no compaction episodes unless a queue fills. Each split deals its steps
    // round-robin to r = ceil(k / S) sub-splits and keeps the best score of each by an LDS max beside
    // the appends; the S r bests are published at r, 2r, 4r, ... steps and every 32, and their minimum
    // bounds the k-th best score of the index from below (S r >= k disjoint sets of rows, each holding
    // a row at or above it). With r == 1 this is "every split publishes its best".
    const bool fast1 = p.S > 1 && p.r <= IPX_MAX_SUBSPLITS;
    const int rs = fast1 ? p.r : 1;             // published values per split and query
    uint32_t* mypub = p.S > 1 ? p.pub + ((size_t)qb * p.S + split) * rs * BQ_ : nullptr;

    uint64_t* myqueues = p.cand + (size_t)blockIdx.x * BQ_ * p.cap;
    uint64_t* scratch = (uint64_t*)smem + (size_t)wave * p.cap;

    const int sw = (lane >> 2) & 3;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // lane (l31, h): row l31 of the wave's corpus rows / of each 32-query group, 16-byte chunk h
    // (set 0) or 2 + h (set 1) of the stage, swizzled as the pack kernels wrote it
    const uint32_t pc0 = ((0 + h) ^ sw) * 16, pc1 = ((2 + h) ^ sw) * 16;
    const uint32_t a_off = th * STAGE_A_BYTES + (wd * 32 + l31) * 64;
    const uint32_t b_off = C::PAIR_A + l31 * 64;

    // Runs of R tile pairs dealt round-robin: split s takes runs s, s + S, s + 2S, ... A walker is
    // a position in that sequence of stages: `w0` is the stage being computed (ring slot `slot`),
    // `wi` the next stage to copy, NBUF stages ahead. Past the end a walker stays on its last stage
    // with ok == false (the pipeline still issues a copy for it, into a slot nobody reads, so the
    // loop has no branches).
    const int R = p.run_tiles;
    const int npairs = (p.ntiles + 1) >> 1;
    struct Walker { int t, ks, end, run; bool ok; };
    auto advance = [&](Walker& w) {
        if (!w.ok) return;
        if (w.ks + 1 < p.nks) { ++w.ks; return; }
        if (w.t + 1 < w.end) { ++w.t; w.ks = 0; return; }
        const int64_t next = (int64_t)(w.run + p.S) * R;
        if (next < npairs) {
            w.run += p.S;
            w.t = (int)next;
            w.end = w.t + R < npairs ? w.t + R : npairs;
            w.ks = 0;
            return;
        }
        w.ok = false;
    };
    Walker w0{split * R, 0, (split + 1) * R < npairs ? (split + 1) * R : npairs, split, split * R < npairs};
    if (!w0.ok) w0.t = 0;                       // (never computed on; keeps the dummy copies in bounds)
    Walker wi = w0;
    auto issue_stage = [&](char* slot_ptr, const Walker& w) {
        const SmallPieces<QT> sp(p, slot_ptr, w.t, qb, w.ks, wave, lane);
        sp.issue_a(0, once);
        sp.issue_a(1, once);
        if (copies_b) sp.issue_b();
    };
    // own pieces of all but the newest NBUF - 1 / NBUF - 2 stages have landed
    auto wait_newest = [&](auto nstages) {
        constexpr int n = decltype(nstages)::value;
        if (copies_b) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * n) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * n) : "memory");
    };
#pragma unroll 1
    for (int i = 0; i < NBUF; ++i) {
        issue_stage(smem + i * STAGE, wi);
        advance(wi);
    }
    wait_newest(std::integral_constant<int, NBUF - 1>{});
    __syncthreads();
    // The stage to copy next (wi), as the wave's piece addresses: inside a tile pair a stage is one constant step further
    // in both operands, so the loop adds instead of multiplying, and the walker's branches run once per pair.
    const char* ga_i;
    const char* gb_i;
    auto wi_ptrs = [&]() {
        const SmallPieces<QT> sp(p, smem, wi.t, qb, wi.ks, wave, lane);
        ga_i = sp.ga;
        gb_i = sp.gb;
    };
    auto step_wi = [&]() {
        if (wi.ok && wi.ks + 1 < p.nks) {
            ++wi.ks;
            ga_i += STAGE_A_BYTES;
            gb_i += C::STAGE_B;
        } else {
            advance(wi);
            wi_ptrs();
        }
    };
    wi_ptrs();

    // operand registers: set 0 = features 0..7 of the stage, set 1 = features 8..15; lane (l31, h)
    // holds features 4h..4h+3 of its row in each: the MFMA order inside an 8-feature group is
    // 0,4,1,5,2,6,3,7, exactly scan_f32_kernel's.
    f32x4 a0, a1, b0[QT], b1[QT];
    read_ops<QT>(a0, b0, smem_base + a_off + pc0, smem_base + b_off + pc0);

    uint64_t t_start = 0;
    if (p.trace) t_start = __builtin_amdgcn_s_memrealtime();
    int slot = 0, tiles_done = 0, n_compact = 0;
    while (w0.ok) {
        const int t = w0.t;
        f32x16 acc[QT];
#pragma unroll
        for (int j = 0; j < QT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#define IPX_M(A_, B_, kk_) \
        _Pragma("unroll") for (int j = 0; j < QT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[kk_], B_[j][kk_], acc[j], 0, 0, 0); \
        pin1<QT>(acc); __builtin_amdgcn_sched_barrier(0)
        for (int ks = 0; ks < p.nks; ++ks) {
            const int nslot = slot + 1 == NBUF ? 0 : slot + 1;
            const uint32_t buf = smem_base + slot * STAGE;
            const uint32_t nbuf = smem_base + nslot * STAGE;
            read_ops<QT>(a1, b1, buf + a_off + pc1, buf + b_off + pc1);
            wait_ops<QT, 1 + QT>(a0, b0);
            IPX_M(a0, b0, 0);
            IPX_M(a0, b0, 1);
            IPX_M(a0, b0, 2);
            IPX_M(a0, b0, 3);
            wait_ops<QT, 0>(a1, b1);
            // this stage's last operands are in registers; the next stage's pieces of this wave
            // have landed (NBUF - 2 younger stages may be in flight); after the barrier both hold
            // for every wave of the workgroup
            const SmallPieces<QT> sp(ga_i, gb_i, smem + slot * STAGE, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
            wait_newest(std::integral_constant<int, NBUF - 2>{});
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            IPX_M(a1, b1, 0);
            read_ops<QT>(a0, b0, nbuf + a_off + pc0, nbuf + b_off + pc0);
            __builtin_amdgcn_sched_barrier(0);
            sp.issue_a(0, once);
            __builtin_amdgcn_sched_barrier(0);
            IPX_M(a1, b1, 1);
            sp.issue_a(1, once);
            __builtin_amdgcn_sched_barrier(0);
            IPX_M(a1, b1, 2);
            if (copies_b) sp.issue_b();
            __builtin_amdgcn_sched_barrier(0);
            IPX_M(a1, b1, 3);
            step_wi();
            slot = nslot;
        }
        w0.ks = p.nks - 1;          // the pair is done: on to the walker's next pair
        advance(w0);
#undef IPX_M

        // ---- epilogue: threshold filter, rare append to the per-query queues ----
        ++tiles_done;
        const int docbase = (t * 2 + th) * BD + wd * 32 + 4 * h;
        // flag raised by the PREVIOUS step's appends: every wave reads the same value (a step's
        // barriers lie between those writes and this read; this step writes the other flag)
        bool episode = lds_read_i32_nodma(&flags[(tiles_done + 1) & 1]) != 0;
        int* myflag = &flags[tiles_done & 1];
        const int sub = fast1 ? (tiles_done - 1) % rs : 0;          // this step's sub-split
#pragma unroll
        for (int j = 0; j < QT; ++j) {
            const int ql = j * 32 + l31;
            const float tau = lds_read_f32_nodma(&taus[ql]);
            float m = acc[j][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[j][r]);
            if (__any(m > tau)) {
                uint64_t* qu = myqueues + (size_t)ql * p.cap;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sc = acc[j][r];
                    if (__any(sc > tau)) {
                        const int64_t doc = docbase + (r & 3) + 8 * (r >> 2);
                        if (sc > tau && doc < p.N) {
                            const int pos = lds_atomic_inc_nodma(&cnts[ql]);
                            if (pos < p.cap) qu[pos] = make_key(sc, (uint32_t)doc);
                            else *p.err = 1;
                            if (pos >= fill_limit) lds_write_i32_nodma(myflag, 1);
                            lds_atomic_max_u32_nodma(&best[sub * BQ_ + ql], f32_ordered(sc));
                        }
                    }
                }
            }
        }
        const int rounds = tiles_done / rs;                        // completed rounds over the sub-splits
        if (fast1 && tiles_done == rounds * rs && is_pow4(rounds)) {
            // publish the sub-splits' bests, take the minimum over all splits as the new floor. After
            // the first round every split is at the same point: wait (bounded) until all of them
            // have published, so the next step already runs against a bound.
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            for (int e = tid; e < rs * BQ_; e += THREADS) {
                const uint32_t b = best[e];
                if (b) __hip_atomic_store(mypub + e, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            constexpr int TPQ = THREADS / BQ_;                         // threads per query
            const int q = tid & (BQ_ - 1), part = tid / BQ_;
            const int nvals = p.S * rs;
            const int per = (nvals + TPQ - 1) / TPQ;
            const int s_lo = part * per, s_hi = s_lo + per < nvals ? s_lo + per : nvals;
            const uint32_t* col = p.pub + (size_t)qb * nvals * BQ_ + q;
            for (int attempt = 0; attempt < (rounds == 1 ? 64 : 1); ++attempt) {
                if (tid < BQ_) tmin[tid] = 0xffffffffu;
                __syncthreads();
                uint32_t v = 0xffffffffu;
#pragma unroll 8
                for (int s2 = s_lo; s2 < s_hi; ++s2) {
                    const uint32_t x = __hip_atomic_load(col + (size_t)s2 * BQ_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = x < v ? x : v;
                }
                atomicMin(&tmin[q], v);
                __syncthreads();
                // queries beyond nq never publish: they do not count as "missing"
                const bool missing = tid < BQ_ && (int64_t)qb * BQ_ + tid < p.nq && tmin[tid] == 0;
                if (!__syncthreads_or(missing)) break;
                __builtin_amdgcn_s_sleep(20);
            }
            if (tid < BQ_) {
                const uint32_t v = tmin[tid];
                if (v != 0 && v != 0xffffffffu) taus[tid] = fmaxf(taus[tid], f32_below(f32_unordered(v)));
            }
            __syncthreads();
        }
        episode = episode || (is_pow4(tiles_done) && !fast1);
        if (episode) {
            // the selection scratch aliases the stage ring: let every copy land, compact, fetch the
            // NBUF stages again (w0's set-0 operands are already in registers)
            wait_ops<QT, 0>(a0, b0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ++n_compact;
            if (tid < 2) flags[tid] = 0;
            if (wave < nsortwaves)
                for (int ql = wave; ql < BQ_; ql += nsortwaves)
                    compact_query_select(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane,
                                         mypub && !fast1 ? mypub + ql : nullptr);
            __syncthreads();
            if (!fast1 && tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
            Walker wr = w0;
#pragma unroll 1
            for (int i = 0; i < NBUF; ++i) {
                const int sl = slot + i < NBUF ? slot + i : slot + i - NBUF;
                issue_stage(smem + sl * STAGE, wr);
                advance(wr);
            }
            wait_newest(std::integral_constant<int, NBUF - 1>{});
            __syncthreads();
        } else if (!fast1 && (tiles_done & 1) == 0 && is_pow4(tiles_done >> 1)) {
            if (tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
        }
    }
    wait_ops<QT, 0>(a0, b0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the tail's dummy copies write LDS too
    __syncthreads();
    if (wave < nsortwaves)
        for (int ql = wave; ql < BQ_; ql += nsortwaves) {
            const int64_t qglobal = (int64_t)qb * BQ_ + ql;
            if (qglobal < p.nq)
                compact_query(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane, true, qglobal, split, nullptr);
        }
    if (p.trace && tid == 0) {
        uint64_t* tr = p.trace + (size_t)blockIdx.x * 4;
        tr[0] = t_start;
        tr[1] = __builtin_amdgcn_s_memrealtime();
        tr[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        tr[3] = ((uint64_t)tiles_done << 32) | (uint64_t)n_compact;
    }
}

}  // namespace ipx
