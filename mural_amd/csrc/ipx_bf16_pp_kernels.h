// bf16 scan, ping-pong variant of the 256 x 256 tile (large batches, one query part).
//
// Why it exists. In scan_bf16_w16_kernel every wave of a SIMD runs the same program in phase: behind each
// per-stage barrier all of them issue copy pieces, then all multiply, then all wait for their LDS reads —
// the matrix core idles through every common wait (MFMA busy 0.55, same cycles per stage on an idle chip).
// Here the 8 waves form two groups of four (one wave of each group per SIMD) that run ONE BARRIER APART:
// while a group multiplies a quadrant of its output (16 MFMAs, pure register work), the other group reads
// its next operands from LDS and issues its copy pieces; at the next barrier the roles swap. The matrix core
// of a SIMD always has exactly one wave in a multiply phase.
//
// Geometry. Wave (grp, wq): docs [128 grp, +128) x queries [64 wq, +64) of the tile, as 8 x 4 tiles of
// v_mfma_f32_16x16x32_bf16 (128 accumulator registers). The unit of the pipeline is a K-PAIR = two K-stages
// (64 features) in two adjacent 32-KiB ring slots; the ring holds two K-pairs (128 KiB). Per K-pair a wave
// runs four (load, multiply) phase pairs, one per quadrant of its output:
//   L0: read A0 (docs 0..63, 8 reads), B0 (queries 0..31, 4 reads)      M0: A0 x B0
//   L1: read A1 (docs 64..127, 8 reads)                                 M1: A1 x B0
//   L2: read B1 (queries 32..63, 4 reads; same registers as B0)         M2: A1 x B1
//   L3: -                                                               M3: A0 x B1
// Copy pieces of a wave per K-pair: 4 corpus + 4 query pieces of 1 KiB, two per load phase: L0, L1 fetch
// the QUERY block of K-pair P+1 (an L2 hit: short flight), L2, L3 the CORPUS block of K-pair P+2 (HBM: it
// has two K-pairs to land). Hazards, counted in barrier intervals i (group 0 is in L0 at i = 0, group 1 at
// i = 1):
//   * the corpus block of K-pair P is last read in L1 (i = 2, 3), its slot is refilled from L2 (i = 4, 5);
//   * the query block of K-pair P is last read in L2 (i = 4, 5), its slot is refilled from L0 of K-pair
//     P+1 (i = 8, 9);
//   * every read completes (lgkmcnt(0) at the head of the multiply phase) before its wave passes the next
//     barrier, so at least one barrier lies between a slot's last read and its first refill piece;
//   * L3 waits vmcnt(4): everything but the four corpus pieces of K-pair P+2 has landed, i.e. K-pair P+1
//     is complete for this wave; both groups have passed that wait before the barrier that opens i = 8,
//     where the first read of K-pair P+1 is issued.
// The streaming top-k machinery (queues, thresholds, episodes, union bound) is the other scans'. At an
// episode group 0 waits one barrier for group 1 and the pipeline is restarted behind it.
// Needs an even number of K-stages; the host falls back to scan_bf16_w16_kernel otherwise.
#pragma once

namespace ipx {

struct PPCfg {
    static constexpr int THREADS = 512;
    static constexpr int STAGE = HSTAGE_A + 256 * 64;    // one K-stage: corpus block + query block, 32 KiB
    static constexpr int PAIR = 2 * STAGE;               // a K-pair: 64 KiB
    static constexpr int RING = 2 * PAIR;                // 128 KiB
};

#define IPX_PP_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(A_), as_bf16x8(B_), C_, 0, 0, 0)
// one quadrant: doc tiles I0..I0+3 x query tiles J0, J0+1 over both K-stages of the pair
#define IPX_PP_QUADRANT(AS_, I0_, J0_)                                                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                 \
            IPX_PP_MFMA(AS_[s_][i_], Bs[s_][0], acc[I0_ + i_][J0_]);                                       \
            IPX_PP_MFMA(AS_[s_][i_], Bs[s_][1], acc[I0_ + i_][J0_ + 1]);                                   \
        }

template <bool SETPRIO>
__global__ void __launch_bounds__(512)
scan_bf16_pp_kernel(const ScanParams p, const int nsortwaves) {
    constexpr int BQ_ = 256, STAGE = PPCfg::STAGE, PAIR = PPCfg::PAIR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int scratch_bytes = nsortwaves * p.cap * 8;
    const int stage_region = scratch_bytes > PPCfg::RING ? scratch_bytes : PPCfg::RING;
    int* cnts = (int*)(smem + stage_region);
    float* taus = (float*)(smem + stage_region + BQ_ * 4);
    int* flags = (int*)(smem + stage_region + 2 * BQ_ * 4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int r15 = lane & 15, g4 = lane >> 4;

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
    uint64_t* scratch = (uint64_t*)smem + (size_t)wave * p.cap;

    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // lane (r15, g4) holds features 8 g4 .. 8 g4 + 7 of row r15 of a 16-row tile: chunk g4 of the row's
    // 64-byte stage line, swizzled as packed (rows 16 apart share the swizzle)
    const uint32_t chunk = (uint32_t)((g4 ^ ((r15 >> 2) & 3)) * 16);
    const uint32_t a_rel = (uint32_t)((grp * 128 + r15) * 64) + chunk;
    const uint32_t b_rel = (uint32_t)(HSTAGE_A + (wq * 64 + r15) * 64) + chunk;

    // runs of R tiles dealt round-robin (split s: runs s, s + S, ...); a walker is a position in that
    // sequence of K-pairs
    const int R = p.run_tiles, nkp = p.nks >> 1;
    struct Walker { int t, kp, end, run; bool ok; };
    auto advance = [&](Walker& w) {
        if (!w.ok) return;
        if (w.kp + 1 < nkp) { ++w.kp; return; }
        if (w.t + 1 < w.end) { ++w.t; w.kp = 0; return; }
        const int64_t next = (int64_t)(w.run + p.S) * R;
        if (next < p.ntiles) {
            w.run += p.S;
            w.t = (int)next;
            w.end = w.t + R < p.ntiles ? w.t + R : p.ntiles;
            w.kp = 0;
            return;
        }
        w.ok = false;          // stays on its last K-pair: the dummy copies go to slots nobody reads
    };
    // the two pieces of this wave in K-stage `st` (0 / 1) of a K-pair, corpus block / query block
    auto issue_a = [&](const Walker& w, int st, int ps) {
        const char* g = (const char*)p.Xp + ((size_t)w.t * p.nks + 2 * w.kp + st) * HSTAGE_A + wave * 1024 + lane * 16;
        char* l = smem + ps * PAIR + st * STAGE + wave * 1024;
        IPX_GLDS(g, l);
        IPX_GLDS(g + 8192, l + 8192);
    };
    auto issue_b = [&](const Walker& w, int st, int ps) {
        const char* g = (const char*)p.Qp + ((size_t)qb * p.nks + 2 * w.kp + st) * (256 * 64) + wave * 1024 + lane * 16;
        char* l = smem + ps * PAIR + st * STAGE + HSTAGE_A + wave * 1024;
        IPX_GLDS(g, l);
        IPX_GLDS(g + 8192, l + 8192);
    };

    Walker w0{split * R, 0, (split + 1) * R < p.ntiles ? (split + 1) * R : p.ntiles, split, split * R < p.ntiles};
    if (!w0.ok) w0.t = 0;
    Walker wb, wa;             // K-pair after w0 (query pieces), two after (corpus pieces)
    int ps = 0;                // ring slot of w0's K-pair
    // (re)start of the pipeline at w0: K-pair w0 complete in slot ps, the corpus block of the next one in flight
    auto restart = [&]() {
        if (!w0.ok) return;
        wb = w0;
        issue_a(wb, 0, ps); issue_a(wb, 1, ps);
        issue_b(wb, 0, ps); issue_b(wb, 1, ps);
        advance(wb);
        wa = wb;
        issue_a(wa, 0, ps ^ 1); issue_a(wa, 1, ps ^ 1);
        advance(wa);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
        if (grp == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    };
    restart();

    int tiles_done = 0;
    f32x4 A0s[2][4], A1s[2][4], Bs[2][2];
    f32x4 acc[8][4];
    while (w0.ok) {
        const int t = w0.t;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kp = 0; kp < nkp; ++kp) {
            const uint32_t abuf = smem_base + ps * PAIR + a_rel;
            const uint32_t bbuf = smem_base + ps * PAIR + b_rel;
            // ---- L0: A0, B0 of this K-pair; query pieces of the next one (first K-stage) ----
            IPX_DS_READ_B128(Bs[0][0], bbuf, 0);     IPX_DS_READ_B128(Bs[0][1], bbuf, 1024);
            IPX_DS_READ_B128(Bs[1][0], bbuf, 32768); IPX_DS_READ_B128(Bs[1][1], bbuf, 33792);
            IPX_DS_READ_B128(A0s[0][0], abuf, 0);    IPX_DS_READ_B128(A0s[0][1], abuf, 1024);
            IPX_DS_READ_B128(A0s[0][2], abuf, 2048); IPX_DS_READ_B128(A0s[0][3], abuf, 3072);
            IPX_DS_READ_B128(A0s[1][0], abuf, 32768); IPX_DS_READ_B128(A0s[1][1], abuf, 33792);
            IPX_DS_READ_B128(A0s[1][2], abuf, 34816); IPX_DS_READ_B128(A0s[1][3], abuf, 35840);
            issue_b(wb, 0, ps ^ 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(Bs[0][0]), "+v"(Bs[0][1]), "+v"(Bs[1][0]), "+v"(Bs[1][1]), "+v"(A0s[0][0]), "+v"(A0s[0][1]),
                           "+v"(A0s[0][2]), "+v"(A0s[0][3]), "+v"(A0s[1][0]), "+v"(A0s[1][1]), "+v"(A0s[1][2]), "+v"(A0s[1][3])
                         :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- M0 ----
            if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);
            IPX_PP_QUADRANT(A0s, 0, 0)
            if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- L1: A1; query pieces of the next K-pair (second K-stage) ----
            IPX_DS_READ_B128(A1s[0][0], abuf, 4096); IPX_DS_READ_B128(A1s[0][1], abuf, 5120);
            IPX_DS_READ_B128(A1s[0][2], abuf, 6144); IPX_DS_READ_B128(A1s[0][3], abuf, 7168);
            IPX_DS_READ_B128(A1s[1][0], abuf, 36864); IPX_DS_READ_B128(A1s[1][1], abuf, 37888);
            IPX_DS_READ_B128(A1s[1][2], abuf, 38912); IPX_DS_READ_B128(A1s[1][3], abuf, 39936);
            issue_b(wb, 1, ps ^ 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(A1s[0][0]), "+v"(A1s[0][1]), "+v"(A1s[0][2]), "+v"(A1s[0][3]), "+v"(A1s[1][0]), "+v"(A1s[1][1]),
                           "+v"(A1s[1][2]), "+v"(A1s[1][3])
                         :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- M1 ----
            if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);
            IPX_PP_QUADRANT(A1s, 4, 0)
            if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- L2: B1 (over B0's registers: M1 has been issued); corpus pieces two K-pairs ahead ----
            IPX_DS_READ_B128(Bs[0][0], bbuf, 2048);  IPX_DS_READ_B128(Bs[0][1], bbuf, 3072);
            IPX_DS_READ_B128(Bs[1][0], bbuf, 34816); IPX_DS_READ_B128(Bs[1][1], bbuf, 35840);
            issue_a(wa, 0, ps);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bs[0][0]), "+v"(Bs[0][1]), "+v"(Bs[1][0]), "+v"(Bs[1][1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- M2 ----
            if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);
            IPX_PP_QUADRANT(A1s, 4, 2)
            if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- L3: second half of the corpus pieces; the next K-pair has landed for this wave ----
            issue_a(wa, 1, ps);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- M3 ----
            if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);
            IPX_PP_QUADRANT(A0s, 0, 2)
            if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            advance(w0);
            advance(wb);
            advance(wa);
            ps ^= 1;
        }

        // ---- epilogue: threshold filter, rare append to the per-query queues ----
        ++tiles_done;
        const int docbase = t * HBD + grp * 128 + 4 * g4;
        bool episode = lds_read_i32_nodma(&flags[(tiles_done + 1) & 1]) != 0;
        int* myflag = &flags[tiles_done & 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ql = wq * 64 + j * 16 + r15;
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
                            const int64_t doc = docbase + i * 16 + r;
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
            // group 1 is one barrier behind: group 0 waits for it here, then the selection scratch (which
            // aliases the ring) is used by everybody and the pipeline is restarted at w0
            if (grp == 0) __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < 2) flags[tid] = 0;
            if (wave < nsortwaves)
                for (int ql = wave; ql < BQ_; ql += nsortwaves)
                    compact_query_select(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane,
                                         mypub ? mypub + ql : nullptr);
            __syncthreads();
            if (tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
            if (w0.ok) restart();
            else __syncthreads();
        } else if ((tiles_done & 1) == 0 && is_pow4(tiles_done >> 1)) {
            if (tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
        }
        if (!w0.ok && !episode && grp == 0) __builtin_amdgcn_s_barrier();     // the scan ends: let group 1 catch up
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave < nsortwaves)
        for (int ql = wave; ql < BQ_; ql += nsortwaves) {
            const int64_t qglobal = (int64_t)qb * BQ_ + ql;
            if (qglobal < p.nq)
                compact_query(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane, true, qglobal, split, nullptr);
        }
}
#undef IPX_PP_MFMA
#undef IPX_PP_QUADRANT

}  // namespace ipx
