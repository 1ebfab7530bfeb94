// bf16 scan, "direct corpus operand" variant of the 256 x 256 tile (large batches, one query part).
//
// What was measured first (round 2, scripts/bf16_variants.py): a wave gets one 1-KiB LDS-DMA piece through per
// ~270-290 ns whatever else it does, so an 8-wave workgroup that copies BOTH operands through LDS (8 pieces per
// wave per 64 features) cannot go below ~2.2 us per 64 features (scan_bf16_kernel<4>, and a ping-pong kernel with
// both operands in LDS: 98 ms), and the 16-wave kernel, which halves the pieces per wave, is held back by its
// waves running in phase (every wave of a SIMD waits for LDS, the barrier and its copy issue at the same time).
//
// This kernel:
//   * the corpus operand never touches LDS. Wave w owns docs [32 w, +32) of the tile and ALL 256 queries
//     (2 x 16 tiles of v_mfma_f32_16x16x32_bf16, 128 accumulator registers); its corpus fragments are private to
//     it, so it loads them straight from HBM into registers: lane (r, g) of a 16-row tile takes chunk g of row r
//     of the packed stage, and the 64 lanes of one global_load_dwordx4 cover one contiguous KiB. Three register
//     sets (one per K-pair of 64 features) keep two K-pairs in flight (mostly L2 hits, see "lockstep" below).
//   * only the query block goes through LDS: 4 pieces per wave per K-pair (half the copy rate), in a ring of
//     three 32-KiB K-pair slots (96 KiB): the vmcnt queue completes in order, so both kinds of op get the same
//     flight time. Queries are packed with a swizzle of their own (slot = chunk ^
//     sigma((row >> 2) & 3), sigma = 0,2,3,1) that makes the 16-row operand reads conflict-free.
//   * ping-pong: two groups of four waves (one wave of each per SIMD) run one barrier apart; a K-pair is four
//     (load, multiply) phase pairs per wave, one per 64-query quarter: 8 ds_read_b128 + one copy piece + one
//     corpus load, then 16 MFMAs. While one group multiplies, the other loads.
// Hazards (barrier intervals i; group 0 is in L0 of K-pair P at i = 0, group 1 at i = 1):
//   * slot P % 3 is read in L0..L3 of K-pair P (i = 0..7; a read completes at the head of the multiply phase, one
//     barrier after its issue, i.e. by i = 8). It is refilled with K-pair P+3, whose pieces are issued at L1, L2,
//     L3 of K-pair P+1 and L0 of K-pair P+2 (i = 10 ...): two barriers after the last read completed.
//   * register set P % 3 is last read by M3 of K-pair P and reloaded from L1 of K-pair P+1 (program order of
//     the wave itself; the loads are issued behind the MFMAs that read the registers).
//   * every load phase issues [copy piece, corpus load]; the ops of K-pair F go out at L1, L2, L3 (cycle F-2) and
//     L0 (cycle F-1). At the end of L3 of cycle P, 6 younger ops follow the last op of K-pair P+1:
//     s_waitcnt vmcnt(6) there means "K-pair P+1 is complete for this wave", and both groups have passed it
//     before the barrier that opens L0 of K-pair P+1.
//   * lockstep. A CU takes in bytes from beyond its XCD's L2 at ~16 GB/s next to ~16 GB/s of L2 hits (measured:
//     2.0 us per K-pair on an idle chip as on a full one; 1.15 us with every load an L2 hit), so most corpus loads
//     have to hit L2. They do while the 3-4 workgroups behind one L2 that scan the same corpus split (xcd_remap
//     puts 3-4 query blocks x all splits on an XCD) run within a few K-pairs of each other: the first one's miss
//     is the others' hit. Left alone they drift apart at the first compaction episodes (measured with every
//     query parked, i.e. no episodes worth the name: 74 ms against 96 ms). So the workgroups of a split
//     re-align: at every episode that all of them reach at the same tile count (1, 4, 16 ... and every
//     SYNC_EVERY tiles) they meet at a counter before the pipeline restarts (group_sync: bounded spin, so a
//     missing partner costs time, never progress; read-only data, so alignment changes timing, never results).
//     (A touch prefetch of the corpus lines, split over those workgroups, was measured as well: slower, 79 ms.)
// Needs the number of K-stages to be a multiple of 6 (the K-pair loop is unrolled by the three register sets);
// the host falls back to scan_bf16_w16_kernel otherwise. Same streaming top-k machinery as the other scans.
#pragma once
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 on a clobber list (the copy pieces set it themselves)

namespace ipx {

struct DXCfg {
    static constexpr int THREADS = 512;
    static constexpr int SLOT = 2 * 256 * 64;            // query block of one K-pair: 32 KiB
    static constexpr int NSET = 3;                       // K-pairs in the ring / corpus register sets
    static constexpr int RING = NSET * SLOT;             // 96 KiB
};

#define IPX_DX_ALOAD(dst, voff, sbase, off)                                                                       \
    do {                                                                                                          \
        if constexpr ((IPX_DX_ABL & 2) == 0)                                                                      \
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #off : "=v"(dst) : "v"(voff), "s"(sbase) : "memory"); \
        else                                                                                                      \
            asm volatile("s_nop 0" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");                               \
    } while (0)
#define IPX_DX_READ(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define IPX_DX_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(A_), as_bf16x8(B_), C_, 0, 0, 0)

// IPX_DX_ABL (diagnostic builds only; results are meaningless, every query is parked so nothing is appended):
// bit 0 = no copy pieces, bit 1 = no corpus loads, bit 2 = corpus loads always from the same two tiles (L2 hits),
// bit 3 = no group sync (this one keeps the results), bit 4 = every query parked, nothing else changed.
#ifndef IPX_DX_ABL
#define IPX_DX_ABL 0
#endif
// Meeting point of the workgroups that scan the same corpus split (one per query block): the k-th call returns
// once all `n` of them have made their k-th call, or after ~100 us (a partner that is late costs time, never
// progress). One thread calls it; the counter only grows and is zeroed by the host per search.
__device__ __forceinline__ void group_sync(uint32_t* ctr, uint32_t target) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
    for (int spin = 0; spin < 256; ++spin) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
        __builtin_amdgcn_s_sleep(16);
    }
}

template <bool SETPRIO>
__global__ void __launch_bounds__(512)
scan_bf16_dx_kernel(const ScanParams p, const int nsortwaves) {
    constexpr int BQ_ = 256, SLOT = DXCfg::SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int scratch_bytes = nsortwaves * p.cap * 8;
    const int stage_region = scratch_bytes > DXCfg::RING ? scratch_bytes : DXCfg::RING;
    int* cnts = (int*)(smem + stage_region);
    float* taus = (float*)(smem + stage_region + BQ_ * 4);
    int* flags = (int*)(smem + stage_region + 2 * BQ_ * 4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r15 = lane & 15, g4 = lane >> 4;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    int qb, split;
    decode_block(p.map, p.S, (int)gridDim.x / p.S, logical, qb, split);

    if (tid < BQ_) {
        cnts[tid] = 0;
        taus[tid] = ((IPX_DX_ABL & 23) == 0 && (int64_t)qb * BQ_ + tid < p.nq) ? -FLT_MAX : INFINITY;
    }
    if (tid < 2) flags[tid] = 0;
    uint32_t* mypub = p.S > 1 ? p.pub + ((size_t)qb * p.S + split) * BQ_ : nullptr;
    const int fill_limit = p.cap - 2 * HBD;
    uint64_t* myqueues = p.cand + (size_t)blockIdx.x * BQ_ * p.cap;
    uint64_t* scratch = (uint64_t*)smem + (size_t)wave * p.cap;

    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // query operand, lane (r15, g4): chunk g4 of row r15 of a 16-row tile, query swizzle
    const uint32_t b_rel = (uint32_t)(r15 * 64 + ((g4 ^ dx_sigma((r15 >> 2) & 3)) * 16));
    const uint32_t bb0 = smem_base + b_rel, bb1 = bb0 + 2 * SLOT;       // ring slots 0, 1 / 2, 3 by immediates
    // corpus operand: byte offset of this lane's 16 bytes inside a packed K-stage (corpus swizzle), row tile 0
    const uint32_t a_voff = (uint32_t)((wave * 32 + r15) * 64 + ((g4 ^ ((r15 >> 2) & 3)) * 16));
    const uint32_t tau_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)taus + r15 * 4;

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
        w.ok = false;          // stays on its last K-pair: the dummy ops go to slots / registers nobody reads
    };
    // op `pi` (0..3) of K-pair w: copy piece wave + 8 pi of the K-pair's 32-KiB query block into ring slot `slot`
    const uint32_t q_voff = (uint32_t)(wave * 1024 + lane * 16);
    auto uniform_ptr = [](const char* ptr) {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    };
    // (issued from asm like the corpus loads: hipcc then tracks no vector-memory op in the loop and adds no
    // vmcnt(0) of its own; M0 = LDS address of the piece, lane l lands at + 16 l)
    auto issue_piece = [&](const Walker& w, int pi, int slot) {
        const char* g = uniform_ptr((const char*)p.Qp + ((size_t)qb * p.nks + 2 * w.kp + (pi >> 1)) * (256 * 64) + (pi & 1) * 8192);
        const uint32_t l = __builtin_amdgcn_readfirstlane(smem_base + slot * SLOT + (wave + 8 * pi) * 1024);
        if constexpr ((IPX_DX_ABL & 1) == 0)
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(q_voff), "s"(g) : "memory", "m0");
        else
            asm volatile("s_nop 0" :: "s"(l), "v"(q_voff), "s"(g) : "memory");
    };
    // ... and the base of K-stage (pi >> 1) of corpus tile w.t for the corpus load (row tile pi & 1 by an immediate)
    auto a_base = [&](const Walker& w, int pi) {
        const int t_ = (IPX_DX_ABL & 4) ? (w.t & 1) : w.t;       // diagnostic: every corpus load an L2 hit
        return uniform_ptr((const char*)p.Xp + ((size_t)t_ * p.nks + 2 * w.kp + (pi >> 1)) * HSTAGE_A);
    };

    Walker w0{split * R, 0, (split + 1) * R < p.ntiles ? (split + 1) * R : p.ntiles, split, split * R < p.ntiles};
    if (!w0.ok) w0.t = 0;
    Walker wf;                 // K-pair whose ops are being issued
    // corpus operand registers: set (K-pair & 3), K-stage of the pair, row tile
    f32x4 A[3][2][2];
    f32x4 Bs[2][4];
    f32x4 acc[2][16];

#define IPX_DX_OPS(W_, PI_, SLOT_, SET_)                                                                   \
    do {                                                                                                    \
        issue_piece(W_, PI_, SLOT_);                                                                        \
        const char* ab_ = a_base(W_, PI_);                                                                  \
        if ((PI_ & 1) == 0) IPX_DX_ALOAD(A[SET_][(PI_) >> 1][0], a_voff, ab_, 0);                           \
        else IPX_DX_ALOAD(A[SET_][(PI_) >> 1][1], a_voff, ab_, 1024);                                       \
    } while (0)

    // (re)start at w0 (always the first K-pair of a tile: set 0, slot 0): all ops of K-pair w0, ops 0..2 of w0+1;
    // then K-pair w0 has landed
    auto restart = [&]() {
        if (!w0.ok) return;
        wf = w0;
        IPX_DX_OPS(wf, 0, 0, 0); IPX_DX_OPS(wf, 1, 0, 0); IPX_DX_OPS(wf, 2, 0, 0); IPX_DX_OPS(wf, 3, 0, 0);
        advance(wf);
        IPX_DX_OPS(wf, 0, 1, 1); IPX_DX_OPS(wf, 1, 1, 1); IPX_DX_OPS(wf, 2, 1, 1);
        asm volatile("s_waitcnt vmcnt(6)"
                     : "+v"(A[0][0][0]), "+v"(A[0][0][1]), "+v"(A[0][1][0]), "+v"(A[0][1][1]) :: "memory");
        __syncthreads();
        if (grp == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind
    };
    restart();

    // one (load, multiply) phase pair: quarter Q_ of K-pair set U_; the ops of this phase belong to K-pair wf,
    // op PI_, ring slot / register set FS_
#define IPX_DX_READS(U_, Q_)                                                                                               \
    {                                                                                                                       \
        const uint32_t bb_ = ((U_) & 2) ? bb1 : bb0;      /* ring slots 0, 1 / 2 */                                          \
        constexpr int o_ = ((U_) & 1) * SLOT + (Q_) * 4096;                                                                 \
        IPX_DX_READ(Bs[0][0], bb_, o_);         IPX_DX_READ(Bs[0][1], bb_, o_ + 1024);                                      \
        IPX_DX_READ(Bs[0][2], bb_, o_ + 2048);  IPX_DX_READ(Bs[0][3], bb_, o_ + 3072);                                      \
        IPX_DX_READ(Bs[1][0], bb_, o_ + 16384); IPX_DX_READ(Bs[1][1], bb_, o_ + 17408);                                     \
        IPX_DX_READ(Bs[1][2], bb_, o_ + 18432); IPX_DX_READ(Bs[1][3], bb_, o_ + 19456);                                     \
    }
#define IPX_DX_MULT(U_, Q_)                                                                                                \
    {                                                                                                                       \
        __builtin_amdgcn_s_barrier();                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                                 \
                     : "+v"(Bs[0][0]), "+v"(Bs[0][1]), "+v"(Bs[0][2]), "+v"(Bs[0][3]), "+v"(Bs[1][0]), "+v"(Bs[1][1]),      \
                       "+v"(Bs[1][2]), "+v"(Bs[1][3]) :: "memory");                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                                    \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                              \
                IPX_DX_MFMA(A[U_][s_][0], Bs[s_][j_], acc[0][4 * (Q_) + j_]);                                               \
                IPX_DX_MFMA(A[U_][s_][1], Bs[s_][j_], acc[1][4 * (Q_) + j_]);                                               \
            }                                                                                                               \
        if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                       \
    }
#define IPX_DX_KPAIR(U_)                                                                                                   \
    {                                                                                                                       \
        IPX_DX_READS(U_, 0)                                                                                                 \
        IPX_DX_OPS(wf, 3, ((U_) + 1) % 3, ((U_) + 1) % 3);                                                                  \
        advance(wf);                                                                                                        \
        IPX_DX_MULT(U_, 0)                                                                                                  \
        IPX_DX_READS(U_, 1)                                                                                                 \
        IPX_DX_OPS(wf, 0, ((U_) + 2) % 3, ((U_) + 2) % 3);                                                                  \
        IPX_DX_MULT(U_, 1)                                                                                                  \
        IPX_DX_READS(U_, 2)                                                                                                 \
        IPX_DX_OPS(wf, 1, ((U_) + 2) % 3, ((U_) + 2) % 3);                                                                  \
        IPX_DX_MULT(U_, 2)                                                                                                  \
        IPX_DX_READS(U_, 3)                                                                                                 \
        IPX_DX_OPS(wf, 2, ((U_) + 2) % 3, ((U_) + 2) % 3);                                                                  \
        asm volatile("s_waitcnt vmcnt(6)"                                                                                   \
                     : "+v"(A[((U_) + 1) % 3][0][0]), "+v"(A[((U_) + 1) % 3][0][1]), "+v"(A[((U_) + 1) % 3][1][0]),         \
                       "+v"(A[((U_) + 1) % 3][1][1]) :: "memory");                                                          \
        IPX_DX_MULT(U_, 3)                                                                                                  \
        advance(w0);                                                                                                        \
    }

    // The destinations of loads in flight (corpus sets of K-pairs past the end of the scan) are
    // dead to the compiler once nothing reads them any more, and it would hand the registers to other values while
    // the loads can still land in them: behind every full wait, pretend to read them all.
#define IPX_DX_KEEP_LIVE()                                                                                                \
    asm volatile("" :: "v"(A[0][0][0]), "v"(A[0][0][1]), "v"(A[0][1][0]), "v"(A[0][1][1]), "v"(A[1][0][0]), "v"(A[1][0][1]), \
                       "v"(A[1][1][0]), "v"(A[1][1][1]), "v"(A[2][0][0]), "v"(A[2][0][1]), "v"(A[2][1][0]), "v"(A[2][1][1]))

    constexpr int SYNC_EVERY = 1 << 20;     // tiles between two meetings of a split's workgroups beyond 1, 4, 16 ... (an episode costs ~100 us)
    const int nqb = (int)gridDim.x / p.S;
    uint32_t* sync_ctr = p.run_ctr + nqb + split;
    uint32_t syncs = 0;
    int tiles_done = 0;
    // optional trace (ipx_set_param "trace"): wave 0's cycles in the append path and in episodes
    uint64_t tr_begin = 0, tr_hit = 0, tr_epi = 0;
    uint32_t tr_hits = 0, tr_episodes = 0;
    if (p.trace) tr_begin = __builtin_readcyclecounter();
    while (w0.ok) {
        const int t = w0.t;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kp = 0; kp < nkp; kp += 3) {
            IPX_DX_KPAIR(0)
            IPX_DX_KPAIR(1)
            IPX_DX_KPAIR(2)
        }

        // ---- epilogue: threshold filter, rare append to the per-query queues ----
        ++tiles_done;
        bool episode = __builtin_amdgcn_readfirstlane(lds_read_i32_nodma(&flags[(tiles_done + 1) & 1])) != 0;
        int* myflag = &flags[tiles_done & 1];
        float tau[16];
        asm volatile(
            "ds_read_b32 %0, %16 offset:0\n\tds_read_b32 %1, %16 offset:64\n\tds_read_b32 %2, %16 offset:128\n\t"
            "ds_read_b32 %3, %16 offset:192\n\tds_read_b32 %4, %16 offset:256\n\tds_read_b32 %5, %16 offset:320\n\t"
            "ds_read_b32 %6, %16 offset:384\n\tds_read_b32 %7, %16 offset:448\n\tds_read_b32 %8, %16 offset:512\n\t"
            "ds_read_b32 %9, %16 offset:576\n\tds_read_b32 %10, %16 offset:640\n\tds_read_b32 %11, %16 offset:704\n\t"
            "ds_read_b32 %12, %16 offset:768\n\tds_read_b32 %13, %16 offset:832\n\tds_read_b32 %14, %16 offset:896\n\t"
            "ds_read_b32 %15, %16 offset:960\n\ts_waitcnt lgkmcnt(0)"
            : "=&v"(tau[0]), "=&v"(tau[1]), "=&v"(tau[2]), "=&v"(tau[3]), "=&v"(tau[4]), "=&v"(tau[5]), "=&v"(tau[6]),
              "=&v"(tau[7]), "=&v"(tau[8]), "=&v"(tau[9]), "=&v"(tau[10]), "=&v"(tau[11]), "=&v"(tau[12]), "=&v"(tau[13]),
              "=&v"(tau[14]), "=&v"(tau[15])
            : "v"(tau_addr) : "memory");
        bool hit = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float m = fmaxf(fmaxf(acc[0][j][0], acc[0][j][1]), fmaxf(acc[0][j][2], acc[0][j][3]));
            m = fmaxf(m, fmaxf(fmaxf(acc[1][j][0], acc[1][j][1]), fmaxf(acc[1][j][2], acc[1][j][3])));
            hit = hit || m > tau[j];
        }
        if (__any(hit)) {
            uint64_t tr0 = 0;
            if (p.trace) { tr0 = __builtin_readcyclecounter(); ++tr_hits; }
            // rare after warm-up. Everything this path needs is derived from a value the compiler cannot see
            // through, so none of it is computed (and kept in registers) outside the branch.
            int r15o = r15, g4o = g4;
            asm volatile("" : "+v"(r15o), "+v"(g4o));
            const int docbase = t * HBD + wave * 32 + 4 * g4o;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ql = j * 16 + r15o;
                float m = fmaxf(fmaxf(acc[0][j][0], acc[0][j][1]), fmaxf(acc[0][j][2], acc[0][j][3]));
                m = fmaxf(m, fmaxf(fmaxf(acc[1][j][0], acc[1][j][1]), fmaxf(acc[1][j][2], acc[1][j][3])));
                if (__any(m > tau[j])) {
                    uint64_t* qu = myqueues + (size_t)ql * p.cap;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sc = acc[i][j][r];
                            const int64_t doc = docbase + i * 16 + r;
                            if (sc > tau[j] && doc < p.N) {
                                const int pos = lds_atomic_inc_nodma(&cnts[ql]);
                                // (stored from asm: a store the compiler knows of would make it wait for vmcnt(0)
                                // somewhere behind it, and that drains the loads in flight)
                                if (pos < p.cap) {
                                    const uint64_t key = make_key(sc, (uint32_t)doc);
                                    asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(qu + pos), "v"(key) : "memory");
                                } else {
                                    const int one = 1;
                                    asm volatile("global_store_dword %0, %1, off" :: "v"(p.err), "v"(one) : "memory");
                                }
                                if (pos >= fill_limit) lds_write_i32_nodma(myflag, 1);
                            }
                        }
                }
            }
            if (p.trace) tr_hit += __builtin_readcyclecounter() - tr0;
        }
        const bool meet = is_pow4(tiles_done) || tiles_done % SYNC_EVERY == 0;     // every workgroup of the split is here
        episode = episode || meet;
        if (episode) {
            uint64_t tr0 = 0;
            if (p.trace) { tr0 = __builtin_readcyclecounter(); ++tr_episodes; }
            // group 1 is one barrier behind: group 0 waits for it here, then the selection scratch (which
            // aliases the ring) is used by everybody and the pipeline is restarted at w0
            if (grp == 0) __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            IPX_DX_KEEP_LIVE();
            __syncthreads();
            if (tid < 2) flags[tid] = 0;
            if (wave < nsortwaves)
                for (int ql = wave; ql < BQ_; ql += nsortwaves)
                    compact_query_select(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane,
                                         mypub ? mypub + ql : nullptr);
            __syncthreads();
            if (tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
            if ((IPX_DX_ABL & 8) == 0 && meet && nqb > 1) {
                ++syncs;
                if (tid == 0) group_sync(sync_ctr, syncs * (uint32_t)nqb);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (w0.ok) restart();
            else __syncthreads();
            if (p.trace) tr_epi += __builtin_readcyclecounter() - tr0;
        } else if ((tiles_done & 1) == 0 && is_pow4(tiles_done >> 1)) {
            if (tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        if (!w0.ok && !episode && grp == 0) __builtin_amdgcn_s_barrier();     // the scan ends: let group 1 catch up
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    IPX_DX_KEEP_LIVE();
    if (p.trace && tid == 0) {
        // [gridDim.x][4]: total cycles, append-path cycles, episode cycles, tiles with appends << 32 | episodes
        uint64_t* tr = p.trace + (size_t)blockIdx.x * 4;
        tr[0] = __builtin_readcyclecounter() - tr_begin;
        tr[1] = tr_hit;
        tr[2] = tr_epi;
        tr[3] = ((uint64_t)tr_hits << 32) | tr_episodes;
    }
    __syncthreads();
    if (wave < nsortwaves)
        for (int ql = wave; ql < BQ_; ql += nsortwaves) {
            const int64_t qglobal = (int64_t)qb * BQ_ + ql;
            if (qglobal < p.nq)
                compact_query(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane, true, qglobal, split, nullptr);
        }
}
#undef IPX_DX_KEEP_LIVE
#undef IPX_DX_ALOAD
#undef IPX_DX_MFMA
#undef IPX_DX_OPS
#undef IPX_DX_READS
#undef IPX_DX_MULT
#undef IPX_DX_KPAIR
#undef IPX_DX_READ

}  // namespace ipx
#pragma clang diagnostic pop
