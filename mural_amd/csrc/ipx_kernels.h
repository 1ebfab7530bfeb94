// ipx_kernels.h — device kernels of the gfx950 exact inner-product search engine.
//
// Replaces the arithmetic faiss does for tevatron.faiss_retriever
// (IndexFlat::search -> knn_inner_product + heap top-k, called from
// src/tevatron/faiss_retriever/retriever.py:19, and ResultHeap in reducer.py:15-18).
//
// Data layout in HBM (chosen for the scan kernel, built once by pack_rows_kernel):
//   rows are grouped in tiles of R rows (R = 128 for the corpus, 256 for a query block);
//   a tile is cut along the feature axis into K-stages of 16 floats; one stage of one
//   tile is a contiguous block  [R rows][4 chunks][4 floats]  (R*64 bytes), and the 16-byte
//   chunk c of row r is stored at chunk slot  c ^ ((r >> 2) & 3).
//   => a stage is copied HBM -> LDS by linear 1-KiB global_load_lds_dwordx4 pieces, and the
//      ds_read_b128 operand reads of the MFMA loop are bank-conflict free (16 lanes of a
//      read group hit 16 different 16-byte slots of the 256-byte bank row).
//   d is zero-padded to a multiple of 16, the row count to a multiple of R.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <float.h>

namespace ipx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BD = 128;       // corpus rows (docs) per score tile
constexpr int BQ = 256;       // queries per workgroup
constexpr int BK = 16;        // floats per K-stage
constexpr int NTHREADS = 256; // 4 waves: 2 (docs) x 2 (queries), each 64 docs x 128 queries
constexpr int STAGE_A_BYTES = BD * BK * 4;                 // 8 KiB
constexpr int STAGE_B_BYTES = BQ * BK * 4;                 // 16 KiB
constexpr int STAGE_BYTES = STAGE_A_BYTES + STAGE_B_BYTES; // 24 KiB
constexpr int LDS_STATE_BYTES = 2 * BQ * 4 + 16;           // queue counters + thresholds + 2 flags + run slot

// ---- order-preserving key for (score, row) pairs -------------------------------------------
// key = ordered(score) << 32 | ~row : larger key = better hit (score desc, row asc).
// ordered() maps float bits to unsigned so that unsigned order == float order; every real
// score maps to a value > 0, so key 0 is the "empty slot" sentinel.
__device__ __forceinline__ uint32_t f32_ordered(float x) {
    uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)f32_ordered(score) << 32) | (uint32_t)(~row);
}

// LDS traffic inside one wave needs no s_barrier: the LDS unit executes a wave's DS
// instructions in issue order. This only stops the compiler from moving or caching them.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// LDS accesses to the small per-query state (counters, thresholds) while an LDS-DMA stage copy
// is in flight. hipcc cannot prove that a DS access does not alias the DMA destination, so it
// puts s_waitcnt vmcnt(0) in front of every one of them — which also waits for the candidate
// stores issued just before (vmcnt counts stores) and serialises the epilogue on HBM round
// trips. These never touch the stage buffers, so they are issued from asm with their own
// lgkmcnt wait (the loads and the wait in ONE statement, outputs early-clobber).
__device__ __forceinline__ float lds_read_f32_nodma(const float* p) {
    float v;
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ int lds_atomic_inc_nodma(int* p) {
    int old;
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)p;
    const int one = 1;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(a), "v"(one) : "memory");
    return old;
}
__device__ __forceinline__ void lds_atomic_max_u32_nodma(uint32_t* p, uint32_t v) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)p;
    asm volatile("ds_max_u32 %0, %1" : : "v"(a), "v"(v) : "memory");
}

__device__ __forceinline__ int lds_read_i32_nodma(const int* p) {
    int v;
    const uint32_t a = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) int*)p;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_write_i32_nodma(int* p, int v) {
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)p;
    asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(v) : "memory");
}

// Largest float that compares strictly below x (x finite or +inf, not NaN): s > below(x) <=> s >= x.
__device__ __forceinline__ float f32_below(float x) {
    const uint32_t u = __float_as_uint(x);
    if (x == 0.f) return __uint_as_float(0x80000001u);      // both zeros compare equal
    return __uint_as_float(x > 0.f ? u - 1 : u + 1);
}

__device__ __forceinline__ bool is_pow4(int n) { return n > 0 && (n & (n - 1)) == 0 && (n & 0x55555555); }
// Workgroup id -> logical id (which query block / split it is). For speed only; any bijection
// is correct. Observed dispatch: blocks b and b+8 share an XCD (and its L2); on an XCD the
// first 32 blocks get a CU each and the next 32 become the second workgroup of those CUs, where
// the earlier-dispatched workgroup wins most MFMA issue slots. So (1) each XCD gets a contiguous
// run of logical ids: the workgroups of one query block sit behind one L2; (2) inside an XCD,
// consecutive logical ids alternate first / second workgroup of a CU: every query block gets
// the same mix of fast and slow workgroups, and the two on a CU read the same query tiles.
__device__ __forceinline__ int xcd_remap(int b, int G) {
    const int q = G >> 3, r = G & 7, x = b & 7;
    const int n_x = q + (x < r ? 1 : 0);          // blocks on this XCD
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    int j = b >> 3;                                 // dispatch order on this XCD
    if (n_x > 32 && n_x <= 64) {
        const int m = n_x - 32;                     // CUs that get a second workgroup
        j = j >= 32 ? 2 * (j - 32) + 1 : (j < m ? 2 * j : 2 * m + (j - m));
    }
    return base + j;
}

// Logical workgroup id -> (query block, corpus split) of the large-batch bf16 scans. Consecutive logical ids sit on
// one XCD (xcd_remap), so the grouping decides what that XCD's L2 sees. gsz = query blocks per group:
//   0: query-block-major — an XCD holds few query blocks x all splits (its query operands stay in L2, every
//      workgroup streams its own part of the corpus);
//   gsz >= 1: groups of gsz query blocks, split-major inside a group — an XCD holds a few splits x gsz query blocks:
//      the gsz workgroups of one split stream the same corpus tiles (gsz = all query blocks: one or two splits per
//      XCD, but then the query operands of all blocks no longer fit its L2).
__device__ __forceinline__ void decode_block(int gsz, int S, int nqb, int logical, int& qb, int& split) {
    if (gsz <= 0) { qb = logical / S; split = logical - qb * S; return; }
    const int per = gsz * S, full = nqb / gsz;
    if (logical < full * per) {
        const int g = logical / per, w = logical - g * per;
        split = w / gsz;
        qb = g * gsz + (w - split * gsz);
    } else {                                        // the last, smaller group
        const int gl = nqb - full * gsz, w = logical - full * per;
        split = w / gl;
        qb = full * gsz + (w - split * gl);
    }
}

// ---- pack_rows: row-major [n, d] (f32 / f16 / bf16) -> tiled, padded, swizzled fp32 ----------
// One thread per (row, 16-byte chunk). Rows [n, n_pad) and features [d, 16*nks) are zeros.
// With A > 0 the source row is the aspect-fused query  sum_a w[r][a] * src[r][a][:]
// (MURAL, modeling/mul_grain.py:97-110); with logits != 0 the weights are soft-maxed first.
__device__ __forceinline__ float load_as_f32(const void* p, int dtype, int64_t i) {
    if (dtype == 0) return ((const float*)p)[i];
    if (dtype == 1) return __half2float(((const __half*)p)[i]);
    return __uint_as_float(((uint32_t)((const uint16_t*)p)[i]) << 16);
}

__global__ void __launch_bounds__(256)
pack_rows_kernel(const void* __restrict__ src, int dtype, int64_t n, int64_t n_pad, int d,
                 float* __restrict__ dst, int64_t row0, int rows_per_tile, int nks,
                 const float* __restrict__ w, int A, int logits) {
    const int nchunks = nks * 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / nchunks;
    const int c = (int)(gid - r * nchunks);
    if (r >= n_pad) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < n) {
        const int k0 = c * 4;
        if (A <= 0) {
            const int64_t base = r * (int64_t)d;
            if (dtype == 0 && (d & 3) == 0 && k0 + 3 < d) {
                v = *(const f32x4*)((const float*)src + base + k0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k0 + e < d) v[e] = load_as_f32(src, dtype, base + k0 + e);
            }
        } else {
            // aspect fusion: weights (optionally softmax of logits), then a fused-multiply-add
            // chain over the aspects in index order.
            float mx = -INFINITY, den = 0.f;
            if (logits) {
                for (int a = 0; a < A; ++a) mx = fmaxf(mx, w[r * A + a]);
                for (int a = 0; a < A; ++a) den += expf(w[r * A + a] - mx);
            }
            for (int a = 0; a < A; ++a) {
                float wa = w[r * A + a];
                if (logits) wa = expf(wa - mx) / den;
                const int64_t base = (r * A + a) * (int64_t)d;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k0 + e < d) v[e] = fmaf(wa, load_as_f32(src, dtype, base + k0 + e), v[e]);
            }
        }
    }
    const int64_t R = row0 + r;
    const int64_t tile = R / rows_per_tile;
    const int rr = (int)(R - tile * rows_per_tile);
    const int ks = c >> 2, pc = (c & 3) ^ ((rr >> 2) & 3);
    float* o = dst + ((tile * nks + ks) * rows_per_tile + rr) * 16 + pc * 4;
    *(f32x4*)o = v;
}

// ---- scan: fused fp32-MFMA score tiles + streaming per-query top-k ---------------------------
struct ScanParams {
    const float* Xp;     // packed corpus  [ntiles][nks][128][16]
    const float* Qp;     // packed queries [nqb][nks][256][16]
    uint64_t* cand;      // candidate queues [gridDim.x][256][cap] (keys)
    float* outD;         // [nq][S][k]  sorted partial (or final, when S == 1) scores
    int64_t* outI;       // [nq][S][k]  ids (id_base + row), -1 for empty slots
    int* err;            // set to 1 if a queue would overflow (invariant violation)
    uint64_t* trace;     // optional [gridDim.x][4]: start, end (100 MHz ticks), HW_ID, compactions
    uint32_t* pub;       // [nqb][S][256] ordered(score of the r-th best hit of that split), 0 = none yet
    uint32_t* run_ctr;   // [nqb] runs handed out beyond the S pre-assigned ones (zeroed per search)
    int64_t N;           // rows in the index
    int64_t id_base;
    int nq, nks, k, cap, S, ntiles;
    int r;               // ceil(k / S): rank each split publishes (S * r >= k)
    int run_tiles;       // R: tiles per run of the dynamic corpus split
    int S_out;           // lists per query in outD / outI (S, or S + helpers in scan_f32_kernel)
    int helpers;         // scan_f32_kernel: list slots per query block for workgroups that finished their own block
    uint32_t* help_ctr;  // [nqb] helper slots handed out (zeroed per search)
    // Deep searches (k beyond IPX_MAX_K, one pass per IPX_MAX_K ranks; scan_f32_kernel<true> only): query q admits only
    // hits strictly after (ceilD[q], ceilI[q]) in the result order — the last hit the previous pass returned;
    // ceilI[q] < 0: that pass ran out of hits, nothing is admitted.
    const float* ceilD = nullptr;
    const int64_t* ceilI = nullptr;
    int map = 0;         // bf16 large-batch scans: query blocks per group of decode_block (0 = query-block-major)
    int nselect = 0;     // scan_bf16_w16_kernel: waves that run the episodes' selections (4 bytes of LDS per queue slot
                         // each; the final sort takes 8: `nsortwaves`); 0 = nsortwaves
    // Deep lists (k beyond what one wave sorts in LDS: IPX_MAX_K < k <= DEEP_MAX_K; scan_f32_kernel<*, true> and the
    // 16-wave bf16 scan): a split's final list is not sorted in the kernel — its best k keys go, in any order, to
    // deepK[(q * S_out + split) * k ...] with their number in deepN[q * S_out + split], and deep_finish_kernel selects
    // and sorts across the splits.
    uint64_t* deepK = nullptr;
    int* deepN = nullptr;
};

// Side list of a selection (nth_largest_lds) per selecting wave when the scratch is 4 bytes per queue slot.
constexpr int SEL_AUX = 1024;
// Deepest list one pass of a scan keeps per query (queue capacity 2 * DEEP_MAX_K slots: one wave's selection scratch,
// 4 bytes per slot + the side list, is 132 KiB of the CU's 160 KiB LDS).
constexpr int DEEP_MAX_K = 16384;

// bitonic sort, descending, of P (power of two) 64-bit keys in LDS by one wave
__device__ __forceinline__ void wave_sort_desc(uint64_t* keys, int P, int lane) {
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = lane; i < (P >> 1); i += 64) {
                const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t x = keys[lo], y = keys[hi];
                if ((x < y) == desc) { keys[lo] = y; keys[hi] = x; }
            }
            wave_lds_sync();
        }
    }
}

// One wave reduces query `ql`'s queue to its best min(n, k) keys (sorted), updates the
// threshold, and — on the final call — writes the sorted result row.
__device__ __forceinline__ void compact_query(const ScanParams& p, uint64_t* scratch, int* cnts,
                                              float* taus, uint64_t* queue, int ql, int lane,
                                              bool final_pass, int64_t qglobal, int split, uint32_t* pub_slot) {
    const int n = cnts[ql];
    // nothing to drop and nothing to publish: leave the queue as it is
    if (!final_pass && n <= p.k && !(pub_slot && n >= p.r)) return;
    int P = 2, nkeep = n;
    if (final_pass) {
        // Only entries at or above the threshold can be in the result (it is this split's k-th best score or the
        // union bound, whichever is larger; an entry equal to it may be the k-th best itself). Entries admitted
        // early, against a weaker threshold, are dropped here: with many splits per query (small batches) a queue
        // ends with hundreds of entries of which a handful count, and the sort below shrinks accordingly.
        const uint32_t cut = f32_ordered(taus[ql]);
        int base = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const uint64_t key = i < n ? queue[i] : 0ull;
            const bool keep = i < n && (uint32_t)(key >> 32) >= cut;
            const uint64_t mask = __ballot(keep);
            if (keep) scratch[base + __popcll(mask & ((1ull << lane) - 1))] = key;
            base += __popcll(mask);
        }
        nkeep = base;
        while (P < nkeep) P <<= 1;
        for (int i = nkeep + lane; i < P; i += 64) scratch[i] = 0ull;
    } else {
        while (P < n) P <<= 1;
        for (int i = lane; i < P; i += 64) scratch[i] = (i < n) ? queue[i] : 0ull;
    }
    wave_lds_sync();
    wave_sort_desc(scratch, P, lane);
    const int m = nkeep < p.k ? nkeep : p.k;
    if (!final_pass) {
        for (int i = lane; i < m; i += 64) queue[i] = scratch[i];
        if (lane == 0) {
            cnts[ql] = m;
            // local threshold: the k-th best of this split so far (strict: later rows have larger ids)
            if (n >= p.k) taus[ql] = fmaxf(taus[ql], f32_unordered((uint32_t)(scratch[p.k - 1] >> 32)));
            // publish this split's r-th best for the union threshold (see poll_union_threshold)
            if (pub_slot && n >= p.r)
                __hip_atomic_store(pub_slot, (uint32_t)(scratch[p.r - 1] >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        float* oD = p.outD + ((size_t)qglobal * p.S_out + split) * p.k;
        int64_t* oI = p.outI + ((size_t)qglobal * p.S_out + split) * p.k;
        for (int i = lane; i < p.k; i += 64) {
            if (i < m) {
                const uint64_t key = scratch[i];
                oD[i] = f32_unordered((uint32_t)(key >> 32));
                oI[i] = p.id_base + (int64_t)(uint32_t)(~(uint32_t)key);
            } else {
                oD[i] = -FLT_MAX;
                oI[i] = -1;
            }
        }
    }
    wave_lds_sync();
}

// Non-final compaction by SELECTION instead of sorting (used by the bf16 scans, whose tiles are
// 8x shorter than the fp32 kernel's, so episodes weigh 8x more): one wave keeps query `ql`'s best
// k keys of the n in its queue (in any order), raises the threshold to the k-th best score and
// publishes the r-th best. The keys' high words (ordered scores) go to the wave's LDS scratch;
// the k-th largest is found bit by bit from the top — 32 steps of "how many are >= candidate",
// counted with wave ballots (a scalar count, no cross-lane reduction); ties at the cut are
// resolved the same way on the low words (row ids); the survivors move to the front of the queue
// in place (every position written lies below every position still to be read). All memory
// traffic goes in batches of 8 independent accesses per lane: the cost is latency, not volume.
// count of entries >= cand among sh[0 .. 64 * kpl8), by wave ballots (kpl8 a multiple of 8)
__device__ __forceinline__ int count_ge_lds(const uint32_t* sh, int kpl8, int lane, uint32_t cand) {
    int cnt = 0;
#pragma unroll 1
    for (int j0 = 0; j0 < kpl8; j0 += 8) {
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = sh[lane + 64 * (j0 + u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt += __popcll(__ballot(x[u] >= cand));
    }
    return cnt;
}

// The need-th largest of sh[0 .. 64 * kpl8) (empty slots are 0), bit by bit from the top. Two levels: the upper 16
// bits are decided over all entries; then only the entries that share those 16 bits can still matter — they are
// copied to `aux` (with the count of the entries above them) and the lower 16 bits are decided over that short list
// (scores of one candidate list lie close together: a few dozen entries per 16-bit bucket out of thousands). If the
// bucket does not fit `aux`, the lower bits are decided over all entries as well. sh is left untouched.
__device__ __forceinline__ uint32_t nth_largest_lds(const uint32_t* sh, int kpl8, int lane, int need, uint32_t* aux, int aux_cap) {
    uint32_t t = 0;
#pragma unroll 1
    for (int b = 31; b >= 16; --b) {
        const uint32_t cand = t | (1u << b);
        if (count_ge_lds(sh, kpl8, lane, cand) >= need) t = cand;
    }
    const uint32_t top = t >> 16;
    int above = 0, m = 0;
#pragma unroll 1
    for (int j = 0; j < kpl8; ++j) {
        const uint32_t x = sh[lane + 64 * j];
        const bool match = (x >> 16) == top;
        above += __popcll(__ballot((x >> 16) > top));
        const uint64_t mask = __ballot(match);
        const int at = m + __popcll(mask & ((1ull << lane) - 1));
        if (match && at < aux_cap) aux[at] = x;
        m += __popcll(mask);
    }
    if (m > aux_cap) {                         // the bucket is too large for the side list: all entries again
#pragma unroll 1
        for (int b = 15; b >= 0; --b) {
            const uint32_t cand = t | (1u << b);
            if (count_ge_lds(sh, kpl8, lane, cand) >= need) t = cand;
        }
        return t;
    }
    wave_lds_sync();
    const int need2 = need - above;            // >= 1: the upper bits were chosen so that fewer than `need` lie above
    const int rounds = (m + 63) >> 6;
#pragma unroll 1
    for (int b = 15; b >= 0; --b) {
        const uint32_t cand = t | (1u << b);
        int cnt = 0;
#pragma unroll 1
        for (int j = 0; j < rounds; ++j) {
            const int i = lane + 64 * j;
            const uint32_t x = i < m ? aux[i] : 0u;
            cnt += __popcll(__ballot(x >= cand));
        }
        if (cnt >= need2) t = cand;
    }
    wave_lds_sync();                           // the side list is free again (the next selection reuses it)
    return t;
}

// wave-wide max / min of a 32-bit value (every lane gets the result)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off); v = o < v ? o : v; }
    return v;
}

// nth_largest_lds for a queue of at most 512 entries held in REGISTERS (x[u] = entry lane + 64 u, 0 = empty slot): the
// same two levels, but the steps over all entries read no memory, and they start below the entries' common prefix —
// vmax / vmin are the largest and smallest non-empty entry (wave-uniform): every entry, so the answer too, shares
// their common leading bits (scores of one candidate list lie close together: sign and exponent, 8-9 steps of 32).
// Measured in round 3 on the headline scan: a selection over LDS took ~29 us (a wave beside MFMA-bound waves gets an
// instruction through every ~40 cycles: what counts is the number of instructions), two per query and episode,
// 4.5 % of a workgroup's time in episodes. need in [1, number of non-empty entries].
__device__ __forceinline__ uint32_t nth_largest_regs8(const uint32_t (&x)[8], int lane, int need, uint32_t vmax, uint32_t vmin,
                                                      uint32_t* aux, int aux_cap) {
    if (vmax == vmin) return vmax;
    const int hb = 31 - __builtin_clz(vmax ^ vmin);               // highest bit in which two entries differ
    uint32_t t = hb >= 31 ? 0u : (vmax >> (hb + 1)) << (hb + 1);  // the common prefix
    int b = hb;
#pragma unroll 1
    for (; b >= 16; --b) {
        const uint32_t cand = t | (1u << b);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) cnt += __popcll(__ballot(x[u] >= cand));
        if (cnt >= need) t = cand;
    }
    // bits 31..16 are decided (or common): the entries that share them go to the side list
    const uint32_t top = t >> 16;
    int above = 0, m = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const bool match = (x[u] >> 16) == top;
        above += __popcll(__ballot((x[u] >> 16) > top));
        const uint64_t mask = __ballot(match);
        const int at = m + __popcll(mask & ((1ull << lane) - 1));
        if (match && at < aux_cap) aux[at] = x[u];
        m += __popcll(mask);
    }
    if (m > aux_cap) {                         // the bucket is too large for the side list: all entries again
#pragma unroll 1
        for (; b >= 0; --b) {
            const uint32_t cand = t | (1u << b);
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) cnt += __popcll(__ballot(x[u] >= cand));
            if (cnt >= need) t = cand;
        }
        return t;
    }
    wave_lds_sync();
    const int need2 = need - above;
    const int rounds = (m + 63) >> 6;
#pragma unroll 1
    for (; b >= 0; --b) {
        const uint32_t cand = t | (1u << b);
        int cnt = 0;
#pragma unroll 1
        for (int j = 0; j < rounds; ++j) {
            const int i = lane + 64 * j;
            const uint32_t y = i < m ? aux[i] : 0u;
            cnt += __popcll(__ballot(y >= cand));
        }
        if (cnt >= need2) t = cand;
    }
    wave_lds_sync();                           // the side list is free again
    return t;
}

// `aux` / `aux_cap`: the selections' side list; nullptr = the second half of a scratch of 8 bytes per queue slot.
// REGS: queues of up to 512 entries are selected in registers (nth_largest_regs8). Only the fp32 headline kernel asks
// for it: its episodes run when the accumulators are dead; in the other kernels the extra live registers cost more than
// the faster selection brings (scan_f32_small_kernel<2> went from 122 to 136 VGPRs, three waves per SIMD instead of
// four, and 15 % slower at 64 queries).
template <bool REGS = false>
__device__ __forceinline__ void compact_query_select(const ScanParams& p, uint64_t* scratch, int* cnts, float* taus,
                                                     uint64_t* queue, int ql, int lane, uint32_t* pub_slot,
                                                     uint32_t* aux = nullptr, int aux_cap = 0, uint64_t* tph = nullptr) {
    if (!aux) { aux = (uint32_t*)scratch + p.cap; aux_cap = p.cap; }
#ifdef IPX_F32_TRACE_HITS
    uint64_t tp0 = __builtin_amdgcn_s_memrealtime(), tp1;
#define IPX_TPH(I_) if (tph) { tp1 = __builtin_amdgcn_s_memrealtime(); tph[I_] += tp1 - tp0; tp0 = tp1; }
#else
#define IPX_TPH(I_)
#endif
    const int n = cnts[ql];
    const bool publish = pub_slot && n >= p.r;
    if (n <= p.k && !publish) return;
    const int kpl8 = (((n + 63) >> 6) + 7) & ~7;         // slots per lane, in batches of 8 (64 * kpl8 <= cap)
    uint32_t* sh = (uint32_t*)scratch;                    // [64 * kpl8] high words, later (in place) the low words of tied keys
    const uint32_t* q32 = (const uint32_t*)queue;
    if (REGS && kpl8 == 8) {
        // up to 512 entries (every episode of a top-100 search): the high words stay in registers
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + 64 * u;
            x[u] = q32[2 * (i < n ? i : n - 1) + 1];
        }
        uint32_t lmax = 0, lmin = 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool valid = lane + 64 * u < n;
            x[u] = valid ? x[u] : 0u;
            lmax = x[u] > lmax ? x[u] : lmax;
            lmin = valid && x[u] < lmin ? x[u] : lmin;
        }
        const uint32_t vmax = wave_max_u32(lmax), vmin = wave_min_u32(lmin);
        IPX_TPH(0)
        if (publish) {
            const uint32_t vr = nth_largest_regs8(x, lane, p.r, vmax, vmin, aux, aux_cap);
            if (lane == 0) __hip_atomic_store(pub_slot, vr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        IPX_TPH(1)
        if (n <= p.k) return;
        const uint32_t v = nth_largest_regs8(x, lane, p.k, vmax, vmin, aux, aux_cap);
        IPX_TPH(2)
        int above = 0, tied = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            above += __popcll(__ballot(x[u] > v));
            tied += __popcll(__ballot(x[u] == v));
        }
        uint32_t lcut = 0;
        if (tied > p.k - above) {              // ties at the cut (rare): the tied keys' low words, through LDS as below
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = lane + 64 * u;
                sh[i] = (i < n && x[u] == v) ? q32[2 * i] : 0u;
            }
            wave_lds_sync();
            lcut = nth_largest_lds(sh, 8, lane, p.k - above, aux, aux_cap);
        }
        int base = 0;
        uint64_t key[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + 64 * u;
            key[u] = queue[i < n ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + 64 * u;
            const uint32_t h = (uint32_t)(key[u] >> 32);
            const bool keep = i < n && (h > v || (h == v && (uint32_t)key[u] >= lcut));
            const uint64_t mask = __ballot(keep);
            if (keep) queue[base + __popcll(mask & ((1ull << lane) - 1))] = key[u];
            base += __popcll(mask);
        }
        if (lane == 0) {
            cnts[ql] = base;                          // == k
            taus[ql] = fmaxf(taus[ql], f32_unordered(v));
        }
        wave_lds_sync();
        IPX_TPH(3)
        return;
    }
#pragma unroll 1
    for (int j0 = 0; j0 < kpl8; j0 += 8) {
        uint32_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + 64 * (j0 + u);
            x[u] = q32[2 * (i < n ? i : n - 1) + 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + 64 * (j0 + u);
            sh[i] = i < n ? x[u] : 0u;
        }
    }
    wave_lds_sync();
    IPX_TPH(0)
    if (publish) {
        const uint32_t vr = nth_largest_lds(sh, kpl8, lane, p.r, aux, aux_cap);
        if (lane == 0) __hip_atomic_store(pub_slot, vr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    IPX_TPH(1)
    if (n <= p.k) return;
    const uint32_t v = nth_largest_lds(sh, kpl8, lane, p.k, aux, aux_cap);      // >= 1: empty slots (0) never pass
    IPX_TPH(2)
    // ties at the cut: only the (k - #above) tied keys with the largest low words survive
    int above = 0, tied = 0;
#pragma unroll 1
    for (int j = 0; j < kpl8; ++j) {
        const uint32_t h = sh[lane + 64 * j];
        above += __popcll(__ballot(h > v));
        tied += __popcll(__ballot(h == v));
    }
    uint32_t lcut = 0;
    if (tied > p.k - above) {
        // in place: the high words are not needed again (the move below reads the keys from the queue), so the
        // scratch of a selection is 4 bytes per queue slot
        uint32_t* sl = sh;
#pragma unroll 1
        for (int j = 0; j < kpl8; ++j) {
            const int i = lane + 64 * j;
            sl[i] = (i < n && sh[i] == v) ? q32[2 * i] : 0u;
        }
        wave_lds_sync();
        lcut = nth_largest_lds(sl, kpl8, lane, p.k - above, aux, aux_cap);
    }
    int base = 0;
#pragma unroll 1
    for (int j0 = 0; j0 < kpl8; j0 += 8) {
        uint64_t key[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + 64 * (j0 + u);
            key[u] = queue[i < n ? i : n - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane + 64 * (j0 + u);
            const uint32_t h = (uint32_t)(key[u] >> 32);
            const bool keep = i < n && (h > v || (h == v && (uint32_t)key[u] >= lcut));
            const uint64_t mask = __ballot(keep);
            if (keep) queue[base + __popcll(mask & ((1ull << lane) - 1))] = key[u];
            base += __popcll(mask);
        }
    }
    if (lane == 0) {
        cnts[ql] = base;                          // == k
        taus[ql] = fmaxf(taus[ql], f32_unordered(v));
    }
    wave_lds_sync();
    IPX_TPH(3)
#undef IPX_TPH
}

// Final pass of a deep list: the queue's best k keys (all of them if there are fewer), filtered by the threshold like
// the sorted final pass, go unsorted to the split's slot of deepK. One wave; `sh` / `aux` as in compact_query_select.
__device__ __forceinline__ void compact_query_deep_final(const ScanParams& p, uint32_t* sh, uint32_t* aux, int* cnts, float* taus,
                                                         uint64_t* queue, int ql, int lane, int64_t qglobal, int split) {
    compact_query_select(p, (uint64_t*)sh, cnts, taus, queue, ql, lane, nullptr, aux, SEL_AUX);     // n > k: keep the best k
    const int n = cnts[ql];
    const uint32_t cut = f32_ordered(taus[ql]);
    uint64_t* o = p.deepK + ((size_t)qglobal * p.S_out + split) * p.k;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const uint64_t key = i < n ? queue[i] : 0ull;
        const bool keep = i < n && (uint32_t)(key >> 32) >= cut;
        const uint64_t mask = __ballot(keep);
        if (keep) o[base + __popcll(mask & ((1ull << lane) - 1))] = key;
        base += __popcll(mask);
    }
    if (lane == 0) p.deepN[(size_t)qglobal * p.S_out + split] = base;
}

// Union threshold. Every split publishes the score of its r-th best hit so far, r = ceil(k/S).
// Taking the minimum v over the S splits of one query: each split holds >= r rows scoring >= v,
// so >= S*r >= k rows of the index score >= v and the global k-th best score is >= v. A row
// scoring below v can therefore never be in the result: thread `tid` raises its query's
// threshold to just below v (rows scoring exactly v may still win on the id tie-break).
// Published values only grow, so a stale read is a weaker but still valid bound; the loads are
// agent-scope (they bypass this CU's L1), no ordering with anything else is needed.
__device__ __forceinline__ void poll_union_threshold(const ScanParams& p, float* taus, int qb, int tid) {
    if (p.S <= 1) return;
    const uint32_t* col = p.pub + (size_t)qb * p.S * BQ + tid;
    uint32_t v = 0xffffffffu;
    for (int s2 = 0; s2 < p.S; ++s2) {
        const uint32_t x = __hip_atomic_load(col + (size_t)s2 * BQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = x < v ? x : v;
    }
    if (v != 0) taus[tid] = fmaxf(taus[tid], f32_below(f32_unordered(v)));
}

__device__ __forceinline__ void issue_stage(const ScanParams& p, char* buf, int64_t tile, int qb,
                                            int ks, int wave, int lane) {
    // A stage: 8 KiB contiguous, 2 x 1 KiB per wave; B stage: 16 KiB contiguous, 4 x 1 KiB per wave.
    const char* ga = (const char*)p.Xp + ((size_t)tile * p.nks + ks) * STAGE_A_BYTES + wave * 2048 + lane * 16;
    const char* gb = (const char*)p.Qp + ((size_t)qb * p.nks + ks) * STAGE_B_BYTES + wave * 4096 + lane * 16;
    char* la = buf + wave * 2048;
    char* lb = buf + STAGE_A_BYTES + wave * 4096;
#define IPX_GLDS(g, l) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g), \
                                                         (__attribute__((address_space(3))) void*)(l), 16, 0, 0)
    IPX_GLDS(ga, la);
    IPX_GLDS(ga + 1024, la + 1024);
    IPX_GLDS(gb, lb);
    IPX_GLDS(gb + 1024, lb + 1024);
    IPX_GLDS(gb + 2048, lb + 2048);
    IPX_GLDS(gb + 3072, lb + 3072);
#undef IPX_GLDS
}

// The same six pieces, each issued from asm in the scalar-base form: M0 = the piece's LDS address, base in scalar
// registers, the lane's 16 bytes as a 32-bit offset. Piece i of 0..5: 0, 1 = corpus, 2..5 = queries.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 on a clobber list (the copy pieces set it themselves)
struct StagePiecesF32 {
    const char* ga;
    const char* gb;
    uint32_t la, lb;
    static __device__ __forceinline__ const char* uniform_ptr(const char* ptr) {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    __device__ __forceinline__ StagePiecesF32(const ScanParams& p, uint32_t lds_buf, int64_t tile, int qb, int ks, int wave) {
        ga = uniform_ptr((const char*)p.Xp + ((size_t)tile * p.nks + ks) * STAGE_A_BYTES + wave * 2048);
        gb = uniform_ptr((const char*)p.Qp + ((size_t)qb * p.nks + ks) * STAGE_B_BYTES + wave * 4096);
        la = __builtin_amdgcn_readfirstlane(lds_buf + wave * 2048);
        lb = __builtin_amdgcn_readfirstlane(lds_buf + STAGE_A_BYTES + wave * 4096);
    }
    template <int I>
    __device__ __forceinline__ void issue(uint32_t lane16) const {
        const char* g = I < 2 ? ga + I * 1024 : gb + (I - 2) * 1024;
        const uint32_t l = I < 2 ? la + I * 1024 : lb + (I - 2) * 1024;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(lane16), "s"(g) : "memory", "m0");
    }
};
#pragma clang diagnostic pop

// A stage of the scan: K-stage `ks` of corpus tile `t`; ok == false marks "past the end" (the
// pipeline still issues a copy for it, into a buffer nobody reads, so the loop has no branches).
struct StageCursor { int t, ks; bool ok; };

// IPX_F32_ABL (ablation builds for measurement only — results are wrong; scripts/f32_ablation.sh): bit 0 = no stage
// barrier, bit 1 = no epilogue filter, bit 2 = no copy pieces in the loop, bit 3 = no episodes, bit 4 = no vmcnt wait,
// bit 5 = every query parked (the filter runs and never passes), bit 6 = accumulators not zeroed per tile,
// bit 7 = hits are voted on but nothing is appended (so no episode is ever triggered by a queue).
#ifndef IPX_F32_ABL
#define IPX_F32_ABL 0
#endif
// IPX_F32_OPT (A/B builds; every combination returns the same bits): bit 0 = the filter's hit path votes on eight
// 4-register groups before it looks at single registers, bit 1 = the copy pieces of the stage loop are issued from asm
// in the scalar-base form (no per-lane 64-bit address arithmetic: a VALU instruction in this loop is not hidden behind
// the other workgroup's MFMAs), bit 2 = accumulators zeroed two registers per instruction.
// (Measured and dropped in round 3: a tile's first MFMAs taking a zero C operand from a peeled first stage — 659.2 vs
// 656.5 ms; the two workgroups of a CU running their scheduled episodes at different tile counts — no difference.)
// bit 3 = the stage loop unrolled by two when a tile has an even number of K-stages (no address VALU in the loop).
#ifndef IPX_F32_OPT
#define IPX_F32_OPT 15
#endif
// DEEP: lists beyond IPX_MAX_K entries (ScanParams::deepK): the episodes' selections take 4 bytes of LDS per queue slot
// + a side list, on p.nselect waves; the final pass writes each split's best k keys unsorted (deep_finish_kernel sorts).
// The hot loop is the same code; the headline instantiation <false, false> is untouched by it.
template <bool CEIL, bool DEEP = false>
__global__ void __launch_bounds__(NTHREADS, 2)
scan_f32_kernel(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [0, 2*STAGE_BYTES) : two K-stage buffers; the same bytes are the sort scratch of a
    // compaction (4 waves x cap keys), which only runs when no stage is live.
    const int scratch_bytes = DEEP ? p.nselect * (p.cap + SEL_AUX) * 4 : 4 * p.cap * 8;
    const int stage_region = scratch_bytes > 2 * STAGE_BYTES ? scratch_bytes : 2 * STAGE_BYTES;
    int* cnts = (int*)(smem + stage_region);
    float* taus = (float*)(smem + stage_region + BQ * 4);
    int* flags = (int*)(smem + stage_region + 2 * BQ * 4);      // [0..1] "a queue is filling up" per tile parity
    int* runslot = flags + 2;                                    // first tile of the run just claimed

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wave & 1, wq = wave >> 1;
    const int h = lane >> 5, l31 = lane & 31;

    uint64_t t_start = 0;
    if (p.trace) t_start = __builtin_amdgcn_s_memrealtime();

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    int qb = logical / p.S, split = logical - qb * p.S;
    bool helper = false;
    const int fill_limit = p.cap - 2 * BD;      // a set flag is acted on one tile later
    int tiles_done = 0, n_compact = 0;
#ifdef IPX_F32_TRACE_HITS       // measurement build (scripts/trace_wgs.py): wave 0's hit groups, appends, ticks in episodes
    uint32_t tr_hits = 0, tr_app = 0;
    uint64_t tr_epi = 0, tr_ph[4] = {0, 0, 0, 0};
#endif
    const int nqb = (p.nq + BQ - 1) / BQ;
    // Thread 0 picks the query block with the most runs left (starting the search at a block of this
    // workgroup's own, so simultaneous pickers spread out) and takes one of its spare output slots.
    auto pick_block = [&]() -> bool {
        __syncthreads();
        if (tid == 0) {
            const int Rr = p.run_tiles, nruns = (p.ntiles + Rr - 1) / Rr;
            int pick = -1, most = 3;                // joining costs a list of its own: not for the last few runs
            for (int i = 0; i < nqb; ++i) {
                const int b = (int)((blockIdx.x + i) % nqb);
                const int left = nruns - p.S - (int)__hip_atomic_load(p.run_ctr + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (left > most && __hip_atomic_load(p.help_ctr + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)p.helpers) {
                    most = left;
                    pick = b;
                }
            }
            int slot_out = -1;
            if (pick >= 0) {
                const uint32_t hslot = atomicAdd(p.help_ctr + pick, 1u);
                if (hslot < (uint32_t)p.helpers) slot_out = p.S + (int)hslot;
            }
            runslot[0] = slot_out >= 0 ? pick : -1;
            flags[0] = slot_out;
        }
        __syncthreads();
        const int next_qb = runslot[0], next_slot = flags[0];
        __syncthreads();
        if (next_qb < 0) return false;
        qb = next_qb;
        split = next_slot;
        helper = true;
        return true;
    };
    // (Filling the eight CU slots the grid leaves empty with workgroups that help from the start was
    // measured too: no gain — the lone workgroups of those CUs already run that much faster.)

  // A workgroup scans for its own query block first; when that block's runs are used up it looks for
  // the block with the most runs left and joins it as a helper (a fresh, unpublished list in one of
  // the block's spare output slots, thresholds taken from the block's union bound): the query
  // blocks' groups of workgroups run at different speeds (a CU's first workgroup wins most MFMA
  // issue slots, eight CUs hold a single workgroup), and without this the fastest blocks' workgroups
  // idle for the last 7 % of the launch.
  for (;;) {
    // thresholds: -FLT_MAX is faiss's heap neutral (scores must be > it to enter);
    // +inf parks the padding queries of the last block.
    cnts[tid] = 0;
    taus[tid] = (!(IPX_F32_ABL & 32) && (int64_t)qb * BQ + tid < p.nq) ? -FLT_MAX : INFINITY;
    if (tid < 2) flags[tid] = 0;
    uint32_t* mypub = (p.S > 1 && !helper) ? p.pub + ((size_t)qb * p.S + split) * BQ : nullptr;
    if (helper) {
        __syncthreads();
        poll_union_threshold(p, taus, qb, tid);
    }

    uint64_t* myqueues = p.cand + (size_t)blockIdx.x * BQ * p.cap;
    uint64_t* scratch = (uint64_t*)smem + (size_t)wave * p.cap;

    const int sw = (lane >> 2) & 3;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t lane16 = (uint32_t)lane * 16;
    const int a_off = (wd * 64 + l31) * 64;     // byte offset of this lane's doc row in a stage
    // Query groups of 32 are dealt to the two wave columns alternately (group 2j + wq is wave column wq's j-th), so
    // the live groups of a ragged last block spread over both columns.
    const int b_off = STAGE_A_BYTES + (wq * 32 + l31) * 64;
    const int pc0 = ((0 + h) ^ sw) * 16, pc1 = ((2 + h) ^ sw) * 16;
    const int nreal = p.nq - qb * BQ < BQ ? p.nq - qb * BQ : BQ;            // real queries of this block
    int nj = (nreal - wq * 32 + 63) / 64;                                    // live groups of this wave column
    nj = nj < 0 ? 0 : (nj > 4 ? 4 : nj);

    // ---- which tiles this workgroup scans ----------------------------------------------------
    // The corpus is cut into runs of R tiles. The S workgroups of a query block take runs from a
    // shared counter (run `split` first, then whatever is next), always in increasing tile order,
    // so they all finish within one run of each other whatever speed each one gets on its CU.
    // The next run is claimed when the scan enters the current one, long before it is needed.
    const int R = p.run_tiles;
    int adv_end = (split + 1) * R < p.ntiles ? (split + 1) * R : p.ntiles;   // end of the run the prefetch is in
    int nxt = -1;                                   // first tile of the claimed next run, -1 = none
    int enter_t = -1;                               // entering this tile means: claim the run after it
    auto claim_next_run = [&]() {
        if (tid == 0) {
            const uint32_t run = (uint32_t)p.S + atomicAdd(p.run_ctr + qb, 1u);
            *runslot = (int64_t)run * R < p.ntiles ? (int)(run * R) : -1;
        }
        __syncthreads();
        nxt = *runslot;
        enter_t = -1;
    };
    auto advance = [&](StageCursor c) {
        if (!c.ok) return c;
        if (++c.ks < p.nks) return c;
        c.ks = 0;
        if (c.t + 1 < adv_end) { ++c.t; return c; }
        if (nxt >= 0) {
            c.t = nxt;
            enter_t = nxt;
            adv_end = nxt + R < p.ntiles ? nxt + R : p.ntiles;
            nxt = -1;
            return c;
        }
        c.ks = p.nks - 1;      // stay on the last real stage: a harmless re-read
        c.ok = false;
        return c;
    };

    // Software pipeline over the stages c0 (computing), c1, c2 (being copied HBM -> LDS):
    //   top of stage c0 : operand set 0 (first 8 features of the stage) is already in registers
    //   read set 1 (last 8 features) from LDS | 32 MFMAs on set 0        (covers the read)
    //   vmcnt(0) + barrier: stage c1 has landed, nobody reads stage c0's buffer any more
    //   issue stage c2 into stage c0's buffer | read set 0 of stage c1 | 32 MFMAs on set 1
    // so no MFMA ever waits on an LDS read, and a stage's HBM->LDS copy has one full stage
    // (4096 MFMA cycles) to land.
    StageCursor c0{split * R, 0, true};
    if (helper) {                                   // a helper has no pre-assigned run: its first one is claimed too
        claim_next_run();
        c0 = StageCursor{nxt >= 0 ? nxt : 0, 0, nxt >= 0};
        adv_end = c0.t + R < p.ntiles ? c0.t + R : p.ntiles;
    }
    claim_next_run();
    StageCursor c1 = advance(c0);
    StageCursor c2 = advance(c1);
    f32x4 a0[2], b0[4], a1[2], b1[4];
    issue_stage(p, smem, c0.t, qb, c0.ks, wave, lane);
    issue_stage(p, smem + STAGE_BYTES, c1.t, qb, c1.ks, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) a0[i] = *(const f32x4*)(smem + a_off + i * 2048 + pc0);
#pragma unroll
    for (int j = 0; j < 4; ++j) b0[j] = *(const f32x4*)(smem + b_off + j * 4096 + pc0);

    int g = 0;
    // The tile loop, instantiated per number of live 32-query groups of this wave (NJ): the last query block of a
    // batch is usually ragged (6980 queries = 27 blocks + 68), and its all-padding groups are not multiplied.
    // EVEN (an even number of K-stages per tile, full blocks only): the stage loop is unrolled by two, so which of the
    // two stage buffers a stage reads is known at compile time and goes into the LDS reads' immediate offsets — the
    // loop then has no VALU instruction at all (the four address additions per stage were the last ones).
    auto scan_tiles = [&](auto nj_const, auto even_tag) {
    constexpr int NJ = decltype(nj_const)::value;
    constexpr bool EVEN = decltype(even_tag)::value && (IPX_F32_OPT & 8) != 0;
    while (c0.ok) {
        const int t = c0.t;
        if (t == enter_t) claim_next_run();
        f32x16 acc[2][4];
        if constexpr ((IPX_F32_ABL & 64) != 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "=v"(acc[i][j]));
        } else if constexpr ((IPX_F32_OPT & 4) != 0) {
            // two registers per instruction (a VALU instruction at a tile boundary is not hidden behind the other
            // workgroup's MFMAs: the 128 v_mov_b32 were measured at 0.5 % of the launch)
            typedef uint64_t u64x8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u64x8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        uint64_t ze;
                        asm volatile("v_mov_b64 %0, 0" : "=v"(ze));
                        z[e] = ze;
                    }
                    acc[i][j] = __builtin_bit_cast(f32x16, z);
                }
        } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }

        // one K-stage; PAR = the buffer it reads (0 / 1), -1 = (g & 1), decided at run time
        auto stage = [&](auto par_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            const int par = PAR >= 0 ? PAR : (g & 1);
            const char* buf = smem + par * STAGE_BYTES;
            const char* nbuf = smem + (1 - par) * STAGE_BYTES;
            // Set 0 was read from LDS half a stage ago (32 MFMAs back): it has landed. Saying so here — a wait that
            // costs nothing, whose outputs the compiler takes as freshly defined — keeps hipcc from waiting for ALL
            // outstanding LDS reads, the six reads of set 1 just below included, in front of the first MFMA on set 0
            // (its counter state at the loop header is the join of two paths, and it falls back to lgkmcnt(0) there:
            // found in the ISA in round 3; 663.8 -> 657.1 ms on the headline workload, same box, interleaved).
            if constexpr (NJ == 4)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(b0[0]), "+v"(b0[1]), "+v"(b0[2]), "+v"(b0[3]));
#pragma unroll
            for (int i = 0; i < 2; ++i) a1[i] = *(const f32x4*)(buf + a_off + i * 2048 + pc1);
#pragma unroll
            for (int j = 0; j < NJ; ++j) b1[j] = *(const f32x4*)(buf + b_off + j * 4096 + pc1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][kk], b0[j][kk], acc[i][j], 0, 0, 0);
            // issue order: the six LDS reads of set 1 first, then the MFMAs that cover them
            if constexpr (NJ == 4) {
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
            }

            // keep the 32 MFMAs above the barrier (hipcc would sink them below it and expose
            // the LDS read latency of set 1)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(IPX_F32_ABL & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (!(IPX_F32_ABL & 1)) __syncthreads();
            if constexpr (NJ == 4 && (IPX_F32_OPT & 2) != 0 && !(IPX_F32_ABL & 4)) {
                // MFMAs start at once; the six copy pieces go out from asm, one behind every second MFMA, in source
                // order (the scheduler does not move what it cannot see into); then, as below, the six LDS reads of
                // the next stage's set 0 between pairs of MFMAs
                const StagePiecesF32 sp(p, smem_base + par * STAGE_BYTES, c2.t, qb, c2.ks, wave);
#define IPX_F32_MF(N_) acc[((N_) >> 2) & 1][(N_) & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32( \
                    a1[((N_) >> 2) & 1][(N_) >> 3], b1[(N_) & 3][(N_) >> 3], acc[((N_) >> 2) & 1][(N_) & 3], 0, 0, 0)
                IPX_F32_MF(0); IPX_F32_MF(1); sp.template issue<0>(lane16); __builtin_amdgcn_sched_barrier(0);
                IPX_F32_MF(2); IPX_F32_MF(3); sp.template issue<1>(lane16); __builtin_amdgcn_sched_barrier(0);
                IPX_F32_MF(4); IPX_F32_MF(5); sp.template issue<2>(lane16); __builtin_amdgcn_sched_barrier(0);
                IPX_F32_MF(6); IPX_F32_MF(7); sp.template issue<3>(lane16); __builtin_amdgcn_sched_barrier(0);
                IPX_F32_MF(8); IPX_F32_MF(9); sp.template issue<4>(lane16); __builtin_amdgcn_sched_barrier(0);
                IPX_F32_MF(10); IPX_F32_MF(11); sp.template issue<5>(lane16); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i) a0[i] = *(const f32x4*)(nbuf + a_off + i * 2048 + pc0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b0[j] = *(const f32x4*)(nbuf + b_off + j * 4096 + pc0);
                IPX_F32_MF(12); IPX_F32_MF(13); IPX_F32_MF(14); IPX_F32_MF(15); IPX_F32_MF(16); IPX_F32_MF(17);
                IPX_F32_MF(18); IPX_F32_MF(19); IPX_F32_MF(20); IPX_F32_MF(21); IPX_F32_MF(22); IPX_F32_MF(23);
                IPX_F32_MF(24); IPX_F32_MF(25); IPX_F32_MF(26); IPX_F32_MF(27); IPX_F32_MF(28); IPX_F32_MF(29);
                IPX_F32_MF(30); IPX_F32_MF(31);
#undef IPX_F32_MF
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            } else {
            // No branches from here to the end of the stage, so hipcc can spread the copy
            // issue and the LDS reads among the 32 MFMAs.
            if constexpr (!(IPX_F32_ABL & 4)) issue_stage(p, smem + par * STAGE_BYTES, c2.t, qb, c2.ks, wave, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i) a0[i] = *(const f32x4*)(nbuf + a_off + i * 2048 + pc0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) b0[j] = *(const f32x4*)(nbuf + b_off + j * 4096 + pc0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][kk], b1[j][kk], acc[i][j], 0, 0, 0);
            // issue order below the barrier: MFMAs start at once; the six copy pieces, then the
            // six LDS reads of the next stage's set 0, go out between pairs of MFMAs
            if constexpr (NJ == 4) {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            c0 = c1;
            c1 = c2;
            c2 = advance(c2);
            if constexpr (PAR < 0) ++g;
        };
        if constexpr (EVEN) {
            for (int ks = 0; ks < p.nks; ks += 2) {
                stage(std::integral_constant<int, 0>{});
                stage(std::integral_constant<int, 1>{});
            }
        } else {
            for (int ks = 0; ks < p.nks; ++ks) stage(std::integral_constant<int, -1>{});
        }
        // here c0 / c1 are the first two stages of the next tile of this workgroup (if any),
        // c0's set-0 operands are in registers, c1's copy is in flight

        // ---- epilogue: threshold filter, rare append to the per-query queues ----
        ++tiles_done;
        const int docbase = t * BD + wd * 64 + 4 * h;
        // flag raised by the PREVIOUS tile's appends: every wave of the workgroup reads the same
        // value (a tile's barriers lie between those writes and this read; this tile writes the other)
        bool episode = lds_read_i32_nodma(&flags[(tiles_done + 1) & 1]) != 0;
        int* myflag = &flags[tiles_done & 1];
        if constexpr ((IPX_F32_ABL & 2) != 0) {          // (keeps the accumulators alive without the filter)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(acc[i][j]));
        }
#pragma unroll
        for (int j = 0; j < ((IPX_F32_ABL & 2) ? 0 : NJ); ++j) {
            const int ql = (2 * j + wq) * 32 + l31;
            const float tau = lds_read_f32_nodma(&taus[ql]);
            // this lane's maximum over its 32 scores of the query, kept per group of four registers (four consecutive rows)
            float g8[8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    g8[i * 4 + q4] = fmaxf(fmaxf(acc[i][j][4 * q4], acc[i][j][4 * q4 + 1]), fmaxf(acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]));
            float m = g8[0];
#pragma unroll
            for (int u = 1; u < 8; ++u) m = fmaxf(m, g8[u]);
            if (__any(m > tau)) {
#ifdef IPX_F32_TRACE_HITS
                ++tr_hits;
#endif
                // Rare per element, but some lane of the wave passes on many tiles (measured, round 3: the hit path +
                // the episodes it feeds cost 2.8 % of the launch, the filter above 0.7 %): wave-wide votes, first per
                // group of four registers, then per register of a group somebody passes in — registers nobody passes
                // in cost a quarter of a compare + branch (round 1-2 voted on all 32 registers).
                uint64_t* qu = myqueues + (size_t)ql * p.cap;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        if constexpr ((IPX_F32_OPT & 1) != 0) {
                            if (!__any(g8[i * 4 + q4] > tau)) continue;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * q4 + e;
                            const float sc = acc[i][j][r];
                            if (__any(sc > tau)) {
                                const int64_t doc = docbase + i * 32 + (r & 3) + 8 * (r >> 2);
                                bool admit = sc > tau && doc < p.N;
                                if constexpr (CEIL) {
                                    if (admit) {        // (padding queries never get here: their tau is +inf)
                                        const int64_t cid = p.ceilI[(int64_t)qb * BQ + ql];
                                        admit = cid >= 0 && make_key(sc, (uint32_t)doc) <
                                                                make_key(p.ceilD[(int64_t)qb * BQ + ql], (uint32_t)(cid - p.id_base));
                                    }
                                }
                                if constexpr ((IPX_F32_ABL & 128) != 0) {
                                    if (admit) asm volatile("" :: "v"(doc));
                                    admit = false;
                                }
#ifdef IPX_F32_TRACE_HITS
                                tr_app += (uint32_t)__popcll(__ballot(admit));
#endif
                                if (admit) {
                                    const int pos = lds_atomic_inc_nodma(&cnts[ql]);
                                    if (pos < p.cap) qu[pos] = make_key(sc, (uint32_t)doc);
                                    else *p.err = 1;
                                    if (pos >= fill_limit) lds_write_i32_nodma(myflag, 1);
                                }
                            }
                        }
                    }
            }
        }
        // Episodes (workgroup-uniform decisions, no per-tile barrier):
        //  * compaction when a queue is filling up (flag of the previous tile: at most 2*BD more
        //    appends land before it runs, and fill_limit leaves room for them) and after
        //    4, 16, 64, ... tiles, so thresholds tighten and get published early in the scan;
        //  * a threshold poll alone after 8, 32, 128, ... tiles, when the other splits' values
        //    of the previous round are there.
        episode = episode || is_pow4(tiles_done);
        if constexpr ((IPX_F32_ABL & 8) != 0) episode = false;
        if (episode) {
            // The sort scratch aliases both stage buffers, which hold stage c0 (landed; its
            // set-0 operands are already in registers) and stage c1 (in flight): let the
            // copy land, sort, then fetch both stages again.
#ifdef IPX_F32_TRACE_HITS
            const uint64_t tr_e0 = __builtin_amdgcn_s_memrealtime();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if constexpr ((IPX_F32_OPT & 16) != 0) __builtin_amdgcn_s_setprio(3);
            ++n_compact;
            if (tid < 2) flags[tid] = 0;
            if constexpr (DEEP) {
                if (wave < p.nselect) {
                    uint32_t* sh = (uint32_t*)smem + (size_t)wave * (p.cap + SEL_AUX);
                    for (int ql = wave; ql < BQ; ql += p.nselect)
                        compact_query_select(p, (uint64_t*)sh, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane,
                                             mypub ? mypub + ql : nullptr, sh + p.cap, SEL_AUX);
                }
            } else {
                for (int ql = wave; ql < BQ; ql += 4)
#ifdef IPX_F32_TRACE_HITS
                    compact_query_select<true>(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane,
                                               mypub ? mypub + ql : nullptr, nullptr, 0, tr_ph);
#else
                    compact_query_select<true>(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane,
                                               mypub ? mypub + ql : nullptr);
#endif
            }
            if constexpr ((IPX_F32_OPT & 16) != 0) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
            poll_union_threshold(p, taus, qb, tid);
            issue_stage(p, smem + (g & 1) * STAGE_BYTES, c0.t, qb, c0.ks, wave, lane);
            issue_stage(p, smem + ((g + 1) & 1) * STAGE_BYTES, c1.t, qb, c1.ks, wave, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#ifdef IPX_F32_TRACE_HITS
            tr_epi += __builtin_amdgcn_s_memrealtime() - tr_e0;
#endif
        } else if ((tiles_done & 1) == 0 && is_pow4(tiles_done >> 1)) {
            poll_union_threshold(p, taus, qb, tid);
        }
    }
    };
    switch (nj) {
        case 4:
            if ((p.nks & 1) == 0) scan_tiles(std::integral_constant<int, 4>{}, std::true_type{});
            else scan_tiles(std::integral_constant<int, 4>{}, std::false_type{});
            break;
        case 3: scan_tiles(std::integral_constant<int, 3>{}, std::false_type{}); break;
        case 2: scan_tiles(std::integral_constant<int, 2>{}, std::false_type{}); break;
        case 1: scan_tiles(std::integral_constant<int, 1>{}, std::false_type{}); break;
        default: scan_tiles(std::integral_constant<int, 0>{}, std::false_type{}); break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the tail's dummy copies write LDS too
    __syncthreads();
    if constexpr (DEEP) {
        if (wave < p.nselect) {
            uint32_t* sh = (uint32_t*)smem + (size_t)wave * (p.cap + SEL_AUX);
            for (int ql = wave; ql < BQ; ql += p.nselect) {
                const int64_t qglobal = (int64_t)qb * BQ + ql;
                if (qglobal < p.nq)
                    compact_query_deep_final(p, sh, sh + p.cap, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane, qglobal, split);
            }
        }
    } else {
        for (int ql = wave; ql < BQ; ql += 4) {
            const int64_t qglobal = (int64_t)qb * BQ + ql;
            if (qglobal < p.nq)
                compact_query(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane, true, qglobal, split, nullptr);
        }
    }
    // ---- anything left to help with? ----
    if (p.helpers <= 0 || !pick_block()) break;
  }
    if (p.trace && tid == 0) {
        uint64_t* tr = p.trace + (size_t)blockIdx.x * 4;
        tr[0] = t_start;
        tr[1] = __builtin_amdgcn_s_memrealtime();
        tr[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        tr[3] = ((uint64_t)tiles_done << 32) | (uint64_t)n_compact;
#ifdef IPX_F32_TRACE_HITS
        tr[2] |= tr_epi << 32;                                             // ticks of 10 ns in episodes
        tr[0] = tr_ph[0] | (tr_ph[1] << 32);                               // ... of them, wave 0: load | publish
        tr[1] = tr_ph[2] | (tr_ph[3] << 32);                               //                     k-th | ties + move
        tr[3] = ((uint64_t)tiles_done << 48) | ((uint64_t)n_compact << 40) | ((uint64_t)(tr_hits & 0xfffff) << 20) | (tr_app & 0xfffff);
#endif
    }
}

// every result slot of an empty index: (-FLT_MAX, -1)
__global__ void __launch_bounds__(256) fill_empty_kernel(float* D, int64_t* I, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { D[i] = -FLT_MAX; I[i] = -1; }
}

// result order of the merges: score descending (as ordered bits), id ascending
__device__ __forceinline__ bool entry_before(uint32_t osa, int64_t ida, uint32_t osb, int64_t idb) {
    return osa > osb || (osa == osb && ida < idb);
}

// ---- deep lists: k beyond the single-pass limit, IPX_MAX_K ranks per pass -----------------------
// A pass's lists [nq][kk] go to columns [off, off + kk) of the [nq][k] result, and each query's last hit becomes
// the next pass's ceiling (id -1: the pass ran out of hits, the next ones return nothing).
__global__ void __launch_bounds__(256)
deep_collect_kernel(const float* __restrict__ tD, const int64_t* __restrict__ tI, int64_t nq, int kk, float* __restrict__ D,
                    int64_t* __restrict__ I, int k, int off, float* __restrict__ ceilD, int64_t* __restrict__ ceilI,
                    int* __restrict__ ceilN) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t q = gid / kk;
    if (q >= nq) return;
    const int i = (int)(gid - q * kk);
    const float sc = tD[gid];
    const int64_t id = tI[gid];
    D[q * k + off + i] = sc;
    I[q * k + off + i] = id;
    if (i == kk - 1) {
        if (ceilN) {
            // Merges only (a scan's row ids are distinct): how many copies of the ceiling entry — same score, same
            // id, e.g. one document present in two shards' lists — have been returned up to here. The next pass
            // admits the remaining copies (deep_stage_kernel); ResultHeap keeps such duplicates too (reducer.py:16).
            const uint32_t os = f32_ordered(sc + 0.0f);
            int c = 1;
            while (c < kk && tI[gid - c] == id && f32_ordered(tD[gid - c] + 0.0f) == os) ++c;
            const bool cont = c == kk && off > 0 && ceilI[q] == id && f32_ordered(ceilD[q] + 0.0f) == os;
            ceilN[q] = (cont ? ceilN[q] : 0) + c;
        }
        ceilD[q] = sc;
        ceilI[q] = id;
    }
}

// Stage of a deep merge: S lists of k_in entries per query, any order, cut into chunks of `chunk` entries ->
// [S * nchunks][nb][chunk] for queries [q0, q0 + nb), entries not strictly after the query's ceiling (and the padding
// of the last chunk) blanked. ceilI == nullptr: first pass, nothing blanked. The order is merge_topk_kernel's:
// score descending (-0.0 == +0.0), id ascending.
__global__ void __launch_bounds__(256)
deep_stage_kernel(const float* __restrict__ D, const int64_t* __restrict__ I, int S, int64_t nq, int k_in, int chunk,
                  int nchunks, int64_t q0, int64_t nb, const float* __restrict__ ceilD, const int64_t* __restrict__ ceilI,
                  const int* __restrict__ ceilN, int* __restrict__ eqctr, float* __restrict__ sD, int64_t* __restrict__ sI) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per_list = nb * chunk;
    const int64_t l = gid / per_list;                    // staged list: s * nchunks + c
    if (l >= (int64_t)S * nchunks) return;
    const int64_t rem = gid - l * per_list;
    const int64_t qb = rem / chunk;
    const int i = (int)(rem - qb * chunk);
    const int s = (int)(l / nchunks), c = (int)(l - (int64_t)s * nchunks);
    const int e = c * chunk + i;
    float sc = -FLT_MAX;
    int64_t id = -1;
    if (e < k_in) {
        const int64_t src = ((int64_t)s * nq + q0 + qb) * k_in + e;
        sc = D[src] + 0.0f;
        id = I[src];
        if (ceilI) {
            const int64_t cid = ceilI[q0 + qb];
            const float csc = ceilD[q0 + qb] + 0.0f;
            bool after = cid >= 0 && sc > -FLT_MAX &&
                         entry_before(f32_ordered(csc), cid, f32_ordered(sc), id);
            // a copy of the ceiling entry itself: ceilN of them were returned already (which ones does not matter,
            // they are identical), the others are still to come
            if (!after && cid >= 0 && id == cid && sc > -FLT_MAX && f32_ordered(sc) == f32_ordered(csc))
                after = atomicAdd(&eqctr[q0 + qb], 1) >= ceilN[q0 + qb];
            if (!after) { sc = -FLT_MAX; id = -1; }
        }
    }
    sD[gid] = sc;
    sI[gid] = id;
}

// ---- merge: per query, S lists of (score, id) -> global top-k ---------------------------------
// One workgroup per query keeps the running best P entries sorted in LDS and folds one list
// at a time into it (the streaming role of faiss ResultHeap.add_result, reducer.py:16):
//   load list -> [bitonic sort] -> M[i] = better(R[i], L[P-1-i]) -> bitonic merge.
// Order: score descending, id ascending; scores <= -FLT_MAX and NaN are empty slots.

__device__ __forceinline__ void wg_cmpx(uint32_t* os, int64_t* id, int lo, int hi, bool desc) {
    const uint32_t a = os[lo], b = os[hi];
    const int64_t ia = id[lo], ib = id[hi];
    // desc: the better entry goes to lo
    if (entry_before(b, ib, a, ia) == desc) { os[lo] = b; id[lo] = ib; os[hi] = a; id[hi] = ia; }
}

// The fold of query q's S lists into its top k_out, by one workgroup of 256 with 24 * P bytes of LDS at `smem`.
__device__ __forceinline__ void fold_lists(char* smem, const float* __restrict__ D, const int64_t* __restrict__ I, int S,
                                           int k_in, int k_out, int P, int lists_sorted, int64_t shard_stride,
                                           int64_t query_stride, float* __restrict__ outD, int64_t* __restrict__ outI,
                                           int64_t shard_stride_I, int64_t q) {
    int64_t* rid = (int64_t*)smem;          // [P]
    int64_t* lid = rid + P;                 // [P]
    uint32_t* ros = (uint32_t*)(lid + P);   // [P]
    uint32_t* los = ros + P;                // [P]
    const int tid = threadIdx.x;

    // os == 0 marks an empty slot (every real score orders above 0)
    for (int i = tid; i < P; i += 256) { ros[i] = 0; rid[i] = INT64_MAX; }
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const float* d = D + s * shard_stride + q * query_stride;
        const int64_t* ids = I + s * shard_stride_I + q * query_stride;
        for (int i = tid; i < P; i += 256) {
            uint32_t o = 0; int64_t id = INT64_MAX;
            if (i < k_in) {
                // ResultHeap only admits scores strictly above its neutral value
                // (-FLT_MAX after the reducer's negation); NaN never compares true.
                // (+ 0.0f: a -0.0 score ties with +0.0, as it does under faiss's float compares;
                // the scans never produce one, a caller's shard results may hold one)
                const float sc = d[i] + 0.0f;
                if (sc > -FLT_MAX) { o = f32_ordered(sc); id = ids[i]; }
            }
            los[i] = o; lid[i] = id;
        }
        __syncthreads();
        if (!lists_sorted) {
            for (int size = 2; size <= P; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int i = tid; i < (P >> 1); i += 256) {
                        const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                        wg_cmpx(los, lid, lo, lo + stride, (lo & size) == 0);
                    }
                    __syncthreads();
                }
        }
        // R and L sorted descending: max(R[i], L[P-1-i]) is bitonic and holds the best P
        for (int i = tid; i < P; i += 256) {
            const int j = P - 1 - i;
            if (entry_before(los[j], lid[j], ros[i], rid[i])) { ros[i] = los[j]; rid[i] = lid[j]; }
        }
        __syncthreads();
        for (int stride = P >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P >> 1); i += 256) {
                const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                wg_cmpx(ros, rid, lo, lo + stride, true);
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k_out; i += 256) {
        const bool empty = (i >= P) || ros[i] == 0;
        outD[q * k_out + i] = empty ? -FLT_MAX : f32_unordered(ros[i]);
        outI[q * k_out + i] = empty ? -1 : rid[i];
    }
}

__global__ void __launch_bounds__(256)
merge_topk_kernel(const float* __restrict__ D, const int64_t* __restrict__ I, int S, int64_t nq,
                  int k_in, int k_out, int P, int lists_sorted, int64_t shard_stride, int64_t query_stride,
                  float* __restrict__ outD, int64_t* __restrict__ outI, const int* __restrict__ only_flagged,
                  int64_t shard_stride_I) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (only_flagged && !only_flagged[blockIdx.x]) return;   // merge_pool_kernel already served this query
    // lists of one shard packed into one buffer (ipx_merge_topk_packed): ids and scores a different number of
    // elements apart from shard to shard; 0 = the same stride as the scores
    if (shard_stride_I == 0) shard_stride_I = shard_stride;
    fold_lists(smem, D, I, S, k_in, k_out, P, lists_sorted, shard_stride, query_stride, outD, outI, shard_stride_I, blockIdx.x);
}

// Merge of S SORTED lists per query, S large (the scan's per-split lists when few queries spread
// over every CU: S = 256 or 512 lists for a batch of <= 64 queries). Folding them one by one costs
// S dependent merge rounds; this kernel cuts first. Take the first m entries of every list
// (S*m <= MERGE_POOL values, m >= ceil(k_out / S) so there are at least k_out of them unless the
// lists are short): the k_out-th largest of these values, v, is the score of an actual entry with
// at least k_out entries at or above it — nothing below v can be in the result. (With m = 1 this
// is "the k-th best list head".) One thread per list then walks its head while entries are >= v
// into an LDS pool (lists are sorted: it stops at the first smaller entry), the pool is sorted
// once, order (score desc, id asc). A pool overflow (masses of ties) sets todo[q] = 1 and
// merge_topk_kernel folds that query the slow way.
constexpr int MERGE_POOL = 4096;

__global__ void __launch_bounds__(256)
merge_pool_kernel(const float* __restrict__ D, const int64_t* __restrict__ I, int S, int k_in, int k_out,
                  int64_t shard_stride, int64_t query_stride, float* __restrict__ outD, int64_t* __restrict__ outI,
                  int* __restrict__ todo, int fold_P) {
    extern __shared__ __attribute__((aligned(16))) char fold_smem[];      // 24 * fold_P bytes (0: overflow -> todo)
    __shared__ int64_t pid[MERGE_POOL];
    __shared__ uint32_t pos_[MERGE_POOL];
    __shared__ uint32_t s_v;
    __shared__ int s_n, s_need;
    __shared__ int hist[256];
    __shared__ int wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q = blockIdx.x;
    const int r = (k_out + S - 1) / S;
    int m = MERGE_POOL / S;
    m = m < r ? r : m;
    m = m > k_in ? k_in : m;
    if (tid == 0) { s_v = 0; s_n = 0; s_need = k_out; }
    if ((int64_t)S * m <= MERGE_POOL && r <= k_in) {
        // the sample: first m entries of each list, as ordered bits (0 = empty slot), in pos_
        const int nv = S * m;
        for (int e = tid; e < nv; e += 256) {
            const int s = e / m, i = e - s * m;
            const float sc = D[s * shard_stride + q * query_stride + i];
            pos_[e] = sc > -FLT_MAX ? f32_ordered(sc) : 0u;
        }
        // v = the k_out-th largest sample value (0 if fewer than k_out are real: nv >= k_out, empty slots are 0), by
        // radix selection, 8 bits per pass: a histogram of the digit among the values that share the prefix found so
        // far (one bin per thread), suffix sums over the bins (wave scans + the waves' totals), the bin where the
        // count from the top reaches what is still needed
        for (int shift = 24; shift >= 0; shift -= 8) {
            hist[tid] = 0;
            __syncthreads();                                   // (also: the sample / the previous pass's s_v, s_need)
            const uint32_t v = s_v, himask = shift == 24 ? 0u : 0xffffffffu << (shift + 8);
            for (int e = tid; e < nv; e += 256) {
                const uint32_t x = pos_[e];
                if ((x & himask) == v) atomicAdd(&hist[(x >> shift) & 255], 1);
            }
            __syncthreads();
            const int c = hist[tid];
            int suf = c;                                       // inclusive suffix sum inside the wave: bins >= mine
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(suf, off);
                if (lane + off < 64) suf += o;
            }
            if (lane == 0) wtot[wave] = suf;
            __syncthreads();
            for (int w = wave + 1; w < 4; ++w) suf += wtot[w];
            const int need = s_need;
            __syncthreads();                                   // every thread has read s_need before one rewrites it
            if (suf >= need && suf - c < need) {               // exactly one bin
                s_v = v | ((uint32_t)tid << shift);
                s_need = need - (suf - c);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const uint32_t v = s_v;
    for (int s = tid; s < S; s += 256) {
        const float* d = D + s * shard_stride + q * query_stride;
        const int64_t* ids = I + s * shard_stride + q * query_stride;
        for (int i = 0; i < k_in; ++i) {
            const float sc = d[i];
            if (!(sc > -FLT_MAX)) break;              // empty slots (and NaN) end a list
            const uint32_t o = f32_ordered(sc);
            if (o < v) break;
            const int at = atomicAdd(&s_n, 1);
            if (at < MERGE_POOL) { pos_[at] = o; pid[at] = ids[i]; }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n > MERGE_POOL) {
        // more entries at the cut than the pool holds (masses of ties: duplicate rows, a zero query): the lists are
        // folded one by one instead — here, when the launch brought the LDS for it, else by merge_topk_kernel (todo)
        if (fold_P > 0) {
            __syncthreads();
            fold_lists(fold_smem, D, I, S, k_in, k_out, fold_P, 1, shard_stride, query_stride, outD, outI, shard_stride, q);
            if (tid == 0 && todo) todo[q] = 0;
        } else if (tid == 0) {
            todo[q] = 1;
        }
        return;
    }
    if (tid == 0 && todo) todo[q] = 0;
    if (n <= 512) {
        // a short pool is ordered by rank: an entry's place is the number of entries before it (distinct rows of one
        // index: no two entries are equal) — one pass, no barriers
        for (int i = tid; i < n; i += 256) {
            const uint32_t o = pos_[i];
            const int64_t id = pid[i];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += entry_before(pos_[j], pid[j], o, id) ? 1 : 0;
            if (rank < k_out) {
                outD[q * k_out + rank] = f32_unordered(o);
                outI[q * k_out + rank] = id;
            }
        }
        for (int i = n + tid; i < k_out; i += 256) {
            outD[q * k_out + i] = -FLT_MAX;
            outI[q * k_out + i] = -1;
        }
        return;
    }
    int P = 2;
    while (P < n) P <<= 1;
    for (int i = n + tid; i < P; i += 256) { pos_[i] = 0; pid[i] = INT64_MAX; }
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P >> 1); i += 256) {
                const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                wg_cmpx(pos_, pid, lo, lo + stride, (lo & size) == 0);
            }
            __syncthreads();
        }
    for (int i = tid; i < k_out; i += 256) {
        const bool empty = i >= n || pos_[i] == 0;
        outD[q * k_out + i] = empty ? -FLT_MAX : f32_unordered(pos_[i]);
        outI[q * k_out + i] = empty ? -1 : pid[i];
    }
}

// ---- deep lists: the splits' unsorted best-k lists of one query -> its sorted top k_out ----------------------
// One workgroup of 1024 per query. keys: [nq][S][k_list] (each split's list in any order, counts[q * S + s] entries,
// as compact_query_deep_final leaves them). T = all entries of the query. T <= DEEP_LDS_KEYS (the rule: thresholds
// and the union bound leave a split little beyond its share of the top k): everything is sorted in LDS at once. More:
// the k_out-th largest key is found first by radix selection over the lists where they lie (8 passes of 8 bits with
// an LDS histogram), then only the entries at or above it — k_out of them — are gathered and sorted. Keys of a scan
// are distinct; equal keys (none today) would be taken as often as needed.
constexpr int DEEP_LDS_KEYS = DEEP_MAX_K;

__global__ void __launch_bounds__(1024)
deep_finish_kernel(const uint64_t* __restrict__ keys, const int* __restrict__ counts, int S, int k_list, int k_out,
                   int P_max, int64_t id_base, float* __restrict__ outD, int64_t* __restrict__ outI) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* lk = (uint64_t*)smem;                       // [P_max]
    int* off = (int*)(lk + P_max);                        // [S + 1]
    int* hist = off + S + 1;                              // [256]
    __shared__ uint64_t s_prefix;
    __shared__ int s_need, s_c1, s_c2;
    const int tid = threadIdx.x;
    const int64_t q = blockIdx.x;
    const uint64_t* base = keys + (size_t)q * S * k_list;
    if (tid == 0) {
        int acc = 0;
        for (int s = 0; s < S; ++s) { off[s] = acc; acc += counts[q * S + s]; }
        off[S] = acc;
    }
    __syncthreads();
    const int T = off[S];
    auto entry = [&](int e) -> uint64_t {                  // e-th entry of the query, lists concatenated
        int lo = 0, hi = S - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (off[mid] <= e) lo = mid; else hi = mid - 1; }
        return base[(size_t)lo * k_list + (e - off[lo])];
    };
    int m;                                                 // entries in lk
    if (T <= P_max) {
        for (int e = tid; e < T; e += 1024) lk[e] = entry(e);
        m = T;
    } else {
        if (tid == 0) { s_prefix = 0; s_need = k_out; s_c1 = 0; s_c2 = 0; }
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int i = tid; i < 256; i += 1024) hist[i] = 0;
            __syncthreads();
            const uint64_t prefix = s_prefix;
            for (int e = tid; e < T; e += 1024) {
                const uint64_t key = entry(e);
                if (shift == 56 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(int)(key >> shift) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int b = 255, acc = 0, need = s_need;
                while (b > 0 && acc + hist[b] < need) { acc += hist[b]; --b; }
                s_need = need - acc;
                s_prefix = prefix | ((uint64_t)b << shift);
            }
            __syncthreads();
        }
        const uint64_t kth = s_prefix;
        const int ncopies = s_need;                        // entries equal to kth that belong to the result (>= 1)
        for (int e = tid; e < T; e += 1024) {
            const uint64_t key = entry(e);
            if (key > kth) lk[ncopies + atomicAdd(&s_c1, 1)] = key;
            else if (key == kth) { const int c = atomicAdd(&s_c2, 1); if (c < ncopies) lk[c] = key; }
        }
        m = k_out;
    }
    __syncthreads();
    int P = 2;
    while (P < m) P <<= 1;
    for (int i = m + tid; i < P; i += 1024) lk[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (P >> 1); i += 1024) {
                const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t x = lk[lo], y = lk[hi];
                if ((x < y) == desc) { lk[lo] = y; lk[hi] = x; }
            }
            __syncthreads();
        }
    for (int i = tid; i < k_out; i += 1024) {
        const uint64_t key = i < m ? lk[i] : 0ull;
        outD[q * k_out + i] = key ? f32_unordered((uint32_t)(key >> 32)) : -FLT_MAX;
        outI[q * k_out + i] = key ? id_base + (int64_t)(uint32_t)(~(uint32_t)key) : -1;
    }
}

}  // namespace ipx
