// ipx_bf16_kernels.h — bf16 scan with exact fp32 rescoring (BASELINE.json configs[3], [4]).
//
// The scan kernel finds, per query, the K' > k rows with the largest APPROXIMATE scores
// (bf16-rounded operands, v_mfma_f32_32x32x16_bf16, fp32 accumulation); rescore_kernel then
// recomputes those rows' scores exactly — the same ordered fp32 fmaf chain the fp32 scan kernel
// produces (ipx_kernels.h) — and keeps the best k. The result is
// PROVABLY the fp32 result whenever the candidate list is long enough, which rescore_kernel
// checks per query (see margin rule there); queries that fail the check are flagged and the host
// repeats them on the fp32 scan. So faiss.IndexFlatIP semantics (retriever.py:19) are unchanged.
//
// bf16 data layout in HBM (pack_rows_bf16_kernel): tiles of 256 rows; a tile is cut along the
// feature axis into K-stages of 32 bf16 (64 bytes per row); one stage of one tile is a contiguous
// [256 rows][4 chunks][8 bf16] block (16 KiB) with the 16-byte chunk c of row r at slot
// c ^ ((r >> 2) & 3) — byte for byte the geometry of the fp32 layout, so the same linear LDS-DMA
// copies and conflict-free ds_read_b128 operand reads apply. d is zero-padded to a multiple of 32.
#pragma once
#include "ipx_kernels.h"

namespace ipx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HBD = 256;          // corpus rows per score tile
constexpr int HBK = 32;           // bf16 per K-stage
constexpr int HTHREADS = 512;     // 8 waves: 4 (docs) x 2 (queries), each 64 docs x 128 queries
constexpr int HNBUF = 4;          // LDS stage ring
constexpr int HSTAGE_A = HBD * 64;             // 16 KiB
constexpr int HSTAGE_B = BQ * 64;              // 16 KiB
constexpr int HSTAGE = HSTAGE_A + HSTAGE_B;    // 32 KiB
constexpr int HLOADS = HSTAGE / (HTHREADS * 16);   // LDS-DMA pieces per thread per stage (4)

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// MURAL aspect weight a of row r (mul_grain.py:103-105): the value itself, or softmax of logits.
// Same arithmetic as the fusion in pack_rows_kernel, so both paths see identical query vectors.
__global__ void __launch_bounds__(256)
fuse_aspect_kernel(const float* __restrict__ src, const float* __restrict__ w, int A, int logits,
                   int64_t n, int d, float* __restrict__ dst) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / d;
    const int e = (int)(gid - r * d);
    if (r >= n) return;
    float mx = -INFINITY, den = 0.f;
    if (logits) {
        for (int a = 0; a < A; ++a) mx = fmaxf(mx, w[r * A + a]);
        for (int a = 0; a < A; ++a) den += expf(w[r * A + a] - mx);
    }
    float v = 0.f;
    for (int a = 0; a < A; ++a) {
        float wa = w[r * A + a];
        if (logits) wa = expf(wa - mx) / den;
        v = fmaf(wa, src[(r * A + a) * (int64_t)d + e], v);
    }
    dst[gid] = v;
}

// MURAL query / document pooler (MULGRAIN_Pooler.forward, 'att' type, mul_grain.py:97-132):
//   logits = W h[0] + b;  rep = sum_a softmax(logits)_a * h[begin_id + a]
// h: [n][rows_kept][d] — row 0 is the [CLS] hidden state, rows 1..A the aspect tokens (the host
// passes only these rows of the encoder output). One workgroup per vector; the weighted sum runs
// as the same fmaf chain over the aspects as fuse_aspect_kernel.
__global__ void __launch_bounds__(256)
mural_pool_kernel(const float* __restrict__ h, const float* __restrict__ W, const float* __restrict__ b,
                  int A, int64_t n, int d, float* __restrict__ dst, float* __restrict__ att_out) {
    __shared__ float s_logit[64];
    __shared__ float s_w[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r = blockIdx.x;
    const float* cls = h + r * (int64_t)(1 + A) * d;
    for (int a = wave; a < A; a += 4) {
        float acc = 0.f;
        for (int e = lane; e < d; e += 64) acc = fmaf(W[(int64_t)a * d + e], cls[e], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) s_logit[a] = acc + b[a];
    }
    __syncthreads();
    if (tid == 0) {
        float mx = -INFINITY, den = 0.f;
        for (int a = 0; a < A; ++a) mx = fmaxf(mx, s_logit[a]);
        for (int a = 0; a < A; ++a) den += expf(s_logit[a] - mx);
        for (int a = 0; a < A; ++a) {
            s_w[a] = expf(s_logit[a] - mx) / den;
            if (att_out) att_out[r * A + a] = s_w[a];
        }
    }
    __syncthreads();
    for (int e = tid; e < d; e += 256) {
        float v = 0.f;
        for (int a = 0; a < A; ++a) v = fmaf(s_w[a], cls[(int64_t)(1 + a) * d + e], v);
        dst[r * (int64_t)d + e] = v;
    }
}

// Swizzle of the query blocks that scan_bf16_dx_kernel reads with 16-row operand reads (ipx_bf16_dx_kernels.h):
// chunk c of row r sits at slot c ^ sigma((r >> 2) & 3), sigma = 0, 2, 3, 1.
__device__ __forceinline__ int dx_sigma(int q) { return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1); }

// row-major [n, d] (f32 / f16 / bf16) -> tiled, padded, swizzled bf16 (round to nearest even).
// One thread per (row, 16-byte chunk of 8 bf16). parts == 2 (queries only) also writes the
// residual lo = bf16(x - hi) as a second block behind each stage: x ~ hi + lo to 16 bits.
// swz == 1 (queries only): the dx_sigma swizzle instead of the corpus one.
__global__ void __launch_bounds__(256)
pack_rows_bf16_kernel(const void* __restrict__ src, int dtype, int64_t n, int64_t n_pad, int d,
                      uint16_t* __restrict__ dst, int64_t row0, int rows_per_tile, int nks, int parts, int swz,
                      uint32_t* __restrict__ zero, int64_t zero_words) {
    const int nchunks = nks * 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // (queries only) the scan's per-search counters and published bounds start at zero: cleared here, by the kernel
    // that runs right before the scan anyway, instead of by a fill of its own
    for (int64_t i = gid; i < zero_words; i += (int64_t)gridDim.x * 256) zero[i] = 0u;
    const int64_t r = gid / nchunks;
    const int c = (int)(gid - r * nchunks);
    if (r >= n_pad) return;
    uint16_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < n) {
        const int k0 = c * 8;
        const int64_t base = r * (int64_t)d;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k0 + e < d) {
                if (dtype == 2) {
                    v[e] = ((const uint16_t*)src)[base + k0 + e];
                } else {
                    const float x = load_as_f32(src, dtype, base + k0 + e);
                    v[e] = f32_to_bf16_rne(x);
                    if (parts == 2) w[e] = f32_to_bf16_rne(x - __uint_as_float((uint32_t)v[e] << 16));
                }
            }
    }
    const int64_t R = row0 + r;
    const int64_t tile = R / rows_per_tile;
    const int rr = (int)(R - tile * rows_per_tile);
    const int ks = c >> 2, pc = (c & 3) ^ (swz ? dx_sigma((rr >> 2) & 3) : ((rr >> 2) & 3));
    uint16_t* o = dst + (((tile * nks + ks) * parts) * rows_per_tile + rr) * 32 + pc * 8;
    uint4 out;
    out.x = v[0] | ((uint32_t)v[1] << 16);
    out.y = v[2] | ((uint32_t)v[3] << 16);
    out.z = v[4] | ((uint32_t)v[5] << 16);
    out.w = v[6] | ((uint32_t)v[7] << 16);
    *(uint4*)o = out;
    if (parts == 2) {
        out.x = w[0] | ((uint32_t)w[1] << 16);
        out.y = w[2] | ((uint32_t)w[3] << 16);
        out.z = w[4] | ((uint32_t)w[5] << 16);
        out.w = w[6] | ((uint32_t)w[7] << 16);
        *(uint4*)(o + (size_t)rows_per_tile * 32) = out;
    }
}

// The two corpus-side numbers of the margin rule (rescore_kernel), as float bits (non-negative floats
// order like unsigned ints; NaN / inf end up on top and make every margin check fail, which is
// intended):
//   maxsq[0] = max over rows of |x|^2 of the values the exact score is defined on: the fp32 rows
//              (mode 1) or the stored, rounded rows bf16(x) (mode 2);
//   maxsq[1] = max over rows of |x - bf16(x)|^2, the ACTUAL rounding error of the scanned copy
//              (mode 1; 0 in mode 2, where the scanned rows are the exact ones).
// One wave per row, grid-stride.
//
// Outlier rows (index with fp32 rows + bf16 copy only). Both numbers are maxima over the whole index, so a handful of
// rows far longer than the rest — or one NaN / inf row — would widen every query's margin. Rows whose |x|^2 is not
// <= *thr2 (4 x the mean |x|^2 of the first rows added: twice their rms norm; NaN and inf included) are therefore
// taken out of the bf16 scan altogether: their ids go to a short list (at most OUTLIER_MAX; rows beyond that count as
// ordinary rows), they do not enter the maxima, nan_rows_bf16_kernel turns their bf16 copies into NaN (a NaN score is
// never above a threshold, so they are never candidates), and every search scores them exactly on the side and merges
// them in (scan_bf16_path). The margin rule then speaks about the remaining rows only, which is all its proof needs.
constexpr int OUTLIER_MAX = 256;
struct OutlierState {          // device memory
    float thr2;                // 0 = not set yet
    int count;                 // rows classified (may exceed OUTLIER_MAX: only the first OUTLIER_MAX are listed)
    int64_t pad;
    int64_t rows[OUTLIER_MAX]; // row numbers in the index, in no particular order
};

// thr2 = 4 x mean |x|^2 over the first min(n, 2048) rows of the first add (one workgroup of 1024; set once)
__global__ void __launch_bounds__(1024)
outlier_threshold_kernel(const void* __restrict__ src, int dtype, int64_t n, int d, OutlierState* st) {
    __shared__ double part[1024];
    const int64_t m = n < 2048 ? n : 2048;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < m * d; i += 1024) {
        const float x = load_as_f32(src, dtype, i);
        if (x <= FLT_MAX && x >= -FLT_MAX) acc += (double)x * (double)x;      // NaN / inf do not set the scale
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) part[threadIdx.x] += part[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = part[0] / (double)m;
        st->thr2 = mean > 0.0 && mean < 1e37 ? (float)(4.0 * mean) : FLT_MAX;
    }
}

// the list the side rescoring reads: id_base + row, -1 behind the last one
__global__ void __launch_bounds__(256)
outlier_list_kernel(const OutlierState* __restrict__ st, int64_t id_base, int64_t* __restrict__ ids) {
    const int cnt = st->count < OUTLIER_MAX ? st->count : OUTLIER_MAX;
    for (int i = threadIdx.x; i < OUTLIER_MAX; i += 256) ids[i] = i < cnt ? id_base + st->rows[i] : -1;
}

__global__ void __launch_bounds__(256)
row_sumsq_max_kernel(const void* __restrict__ src, int dtype, int64_t n, int d, uint32_t* __restrict__ maxsq, int mode,
                     OutlierState* outl, int64_t row0) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float thr2 = outl ? outl->thr2 : 0.f;
    float best = 0.f, dbest = 0.f;
    bool bad = false;
    for (int64_t r = wave; r < n; r += nwaves) {
        float s = 0.f, ds = 0.f;
        for (int e = lane; e < d; e += 64) {
            const float x = load_as_f32(src, dtype, r * (int64_t)d + e);
            const float xr = __uint_as_float((uint32_t)f32_to_bf16_rne(x) << 16);
            const float v = mode == 2 ? xr : x;
            const float dx = x - xr;
            s = fmaf(v, v, s);
            ds = fmaf(dx, dx, ds);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); ds += __shfl_xor(ds, off); }
        if (thr2 > 0.f && !(s <= thr2)) {            // wave-uniform: an outlier row (or NaN / inf)
            int pos = 0;
            if (lane == 0) pos = atomicAdd(&outl->count, 1);
            pos = __shfl(pos, 0);
            if (pos < OUTLIER_MAX) {
                if (lane == 0) outl->rows[pos] = row0 + r;
                continue;                            // listed: it does not enter the maxima
            }
        }
        if (!(s <= FLT_MAX) || !(ds <= FLT_MAX)) bad = true;      // NaN or inf
        best = fmaxf(best, s);
        dbest = fmaxf(dbest, ds);
    }
    if (lane == 0) {
        atomicMax(maxsq, bad ? 0x7fc00000u : __float_as_uint(best));
        if (mode != 2) atomicMax(maxsq + 1, bad ? 0x7fc00000u : __float_as_uint(dbest));
    }
}

// bf16 copies of the listed outlier rows -> NaN (every listed row, every time: idempotent). One workgroup per row.
__global__ void __launch_bounds__(64)
nan_rows_bf16_kernel(uint16_t* __restrict__ Xh, int nks, const OutlierState* __restrict__ outl) {
    const int cnt = outl->count < OUTLIER_MAX ? outl->count : OUTLIER_MAX;
    if ((int)blockIdx.x >= cnt) return;
    const int64_t row = outl->rows[blockIdx.x];
    const int64_t tile = row / HBD;
    const int rr = (int)(row - tile * HBD);
    for (int i = threadIdx.x; i < nks * 32; i += 64) {           // 32 bf16 per K-stage line
        const int ks = i >> 5, e = i & 31;
        Xh[((tile * nks + ks) * HBD + rr) * 32 + e] = 0x7fc0;
    }
}

// ---- bf16 scan ---------------------------------------------------------------------------------
#define IPX_DS_READ_B128(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
#define IPX_GLDS(g_, l_) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g_), \
                                                           (__attribute__((address_space(3))) void*)(l_), 16, 0, 0)

__device__ __forceinline__ bf16x8 as_bf16x8(f32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// Geometry of one scan variant. QW = 32-query groups per wave: 4 -> 256 queries per workgroup
// (large batches, MFMA-bound), 1 -> 64 queries per workgroup (small batches: a quarter of the
// MFMA work per corpus byte and a deeper ring, so the corpus stream, not the matrix core, sets
// the pace — the HBM-bound regime of BASELINE.json configs[4]).
// QP = bf16 parts per query: 1 -> q ~ bf16(q); 2 -> q ~ hi + lo with hi = bf16(q), lo = bf16(q - hi)
// (16 significant bits: the query's rounding error all but vanishes from the margin, at twice
// the MFMA work; both parts accumulate into the same fp32 accumulator).
template <int QW, int QP> struct HCfg {
    static constexpr int BQ_ = 64 * QW;                  // queries per workgroup (2 wave columns)
    static constexpr int PART_B = BQ_ * 64;              // one part of the query stage
    static constexpr int STAGE_B = PART_B * QP;
    static constexpr int STAGE = HSTAGE_A + STAGE_B;     // 32 / 48 KiB (QW 4), 20 / 24 KiB (QW 1)
    static constexpr int NBUF = QW == 4 ? (QP == 1 ? 4 : 3) : 6;      // LDS stage ring (<= 144 KiB)
    static constexpr int LOADS = 2 + (QW == 4 ? 2 * QP : 1);          // LDS-DMA pieces per thread per stage
    static constexpr int RING = NBUF * STAGE;
};

// The copy pieces of one stage: A = 16 KiB contiguous = 2 rounds of 8 waves x 1 KiB; B = QP parts
// of 16 KiB (QW 4: 2 rounds per part) or of 4 KiB (QW 1: with one part waves 4..7 repeat the
// pieces of waves 0..3 — same bytes to the same place — and with two parts every wave has its own
// piece; either way every wave issues the same number of copies, so one vmcnt count serves all).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"     // m0 on a clobber list (the copy pieces set it themselves)
// (Round 3: issued from asm in the scalar-base form — M0 = the piece's LDS address, base in scalar registers, the
// lane's 16 bytes as a 32-bit offset: no per-lane 64-bit address arithmetic in the stage loop.)
template <int QW, int QP>
struct StagePieces {
    const char* ga;     // the wave's first A piece (wave-uniform)
    const char* gb;     // ... first B piece
    uint32_t la;        // the wave's first A piece in LDS
    uint32_t lb;
    uint32_t lane16;
    static __device__ __forceinline__ const char* uniform_ptr(const char* ptr) {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    __device__ __forceinline__ StagePieces(const ScanParams& p, char* buf, int64_t tile, int qb, int ks, int wave, int lane) {
        using C = HCfg<QW, QP>;
        const int wb = (QW == 4 || QP == 2) ? wave : (wave & 3);
        ga = uniform_ptr((const char*)p.Xp + ((size_t)tile * p.nks + ks) * HSTAGE_A + wave * 1024);
        gb = uniform_ptr((const char*)p.Qp + ((size_t)qb * p.nks + ks) * C::STAGE_B + wb * 1024);
        const uint32_t lbuf = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)buf;
        la = __builtin_amdgcn_readfirstlane(lbuf + wave * 1024);
        lb = __builtin_amdgcn_readfirstlane(lbuf + HSTAGE_A + wb * 1024);
        lane16 = (uint32_t)lane * 16;
    }
    // from the wave's piece addresses of the stage (kept as running pointers by the scan loop)
    __device__ __forceinline__ StagePieces(const char* ga_, const char* gb_, char* buf, int wave, int lane) {
        const int wb = (QW == 4 || QP == 2) ? wave : (wave & 3);
        ga = ga_;
        gb = gb_;
        const uint32_t lbuf = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)buf;
        la = __builtin_amdgcn_readfirstlane(lbuf + wave * 1024);
        lb = __builtin_amdgcn_readfirstlane(lbuf + HSTAGE_A + wb * 1024);
        lane16 = (uint32_t)lane * 16;
    }
    // piece i of HCfg::LOADS: 0, 1 = the A rounds; 2.. = the B rounds (8 KiB apart)
    __device__ __forceinline__ void issue(int i) const {
        const char* g = i < 2 ? ga + i * 8192 : gb + (i - 2) * 8192;
        const uint32_t l = i < 2 ? la + i * 8192 : lb + (i - 2) * 8192;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(l), "v"(lane16), "s"(g) : "memory", "m0");
    }
};
#pragma clang diagnostic pop

template <int QW, int QP>
__device__ __forceinline__ void issue_stage_bf16(const ScanParams& p, char* buf, int64_t tile, int qb,
                                                 int ks, int wave, int lane) {
    const StagePieces<QW, QP> sp(p, buf, tile, qb, ks, wave, lane);
#pragma unroll
    for (int i = 0; i < HCfg<QW, QP>::LOADS; ++i) sp.issue(i);
}

// Short-list path of the small-batch scans: the K-th largest of the nvals values published for one query (ordered
// score bits, 0 = not published yet; stride elements apart), rounded down to its top 18 bits — 0 if fewer than K are
// there. The values are the best scores of disjoint sets of rows, so at least K rows of the index score at or above
// the result: a floor for a list of K candidates. (With K = nvals it is the minimum; with a list shorter than the
// number of sets it is far tighter: 150 of 256 sets admit ~ 230 / N of the rows, the minimum ~ 1400 / N.) TPQ adjacent
// lanes per query hold NV values each in registers and decide bit by bit; part = lane's index among the TPQ.
template <int NV, int TPQ>
__device__ __forceinline__ uint32_t kth_largest_published(const uint32_t* col, size_t stride, int nvals, int K, int part) {
    uint32_t vals[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int s2 = part + TPQ * i;
        vals[i] = s2 < nvals ? __hip_atomic_load(col + (size_t)s2 * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    uint32_t pre = 0;
#pragma unroll 1
    for (int b = 31; b >= 14; --b) {
        const uint32_t cand = pre | (1u << b);
        int c = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) c += vals[i] >= cand ? 1 : 0;
        if constexpr (TPQ == 8) {
            // sum over the 8 lanes by DPP moves (a few cycles each; a shuffle through LDS costs ~100): the other lane
            // of the pair, the other pair of the quad, the mirror lane of the half-row (= the other quad's sum)
            c += __builtin_amdgcn_update_dpp(0, c, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
            c += __builtin_amdgcn_update_dpp(0, c, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
            c += __builtin_amdgcn_update_dpp(0, c, 0x141, 0xF, 0xF, true);      // row_half_mirror
        } else {
#pragma unroll
            for (int off = 1; off < TPQ; off <<= 1) c += __shfl_xor(c, off);
        }
        if (c >= K) pre = cand;
    }
    return pre;
}

template <int BQ_>
__device__ __forceinline__ void poll_union_threshold_n(const ScanParams& p, float* taus, int qb, int tid) {
    if (p.S <= 1) return;
    const uint32_t* col = p.pub + (size_t)qb * p.S * BQ_ + tid;
    uint32_t v = 0xffffffffu;
    for (int s2 = 0; s2 < p.S; ++s2) {
        const uint32_t x = __hip_atomic_load(col + (size_t)s2 * BQ_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = x < v ? x : v;
    }
    if (v != 0) taus[tid] = fmaxf(taus[tid], f32_below(f32_unordered(v)));
}

template <int QW>
__device__ __forceinline__ void read_a(f32x4 (&a)[2], uint32_t aa) {
    IPX_DS_READ_B128(a[0], aa, 0); IPX_DS_READ_B128(a[1], aa, 2048);
}
template <int QW>
__device__ __forceinline__ void read_b(f32x4 (&b)[QW], uint32_t bb) {
    IPX_DS_READ_B128(b[0], bb, 0);
    if constexpr (QW == 4) {
        IPX_DS_READ_B128(b[1], bb, 2048); IPX_DS_READ_B128(b[2], bb, 4096); IPX_DS_READ_B128(b[3], bb, 6144);
    }
}
// s_waitcnt lgkmcnt(N) that the compiler sees as (re)defining the operand registers waited for, so
// no MFMA that consumes them can be moved above it
template <int QW, int N>
__device__ __forceinline__ void wait_ab(f32x4 (&a)[2], f32x4 (&b)[QW]) {
    if constexpr (QW == 4)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]) : "n"(N) : "memory");
}
template <int QW, int N>
__device__ __forceinline__ void wait_b(f32x4 (&b)[QW]) {
    if constexpr (QW == 4)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b[0]) : "n"(N) : "memory");
}
// An empty asm that (to the compiler) reads and rewrites every accumulator: the MFMAs issued
// before it cannot be sunk below it. Without this pin hipcc moves most of a phase's MFMAs behind
// the later phases' operand reads, renames the operand registers to keep the old values alive,
// spills, and the spill reloads' vmcnt(0) drains the copy pipeline every stage.
template <int QW>
__device__ __forceinline__ void pin_acc(f32x16 (&acc)[2][QW]) {
    if constexpr (QW == 4)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]),
                          "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
    else
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]));
}
template <int QW>
__device__ __forceinline__ void mfma_set(f32x16 (&acc)[2][QW], const f32x4 (&a)[2], const f32x4 (&b)[QW]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < QW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a[i]), as_bf16x8(b[j]), acc[i][j], 0, 0, 0);
    pin_acc<QW>(acc);
}

// Same streaming top-k machinery as scan_f32_kernel (queues, thresholds, compaction episodes,
// union threshold); the tile shape, the stage ring and the MFMA differ. The pipeline keeps
// NBUF - 1 stages in flight with counted vmcnt waits; the operand reads are issued from asm
// (a compiler-visible LDS read after an LDS-DMA copy would get a vmcnt(0)).
template <int QW, int QP>
__global__ void __launch_bounds__(HTHREADS)
scan_bf16_kernel(const ScanParams p, const int nsortwaves) {
    using C = HCfg<QW, QP>;
    constexpr int BQ_ = C::BQ_, NBUF = C::NBUF, STAGE = C::STAGE, LOADS = C::LOADS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int scratch_bytes = nsortwaves * p.cap * 8;
    const int stage_region = scratch_bytes > C::RING ? scratch_bytes : C::RING;
    int* cnts = (int*)(smem + stage_region);
    float* taus = (float*)(smem + stage_region + BQ_ * 4);
    int* flags = (int*)(smem + stage_region + 2 * BQ_ * 4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wave & 3, wq = wave >> 2;
    const int h = lane >> 5, l31 = lane & 31;

    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int nqb_ = (int)gridDim.x / p.S;
    int qb, split;
    decode_block(p.map, p.S, nqb_, logical, qb, split);

    if (tid < BQ_) {
        cnts[tid] = 0;
        taus[tid] = ((int64_t)qb * BQ_ + tid < p.nq) ? -FLT_MAX : INFINITY;
#ifdef IPX_SMALL_ABL       // measurement build: only query 0 is live (the others are multiplied and never pass the filter)
        if (tid >= 1) taus[tid] = INFINITY;
#endif
    }
    if (tid < 2) flags[tid] = 0;
    const int fill_limit = p.cap - 2 * HBD;
    // Small batches with short lists (candidates <= 4 S): no compaction episodes unless a queue fills.
    // Each split deals its tiles round-robin to r = ceil(candidates / S) sub-splits and keeps the best
    // score of each by an LDS max beside the appends; the S r bests are published / polled at r, 4r,
    // 16r, 64r ... tiles (round 3: half the polls of the power-of-two schedule, ~1 % less scan time at 1-16 queries, the
    // same at 64 — scripts/small_ab.py), and their minimum bounds the list's last score from below (S r
    // disjoint sets of rows, each holding a row at or above it). With r == 1: every split publishes
    // its best. (Each split sees only N / S rows, so its own k-th best is a poor threshold anyway;
    // the union bound is the one that prunes.)
    constexpr int MAXSUB = 4;
    uint32_t* best = (uint32_t*)(flags + 4);              // [MAXSUB][BQ_]   (QW == 1 only)
    uint32_t* tmin = best + MAXSUB * BQ_;                 // [BQ_]
    const bool fast1 = QW == 1 && p.S > 1 && p.r <= MAXSUB;
    const int rs = fast1 ? p.r : 1;                       // published values per split and query
    uint32_t* mypub = p.S > 1 ? p.pub + ((size_t)qb * p.S + split) * rs * BQ_ : nullptr;
    if (QW == 1 && tid < BQ_) {
#pragma unroll
        for (int u = 0; u < MAXSUB; ++u) best[u * BQ_ + tid] = 0;
    }
    uint64_t* myqueues = p.cand + (size_t)blockIdx.x * BQ_ * p.cap;
    uint64_t* scratch = (uint64_t*)smem + (size_t)wave * p.cap;

    const int sw = (lane >> 2) & 3;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t a_off = (wd * 64 + l31) * 64;
    const uint32_t b_off = HSTAGE_A + (wq * (32 * QW) + l31) * 64;
    const uint32_t pc0 = ((0 + h) ^ sw) * 16, pc1 = ((2 + h) ^ sw) * 16;

    // Runs of R tiles (R a power of two) are dealt round-robin: split s takes runs s, s + S,
    // s + 2S, ... (one workgroup per CU here, so there are no fast / slow pairs to balance
    // dynamically as in the fp32 kernel). A cursor finds its successor by itself — the stage ring
    // may look further ahead than one run is long (one K-stage per tile when d <= 32).
    const int R = p.run_tiles, logR = __builtin_ctz((unsigned)R);
    auto advance = [&](StageCursor c) {
        if (!c.ok) return c;
        if (++c.ks < p.nks) return c;
        c.ks = 0;
        const int t1 = c.t + 1;
        if ((t1 & (R - 1)) != 0 && t1 < p.ntiles) { c.t = t1; return c; }      // still inside the run
        const int64_t next = ((int64_t)(c.t >> logR) + p.S) << logR;
        if (next < p.ntiles) { c.t = (int)next; return c; }
        c.ks = p.nks - 1;      // past the end: keep re-reading the last real stage into a buffer nobody reads
        c.ok = false;
        return c;
    };

    // c0 = the tile being computed (its first stage in ring slot `slot` at the top of a tile); ci = the next stage to
    // copy, NBUF stages ahead, with the wave's piece addresses of that stage kept as running pointers: inside a tile a
    // stage is one constant step further in both operands, so the loop adds instead of multiplying and the cursor's
    // branches run once per tile (round 3; rounds 1-2 shifted an array of NBUF + 1 cursors every stage).
    StageCursor c0{split * R, 0, split * R < p.ntiles};
    StageCursor ci = c0;
#pragma unroll 1
    for (int i = 0; i < NBUF; ++i) {
        issue_stage_bf16<QW, QP>(p, smem + i * STAGE, ci.t, qb, ci.ks, wave, lane);
        ci = advance(ci);
    }
    const char* ga_i;
    const char* gb_i;
    auto ci_ptrs = [&]() {
        const StagePieces<QW, QP> sp0(p, smem, ci.t, qb, ci.ks, wave, lane);
        ga_i = sp0.ga;
        gb_i = sp0.gb;
    };
    auto step_ci = [&]() {
        if (ci.ok && ci.ks + 1 < p.nks) {
            ++ci.ks;
            ga_i += HSTAGE_A;
            gb_i += C::STAGE_B;
        } else {
            ci = advance(ci);
            ci_ptrs();
        }
    };
    ci_ptrs();
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * (NBUF - 1)) : "memory");
    __syncthreads();

    // operand registers: A[s] = docs of K-set s (first / last 16 features of the stage); B[x] = a
    // ping-pong pair of query operand sets. At the top of a stage A[0], B[0] hold set 0 (part hi).
    f32x4 A[2][2], B[2][QW];
    // (small batches with hi + lo parts: B[s] = the hi part of K-set s, BL[s] = its lo part; see TWO_PHASE below)
    constexpr bool TWO_PHASE = QW == 1 && QP == 2;
    static_assert(!TWO_PHASE || C::PART_B == 4096, "the lo part's immediate offset below");
    f32x4 BL[2];
    read_a<QW>(A[0], smem_base + a_off + pc0);
    read_b<QW>(B[0], smem_base + b_off + pc0);
    if constexpr (TWO_PHASE) IPX_DS_READ_B128(BL[0], smem_base + b_off + pc0, 4096);

    int slot = 0, tiles_done = 0;
    // optional trace (ipx_set_param "trace"): wave 0's view of where the time goes
    uint64_t tr_epi = 0, tr_sort = 0, tr_begin = 0, tr_w1 = 0, tr_own = 0, tr_w2 = 0;
    uint32_t tr_appends = 0, tr_episodes = 0;
    if (p.trace) tr_begin = __builtin_readcyclecounter();
    while (c0.ok) {
        const int t = c0.t;
        f32x16 acc[2][QW];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < QW; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int ks = 0; ks < p.nks; ++ks) {
            const int nslot = slot + 1 == NBUF ? 0 : slot + 1;
            const uint32_t buf = smem_base + slot * STAGE;
            const uint32_t nbuf = smem_base + nslot * STAGE;
            constexpr int NB = QW;                 // LDS reads per B operand set
            if constexpr (QP == 1) {
                // first half: set 1 is read while set 0 (read one half-stage ago) is multiplied
                read_a<QW>(A[1], buf + a_off + pc1);
                read_b<QW>(B[1], buf + b_off + pc1);
                wait_ab<QW, 2 + NB>(A[0], B[0]);
                mfma_set<QW>(acc, A[0], B[0]);
                __builtin_amdgcn_sched_barrier(0);
                wait_ab<QW, 0>(A[1], B[1]);
            } else if constexpr (TWO_PHASE) {
                // Small batches, hi + lo parts (round 3): two phases of four MFMAs — (set 0: hi, lo) | (set 1: hi, lo) —
                // with both parts of a set read together while the other set is multiplied. (Rounds 1-2 ran three
                // phases of two MFMAs over one ping-pong pair of query registers: 64 MFMA cycles do not cover an LDS
                // read, and at 64 queries, where the chip holds ~1.6 GHz, the stage chain of ~1190 cycles — not HBM —
                // set the pace. A 64-query tile has registers to spare.)
                read_a<QW>(A[1], buf + a_off + pc1);
                IPX_DS_READ_B128(B[1][0], buf + b_off + pc1, 0);
                IPX_DS_READ_B128(BL[1], buf + b_off + pc1, 4096);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(B[0][0]), "+v"(BL[0]) :: "memory");
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A[0][0]), as_bf16x8(B[0][0]), acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A[0][1]), as_bf16x8(B[0][0]), acc[1][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A[0][0]), as_bf16x8(BL[0]), acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A[0][1]), as_bf16x8(BL[0]), acc[1][0], 0, 0, 0);
                pin_acc<QW>(acc);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[1][0]), "+v"(A[1][1]), "+v"(B[1][0]), "+v"(BL[1]) :: "memory");
            } else {
                // three phases on this stage's buffer: (set 0, lo) | (set 1, hi) | (set 1, lo); each
                // phase's operands are read while the previous phase is multiplied
                read_b<QW>(B[1], buf + b_off + C::PART_B + pc0);
                wait_ab<QW, NB>(A[0], B[0]);
                mfma_set<QW>(acc, A[0], B[0]);
                __builtin_amdgcn_sched_barrier(0);
                read_a<QW>(A[1], buf + a_off + pc1);
                read_b<QW>(B[0], buf + b_off + pc1);
                wait_b<QW, 2 + NB>(B[1]);
                mfma_set<QW>(acc, A[0], B[1]);
                __builtin_amdgcn_sched_barrier(0);
                read_b<QW>(B[1], buf + b_off + C::PART_B + pc1);
                wait_ab<QW, NB>(A[1], B[0]);
                mfma_set<QW>(acc, A[1], B[0]);
                __builtin_amdgcn_sched_barrier(0);
                wait_b<QW, 0>(B[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the stage's last operands are in registers (so nobody reads its buffer any more), the
            // next stage has landed (NBUF - 2 younger stages may still be in flight)
#define IPX_MFMA1(i_, j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A[1][i_]), as_bf16x8(B[1][j_]), acc[i_][j_], 0, 0, 0)
            // the addresses of the next copy pieces are worked out before the barrier, and the first
            // MFMAs of the last phase go out right behind it: the matrix core has work queued while
            // this wave issues the next stage's operand reads
            const StagePieces<QW, QP> sp(ga_i, gb_i, smem + slot * STAGE, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * (NBUF - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // last phase, in pinned issue order: two MFMAs, the next stage's first operand reads (they
            // have the rest of the phase to land), then the copy pieces of the stage NBUF ahead
            // among the MFMAs — a piece costs the wave ~60-180 issue cycles, which the matrix core
            // spends on the MFMAs already queued (and on the other wave of this SIMD)
            {
                if constexpr (QW == 4) {
                    IPX_MFMA1(0, 0); IPX_MFMA1(0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    read_a<QW>(A[0], nbuf + a_off + pc0);
                    read_b<QW>(B[0], nbuf + b_off + pc0);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(0);
                    __builtin_amdgcn_sched_barrier(0);
                    IPX_MFMA1(0, 2); IPX_MFMA1(0, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(1);
                    __builtin_amdgcn_sched_barrier(0);
                    IPX_MFMA1(1, 0); IPX_MFMA1(1, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(2);
                    if constexpr (QP == 2) sp.issue(4);
                    __builtin_amdgcn_sched_barrier(0);
                    IPX_MFMA1(1, 2); IPX_MFMA1(1, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(3);
                    if constexpr (QP == 2) sp.issue(5);
                } else if constexpr (TWO_PHASE) {
                    IPX_MFMA1(0, 0); IPX_MFMA1(1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    read_a<QW>(A[0], nbuf + a_off + pc0);
                    IPX_DS_READ_B128(B[0][0], nbuf + b_off + pc0, 0);
                    IPX_DS_READ_B128(BL[0], nbuf + b_off + pc0, 4096);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(0);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A[1][0]), as_bf16x8(BL[1]), acc[0][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(1);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(A[1][1]), as_bf16x8(BL[1]), acc[1][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(2);
                } else {
                    IPX_MFMA1(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    read_a<QW>(A[0], nbuf + a_off + pc0);
                    read_b<QW>(B[0], nbuf + b_off + pc0);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(0);
                    __builtin_amdgcn_sched_barrier(0);
                    IPX_MFMA1(1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    sp.issue(1);
                    sp.issue(2);
                }
                __builtin_amdgcn_sched_barrier(0);
#undef IPX_MFMA1
            }
            step_ci();
            slot = nslot;
        }
        c0.ks = p.nks - 1;          // the tile is done: on to the cursor's next tile
        c0 = advance(c0);

        // ---- epilogue: threshold filter, rare append to the per-query queues ----
        ++tiles_done;
        uint64_t tr_e0 = 0;
        if (p.trace) tr_e0 = __builtin_readcyclecounter();
        const int docbase = t * HBD + wd * 64 + 4 * h;
        bool episode = lds_read_i32_nodma(&flags[(tiles_done + 1) & 1]) != 0;
        int* myflag = &flags[tiles_done & 1];
        const int sub = fast1 ? (tiles_done - 1) % rs : 0;          // this tile's sub-split
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            const int ql = wq * (32 * QW) + j * 32 + l31;
            const float tau = lds_read_f32_nodma(&taus[ql]);
            // the lane's maximum, kept per group of four registers (four consecutive rows): the hit path votes on the
            // eight groups first, then on the registers of a group somebody passes in (round 3, as in scan_f32_kernel)
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
                uint64_t* qu = myqueues + (size_t)ql * p.cap;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        if (!__any(g8[i * 4 + q4] > tau)) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * q4 + e;
                            const float sc = acc[i][j][r];
                            if (__any(sc > tau)) {
                                const int64_t doc = docbase + i * 32 + (r & 3) + 8 * (r >> 2);
                                if (sc > tau && doc < p.N) {
                                    const int pos = lds_atomic_inc_nodma(&cnts[ql]);
                                    if (pos < p.cap) qu[pos] = make_key(sc, (uint32_t)doc);
                                    else *p.err = 1;
                                    if (pos >= fill_limit) lds_write_i32_nodma(myflag, 1);
                                    if constexpr (QW == 1) lds_atomic_max_u32_nodma(&best[sub * BQ_ + ql], f32_ordered(sc));
                                    ++tr_appends;
                                }
                            }
                        }
                    }
            }
        }
        uint64_t tr_e1 = 0;
        if (p.trace) { tr_e1 = __builtin_readcyclecounter(); tr_epi += tr_e1 - tr_e0; }
        if constexpr (QW == 1) {
            const int rounds = tiles_done / rs;                    // completed rounds over the sub-splits
            if (fast1 && tiles_done == rounds * rs && is_pow4(rounds)) {
                // publish the sub-splits' bests, take a bound from every split's values as the new floor.
                // After the first round every split is at the same point: wait (bounded) until all of
                // them have published, so the next tile already runs against a bound.
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                for (int e = tid; e < rs * BQ_; e += HTHREADS) {
                    const uint32_t b = best[e];
                    if (b) __hip_atomic_store(mypub + e, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const int nvals = p.S * rs;
                if (nvals <= 256) {
                    // the K-th largest of the published bests, K = min(candidates, values): 8 lanes per query
                    const int K = p.k < nvals ? p.k : nvals;
                    const uint32_t* col = p.pub + (size_t)qb * nvals * BQ_ + (tid >> 3);
                    for (int attempt = 0; attempt < (rounds == 1 ? 64 : 1); ++attempt) {
                        const uint32_t pre = kth_largest_published<32, 8>(col, BQ_, nvals, K, tid & 7);
                        if ((tid & 7) == 0) tmin[tid >> 3] = pre;
                        __syncthreads();
                        // queries beyond nq never publish: they do not count as "missing"
                        const bool missing = tid < BQ_ && (int64_t)qb * BQ_ + tid < p.nq && tmin[tid] == 0;
                        if (!__syncthreads_or(missing)) break;
                        __builtin_amdgcn_s_sleep(20);
                    }
                } else {
                const int q = tid & (BQ_ - 1), part = tid / BQ_;           // 8 threads per query: the minimum
                const int per = (nvals + 7) / 8;
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
                }
                if (tid < BQ_) {
                    const uint32_t v = tmin[tid];
                    if (v != 0 && v != 0xffffffffu) taus[tid] = fmaxf(taus[tid], f32_below(f32_unordered(v)));
                }
                __syncthreads();
            }
        }
        episode = episode || (is_pow4(tiles_done) && !fast1);
        if (episode) {
            ++tr_episodes;
            // the sort scratch aliases the stage ring: let every copy land, sort, fetch the
            // NBUF stages again (c0's set-0 operands are already in registers)
            wait_ab<QW, 0>(A[0], B[0]);
            if constexpr (TWO_PHASE) asm volatile("" : "+v"(BL[0]));       // (the wait above covered its read too)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            uint64_t tr_a = 0, tr_b = 0;
            if (p.trace) { tr_a = __builtin_readcyclecounter(); tr_w1 += tr_a - tr_e1; }
            if (tid < 2) flags[tid] = 0;
            if (wave < nsortwaves)
                for (int ql = wave; ql < BQ_; ql += nsortwaves)
                    compact_query_select(p, scratch, cnts, taus, myqueues + (size_t)ql * p.cap, ql, lane,
                                         mypub && !fast1 ? mypub + ql : nullptr);
            if (p.trace) { tr_b = __builtin_readcyclecounter(); tr_own += tr_b - tr_a; }
            __syncthreads();
            if (p.trace) tr_w2 += __builtin_readcyclecounter() - tr_b;
            if (!fast1 && tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
            StageCursor cr = c0;
#pragma unroll 1
            for (int i = 0; i < NBUF; ++i) {
                const int sl = slot + i < NBUF ? slot + i : slot + i - NBUF;
                issue_stage_bf16<QW, QP>(p, smem + sl * STAGE, cr.t, qb, cr.ks, wave, lane);
                cr = advance(cr);
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LOADS * (NBUF - 1)) : "memory");
            __syncthreads();
            if (p.trace) tr_sort += __builtin_readcyclecounter() - tr_e1;
        } else if (!fast1 && (tiles_done & 1) == 0 && is_pow4(tiles_done >> 1)) {
            if (tid < BQ_) poll_union_threshold_n<BQ_>(p, taus, qb, tid);
        }
    }
    if (p.trace) {
        // [gridDim.x][4]: total cycles, filter cycles, compaction cycles (wave 0), appends (all lanes) << 16 | episodes
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tr_appends += __shfl_xor(tr_appends, off);
        if (lane == 0) atomicAdd((unsigned long long*)(p.trace + (size_t)blockIdx.x * 4 + 3), ((unsigned long long)tr_appends << 16));
        if (tid == 0) {
            uint64_t* tr = p.trace + (size_t)blockIdx.x * 4;
            tr[0] = __builtin_readcyclecounter() - tr_begin;
            tr[1] = tr_epi;
            tr[2] = tr_sort;
            uint64_t* tx = p.trace + (size_t)gridDim.x * 4 + (size_t)blockIdx.x * 4;   // episode split (wave 0)
            tx[0] = tr_w1; tx[1] = tr_own; tx[2] = tr_w2; tx[3] = 0;
            atomicAdd((unsigned long long*)(tr + 3), (unsigned long long)tr_episodes);
        }
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

// ---- rescore: candidates (approximate scores) -> exact fp32 scores -> top-k ---------------------
// Margin rule. Notation: q the fp32 query, q^ = the bf16 part(s) the scan multiplied (hi, or
// hi + lo), dq = q - q^; x a row the exact score is defined on, x^ its scanned bf16 copy,
// dx = x - x^ (0 when the index stores bf16 rows only). approx = MFMA chain of q^.x^, exact =
// the fp32 scan's chain of q.x. Then
//   |approx - exact| <= |q^.x^ - q.x| + (accumulation of both chains)
//                     = |q^.dx + dq.x| + acc  <=  (|q| + |dq|) Dx + |dq| Mx + acc =: e
// with Mx = max_x |x|_2 and Dx = max_x |dx|_2 measured when the rows were added
// (row_sumsq_max_kernel: the actual rounding errors, not the worst case 2^-8 |x| — typically
// 0.4 of it), |q| and |dq| computed here from the query itself, and
// acc = acc_coef (|q| + |dq|)(Mx + Dx): (d+1) 2^-24 for the exact chain plus 4 d 2^-24 per query
// part for the MFMA chain (4x in case the matrix core's adds are not round-to-nearest; the
// products of two bf16 numbers are exact in fp32). 0.2 % slack covers the fp32 norms.
// Let a_k be the k-th best approximate score. The k rows with the best approximate scores all have
// exact scores >= a_k - e, so the exact k-th best score is >= a_k - e, so every row of the exact
// top-k has an approximate score >= a_k - 2e. If the candidate list (the K' best approximate
// scores) reaches below a_k - 2e — or holds every admissible row of the index — it contains the
// exact top-k: flag = 0.
// Second chance, after the fact (`post`): when that test fails, every candidate is scored exactly and s_k, the k-th
// best EXACT score among them, is known; it bounds the true k-th best exact score from below. A row outside the list
// has an approximate score <= a_last (the list's last one), hence an exact score <= a_last + e: if a_last + e < s_k
// no such row can enter the top-k (strictly below: ties cannot either) and the list's exact top-k is the answer.
// Since s_k >= a_k - e always, this test passes whenever the first one does, and because the measured |approx - exact|
// is far below the bound e, in practice it asks the list to reach about e below a_k instead of 2e — lists about
// sqrt() as many times k long (Gaussian tails: the score falls with the log of the rank).
// Otherwise flag = 1 and the host repeats the query with a longer list or on the fp32 scan.
struct RescoreParams {
    const float* candD;      // [nq][kc] approximate scores, descending
    const int64_t* candI;    // [nq][kc] id_base + row, -1 = empty (at the end)
    const float* q;          // [nq][d] fp32 queries, row-major
    const int* qmap;         // optional: list qi belongs to query qmap[qi] (exact_rows_path: many lists, one query)
    int shared_list;         // 1: every query rescoring the same list candD / candI [kc] (the outlier rows)
    const void* X;           // packed corpus: fp32 (tiles of 128) or bf16 (tiles of 256)
    const uint32_t* maxsq;   // float bits of max_x |x|^2 and max_x |x - bf16(x)|^2
    float* outD;             // [nq][k]
    int64_t* outI;           // [nq][k]
    int* flags;              // [nq] (+ 1 when err is given)
    const int* err;          // optional: the scan's error word, copied to flags[nq] so that one readback brings both
    const int* outl_count;   // optional (with err): the index's outlier-row count, copied to flags[nq + 1]
    const int* todo = nullptr;   // optional [nq]: 1 = merge_pool_kernel left this query without a list (pool overflow)
    int nq;                  // queries of the pass (position of that copy)
    int64_t id_base;
    float acc_coef;          // accumulation-error coefficient (host: margin_acc_coef)
    int qparts;              // bf16 parts of the query the scan used (1: hi, 2: hi + lo)
    int post;                // 1: try the after-the-fact margin test when the a-priori one fails
    int d, nks, x_is_bf16, k, kc, P;
};

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

constexpr int RS_KB = 4;        // K-stages fetched per round
constexpr int RS_ROW = 80;      // LDS bytes per staged 64-byte line (80: 16 lines, 16 different bank groups)
constexpr int RS_RANK_SORT = 1024;   // lists of up to this many scored candidates are sorted by rank, not bitonic

// NW waves per query: 4 for large batches (the chip is full of workgroups anyway), 16 for small ones, where a query's
// workgroup is alone on its CU and the time is the latency of its dependent rounds — 256 candidates in flight per
// round instead of 64. The rows' K-stages are fetched one group ahead of the fmaf chains that consume them.
template <int NW>
__global__ void __launch_bounds__(NW * 64)
rescore_kernel(const RescoreParams p) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dpad = p.nks * (p.x_is_bf16 ? 32 : 16);
    float* qs = (float*)smem;                                  // [dpad]
    uint64_t* keys = (uint64_t*)(smem + (size_t)((dpad * 4 + 15) & ~15));   // [P]
    __shared__ float red[2 * NW];
    __shared__ int s_m, s_post;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t qi = blockIdx.x;
    const float* q = p.q + (p.qmap ? (int64_t)p.qmap[qi] : qi) * (int64_t)p.d;
    float ss = 0.f, dd = 0.f;
    for (int e = tid; e < dpad; e += NT) {
        const float v = e < p.d ? q[e] : 0.f;
        qs[e] = v;
        ss = fmaf(v, v, ss);
        // what the scan did not see of this element (pack_rows_bf16_kernel's rounding, replayed)
        float rest = v - __uint_as_float((uint32_t)f32_to_bf16_rne(v) << 16);
        if (p.qparts == 2) rest -= __uint_as_float((uint32_t)f32_to_bf16_rne(rest) << 16);
        dd = fmaf(rest, rest, dd);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off); dd += __shfl_xor(dd, off); }
    if (lane == 0) { red[wave] = ss; red[NW + wave] = dd; }
    __syncthreads();
    float qn2 = 0.f, dqn2 = 0.f;
    for (int w = 0; w < NW; ++w) { qn2 += red[w]; dqn2 += red[NW + w]; }
    const float qn = sqrtf(qn2);
    const float dqn = sqrtf(dqn2);
    const float Mx = sqrtf(__uint_as_float(p.maxsq[0])), Dx = sqrtf(__uint_as_float(p.maxsq[1]));
    const float eps = 1.002f * ((qn + dqn) * Dx + dqn * Mx) + p.acc_coef * (qn + dqn) * (Mx + Dx);
    // Every partial sum of either chain is bounded by (|q| + |dq|)(Mx + Dx). Only while that stays well inside the
    // fp32 range is an approximate score a finite number whenever the exact one is: with an inf (or overflowing)
    // element among the scanned rows the bf16 chain can end in NaN — inf * hi + inf * lo with parts of opposite
    // sign, inf * 0 with a query element that rounds to zero — where the exact chain ends in +inf, and a NaN score
    // is never admitted to a candidate list. Such a row would be missing from the list WITHOUT the list being full,
    // so "the list holds every admissible row" proves nothing then: the query is flagged 2 (no bf16 list of any
    // length can be trusted; the host sends it to the exact path directly).
    const float span = (qn + dqn) * (Mx + Dx);
    const bool tame = span < 1e38f;                  // false for NaN / inf

    const float* cD = p.candD + (p.shared_list ? 0 : qi * (int64_t)p.kc);
    const int64_t* cI = p.candI + (p.shared_list ? 0 : qi * (int64_t)p.kc);
    // The list is sorted: valid entries first, scores descending. Its length and the margin cut are counted by all
    // threads at once (one round of loads each) instead of searched by one thread (a dozen dependent loads).
    __shared__ int s_nvalid, s_nge;
    const bool have_list = !(p.todo && p.todo[qi]);
    if (tid == 0) { s_nvalid = 0; s_nge = 0; }
    __syncthreads();
    // a-priori margin test of a full list: the entries at or above thr = a_k - 2 eps (kc >= k; a list that is not
    // full, or shared, never uses the count: its first entries may be padding, cD may be null)
    const float thr = (have_list && !p.shared_list) ? cD[p.k - 1] - 2.f * eps : 0.f;
    if (have_list) {
        int nv = 0, ng = 0;
        for (int i = tid; i < p.kc; i += NT) {
            nv += cI[i] >= 0 ? 1 : 0;
            if (!p.shared_list) ng += cD[i] >= thr ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { nv += __shfl_xor(nv, off); ng += __shfl_xor(ng, off); }
        if (lane == 0) { if (nv) atomicAdd(&s_nvalid, nv); if (ng) atomicAdd(&s_nge, ng); }
    }
    __syncthreads();
    if (tid == 0) {
        int m = 0, flag = 0, post = 0;
        if (!have_list) {
            // merge_pool_kernel could not build this query's list (more than its pool of near-ties at the cut):
            // there is no list to rescore; the query is repeated like one whose list failed the margin test
            flag = 1;
        } else {
        const int nvalid = s_nvalid;
        m = nvalid;
        if (p.shared_list) {
            // the outlier rows: every listed row is scored, nothing was cut off
        } else if (!tame) {
            flag = 2;
        } else if (eps == 0.f) {
            // a zero query (or an all-zero index): every approximate score IS the exact score, the list is in the
            // result order (score desc, id asc) already — its first k entries are the answer, whatever ties follow
            m = nvalid < p.k ? nvalid : p.k;
        } else if (nvalid >= p.kc) {                // the list is full: it may have cut rows off
            if (s_nge < p.kc && s_nge >= p.k) {     // its last score lies below thr: everything from there on is cut
                m = s_nge;                          // (the k-th score itself is >= thr; a NaN thr counts nothing: flagged)
            } else {
                flag = 1;
                post = p.post && eps < INFINITY;    // every candidate is scored; decided behind the sort
            }
        }
        }
        s_m = m;
        s_post = post;
        p.flags[qi] = flag;
        if (p.err && qi == 0) {                            // the scans are done: same stream
            p.flags[p.nq] = *p.err;
            p.flags[p.nq + 1] = p.outl_count ? *p.outl_count : 0;
        }
    }
    __syncthreads();
    const int m = s_m;
    // Exact scores, 16 candidates per wave at a time. A candidate row is 64 contiguous bytes per
    // K-stage, stages one tile-stage block apart: four lanes fetch one such line (16 bytes each),
    // so a wave instruction brings 16 whole lines — one request per line instead of four partial
    // ones — for RS_KB K-stages in a row; the lines go through LDS to lane r < 16, which owns
    // candidate r and runs the ordered fmaf chain (the order of the fp32 scan's MFMA).
    for (int i = tid; i < p.P; i += NT) keys[i] = 0;
    __syncthreads();            // a wave writes keys that other waves' threads cleared
    // short lists are sorted by rank (one pass, no barriers inside) into a second array behind the first
    const bool by_rank = p.P <= RS_RANK_SORT;
    uint64_t* sorted = keys + p.P;                            // [P] when by_rank
    char* stage = (char*)(sorted + (by_rank ? p.P : 0)) + wave * (RS_KB * 16 * RS_ROW);
    const int R = p.x_is_bf16 ? HBD : BD;                 // rows per tile of the packed layout
    const size_t ks_stride = (size_t)R * 64;              // bytes between a row's consecutive K-stages
    for (int g0 = wave * 16; g0 < m; g0 += NW * 16) {
        const int r = lane >> 2, pc = lane & 3;
        const int ci = g0 + r < m ? g0 + r : m - 1;
        const int64_t row = cI[ci] - p.id_base;
        const int64_t tile = row / R;
        const int rr = (int)(row - tile * R);
        const char* src = (const char*)p.X + ((size_t)tile * p.nks * R + rr) * 64 + pc * 16;
        // lane r < 16 computes candidate g0 + r: its row's swizzle
        const int my = g0 + lane < m ? g0 + lane : m - 1;
        const int64_t myrow = lane < 16 ? cI[my] - p.id_base : 0;
        const int mysw = (int)(((myrow % R) >> 2) & 3);
        float acc = 0.f;
        // (four named registers, not an array: hipcc kept an array that is live across the loop in scratch memory)
        static_assert(RS_KB == 4, "the prefetch below is written out for four K-stages per round");
        const int last = p.nks - 1;
        auto fetch = [&](int ks) { return *(const uint4*)(src + (size_t)(ks < last ? ks : last) * ks_stride); };
        uint4 v0 = fetch(0), v1 = fetch(1), v2 = fetch(2), v3 = fetch(3);
        for (int ks0 = 0; ks0 < p.nks; ks0 += RS_KB) {
            *(uint4*)(stage + (0 * 16 + r) * RS_ROW + pc * 16) = v0;
            *(uint4*)(stage + (1 * 16 + r) * RS_ROW + pc * 16) = v1;
            *(uint4*)(stage + (2 * 16 + r) * RS_ROW + pc * 16) = v2;
            *(uint4*)(stage + (3 * 16 + r) * RS_ROW + pc * 16) = v3;
            // the next group is on its way while this one is multiplied (past the end: a harmless re-read of the
            // last stage)
            v0 = fetch(ks0 + 4); v1 = fetch(ks0 + 5); v2 = fetch(ks0 + 6); v3 = fetch(ks0 + 7);
            wave_lds_sync();
            if (lane < 16) {
                const int nb = p.nks - ks0 < RS_KB ? p.nks - ks0 : RS_KB;
                for (int u = 0; u < nb; ++u) {
                    const char* line = stage + (u * 16 + lane) * RS_ROW;
                    if (!p.x_is_bf16) {
                        const f32x4 c0 = *(const f32x4*)(line + ((0 ^ mysw) << 4)), c1 = *(const f32x4*)(line + ((1 ^ mysw) << 4));
                        const f32x4 c2 = *(const f32x4*)(line + ((2 ^ mysw) << 4)), c3 = *(const f32x4*)(line + ((3 ^ mysw) << 4));
                        const float* qq = qs + (ks0 + u) * 16;
                        // MFMA 32x32x2 order inside a group of 8 features: 0,4,1,5,2,6,3,7
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc = fmaf(c0[e], qq[e], acc); acc = fmaf(c1[e], qq[4 + e], acc); }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc = fmaf(c2[e], qq[8 + e], acc); acc = fmaf(c3[e], qq[12 + e], acc); }
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint4 w = *(const uint4*)(line + ((c ^ mysw) << 4));
                            const float* qq = qs + (ks0 + u) * 32 + c * 8;
                            acc = fmaf(bf16_lo(w.x), qq[0], acc); acc = fmaf(bf16_lo(w.z), qq[4], acc);
                            acc = fmaf(bf16_hi(w.x), qq[1], acc); acc = fmaf(bf16_hi(w.z), qq[5], acc);
                            acc = fmaf(bf16_lo(w.y), qq[2], acc); acc = fmaf(bf16_lo(w.w), qq[6], acc);
                            acc = fmaf(bf16_hi(w.y), qq[3], acc); acc = fmaf(bf16_hi(w.w), qq[7], acc);
                        }
                    }
                }
            }
            wave_lds_sync();
        }
        if (lane < 16 && g0 + lane < m && acc > -FLT_MAX)        // the fp32 scan's admission rule
            keys[g0 + lane] = make_key(acc, (uint32_t)myrow);
    }
    __syncthreads();
    const uint64_t* res = keys;
    if (by_rank) {
        // rank of a key = how many keys are larger (distinct rows: distinct keys; the zeros of rows that were not
        // admitted all rank behind every real key and are never read)
        for (int i = tid; i < m; i += NT) {
            const uint64_t key = keys[i];
            if (key) {
                int rank = 0;
                for (int j = 0; j < m; ++j) rank += keys[j] > key ? 1 : 0;
                sorted[rank] = key;
            }
        }
        for (int i = tid; i < m; i += NT) if (!keys[i]) atomicAdd(&s_m, -1);     // s_m: the number of real keys
        __syncthreads();
        res = sorted;
    } else {
        for (int size = 2; size <= p.P; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = tid; i < (p.P >> 1); i += NT) {
                    const int lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));
                    const int hi = lo + stride;
                    const bool desc = (lo & size) == 0;
                    const uint64_t x = keys[lo], y = keys[hi];
                    if ((x < y) == desc) { keys[lo] = y; keys[hi] = x; }
                }
                __syncthreads();
            }
    }
    const int nres = by_rank ? s_m : p.P;                      // entries of res that hold a key (rank sort) / slots (sort)
    if (s_post && tid == 0) {
        const uint64_t key = p.k - 1 < nres ? res[p.k - 1] : 0;        // k <= kc <= P
        if (key && cD[p.kc - 1] + eps < f32_unordered((uint32_t)(key >> 32))) p.flags[qi] = 0;
    }
    for (int i = tid; i < p.k; i += NT) {
        const uint64_t key = i < nres ? res[i] : 0;
        p.outD[qi * p.k + i] = key ? f32_unordered((uint32_t)(key >> 32)) : -FLT_MAX;
        p.outI[qi * p.k + i] = key ? p.id_base + (int64_t)(uint32_t)(~(uint32_t)key) : -1;
    }
}

// exact_rows_path: candidate lists that enumerate every row of the index, kc rows per list
__global__ void __launch_bounds__(256)
all_rows_lists_kernel(int64_t N, int64_t id_base, int64_t total, float* __restrict__ D, int64_t* __restrict__ I) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    D[i] = 0.f;
    I[i] = i < N ? id_base + i : -1;
}

// rows sel[i] of src -> row i of dst (queries that go back to the fp32 scan), and the way back
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ sel, int64_t nsel, int d, float* __restrict__ dst) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / d;
    if (r >= nsel) return;
    const int e = (int)(gid - r * d);
    dst[gid] = src[(int64_t)sel[r] * d + e];
}
__global__ void __launch_bounds__(256)
scatter_results_kernel(const float* __restrict__ sD, const int64_t* __restrict__ sI, const int* __restrict__ sel,
                       int64_t nsel, int k, float* __restrict__ D, int64_t* __restrict__ I) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / k;
    if (r >= nsel) return;
    const int e = (int)(gid - r * k);
    D[(int64_t)sel[r] * k + e] = sD[gid];
    I[(int64_t)sel[r] * k + e] = sI[gid];
}

}  // namespace ipx
