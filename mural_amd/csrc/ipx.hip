// ipx.hip — host side of the C ABI declared in include/ipx.h.
//
// Owns the device memory of a flat inner-product index (packed layout, see ipx_kernels.h)
// and launches the gfx950 kernels. Every entry point cites the reference call it replaces
// in include/ipx.h. No CPU fallback: without a HIP device every compute call fails with
// IPX_ERR_HIP.
#include "../../include/ipx.h"
#include "ipx_bf16_kernels.h"
#include "ipx_f32_small_kernels.h"
#include "ipx_bf16_w16_kernels.h"
// The default library ships ONE large-batch bf16 scan (the 16-wave kernel, 2-deep ring: 16x16x32 MFMA with one query
// part, 32x32x16 with hi + lo parts). The kernels and kernel variants that were built to find it and never won — the
// 8-wave tile, the ping-pong and direct-corpus-operand kernels, the 384 x 256 tile on 12 waves, deeper rings, issue
// orders, workgroup maps — only exist
// in a -DIPX_DIAG build (libipx_diag.so: __graft_entry__.build_diag(), used by scripts/), where the "bf16_*" knobs of
// ipx_set_param select them.
#ifdef IPX_DIAG
#include "ipx_bf16_pp_kernels.h"
#include "ipx_bf16_dx_kernels.h"
#include "ipx_bf16_w12_kernels.h"
#endif
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

// Workspace of ipx_search_noalloc (include/ipx.h): device memory that exists from the moment the code object is
// loaded, so that entry point never allocates. Every GPU the library runs on holds its own copy at its own
// address until the process ends. In host code this NAME is only a handle for hipGetSymbolAddress — never a
// device address: the per-device address is resolved once (scratch_ptr below) and only that pointer is handed
// to kernels, memsets and copies.
static __device__ __align__(256) unsigned char scratch[IPX_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int IPX_MAX_DEVICES = 64;
void* g_scratch_ptr[IPX_MAX_DEVICES] = {};      // address of `scratch` on each device, resolved once
std::mutex g_scratch_mu;

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define IPX_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? IPX_ERR_NOMEM : IPX_ERR_HIP,           \
                        std::string(#call) + ": " + hipGetErrorString(e_));                \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return 0;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(IPX_ERR_NOMEM, "hipMalloc(" + std::to_string(need) + " bytes): " + hipGetErrorString(e));
        }
        bytes = need;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// Makes `device` current for the duration of a call and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err != hipSuccess) { prev = -1; err = hipSetDevice(device); }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// Every entry point that works on a GPU: the index's device for the duration of the call, the caller's afterwards
// (faiss leaves the caller's current device alone too).
#define IPX_ON_DEVICE(dev_)                                                                                       \
    DeviceGuard ipx_guard_(dev_);                                                                                 \
    if (ipx_guard_.err != hipSuccess)                                                                             \
        return fail(IPX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(ipx_guard_.err))

// Address of the static workspace on `device` (which must be current). Resolved once per device.
int scratch_ptr(int device, void** out) {
    if (device < 0 || device >= IPX_MAX_DEVICES) return fail(IPX_ERR_ARG, "device index beyond " + std::to_string(IPX_MAX_DEVICES));
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(IPX_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return 0;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and device while the size does not grow (the call is
// not free, and the small-batch paths launch several kernels per search that need it)
int ensure_dynamic_lds(const void* kern, int bytes) {
    struct Entry { const void* kern; int device; int bytes; };
    static std::mutex mu;
    static std::vector<Entry> seen;
    if (bytes <= 64 * 1024) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    for (Entry& e : seen)
        if (e.kern == kern && e.device == dev) {
            if (e.bytes >= bytes) return 0;
            IPX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            e.bytes = bytes;
            return 0;
        }
    IPX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    seen.push_back({kern, dev, bytes});
    return 0;
}

int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

struct ipx_index {
    int d = 0, nks = 0, device = 0;
    int num_cus = 0;         // multiProcessorCount of `device`, read once at create
    int* last_err = nullptr; // error word of the most recent search (ipx_check reads it)
    int64_t ntotal = 0;
    int64_t cap_tiles = 0;   // tiles of 128 rows the corpus buffer holds
    int64_t id_base = 0;
    float* Xp = nullptr;     // packed fp32 corpus (storage 0, 1)
    // storage: 0 = fp32 rows, fp32 scan; 1 = fp32 rows + a bf16 copy: bf16 scan, fp32 rescoring;
    //          2 = bf16 rows only (rows given as bf16, or rounded on add): bf16 scan, exact
    //              rescoring of the stored bf16 values against the fp32 query
    int storage = 0;
    int nks_h = 0;           // K-stages of 32 bf16
    int64_t cap_tiles_h = 0; // tiles of 256 rows the bf16 buffer holds
    uint16_t* Xh = nullptr;  // packed bf16 corpus (storage 1, 2)
    DevBuf Qp, cand, partD, partI, err, stageIn, stageQ, stageW, outD, outI, trace, pub;
    DevBuf outl, olist, sideD, sideI;   // outlier rows of an index with fp32 rows + bf16 copy (row_sumsq_max_kernel)
    bool outl_thr_set = false;
    int n_outliers = 0;                 // as of the last bf16 pass
    DevBuf maxsq, qfused, candD, candI, flags, sel, qsel, selD, selI, poolH, poolW, poolQ;
    DevBuf deepD, deepI, ceilD, ceilI;   // searches beyond DEEP_MAX_K: one pass's lists, the next pass's ceilings
    DevBuf deepK, deepN;                 // deep lists (k > IPX_MAX_K): the splits' unsorted lists and their lengths
    // Smallest depth at which the margin rule sent most of a batch to the fp32 scan (storage 1): a few rows with a far
    // larger norm than the rest widen every query's margin, the bf16 passes are then wasted work, and searches at that
    // depth or beyond go to the fp32 scan directly. Forgotten when rows are added or a parameter changes.
    int bf16_fail_k = INT32_MAX;
    // What this index's data has taught the automatic list length (rows whose norms spread widen every margin: MURAL
    // representations are unnormalised): the automatic candidate count is multiplied by kc_boost / 8 after a large batch
    // that sent more than 0.1 % of its queries to the repeat pass, and once the longest list is reached large batches
    // start with hi + lo query parts. Forgotten when rows are added or a parameter changes.
    int kc_boost = 8;
    bool parts_boost = false;
    int* hflags = nullptr;               // pinned host copy of a pass's margin flags + the error word
    size_t hflags_n = 0;
    void* pinned = nullptr;          // host staging of ipx_add (two slices)
    size_t pinned_bytes = 0;
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    bool trace_on = false;
    bool timing = false;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false, ev_merge = false, ev_rescore = false;
    int last_wgs = 0, last_S = 0, last_cap = 0;
    int last_kc = 0;                 // candidates per query of the last bf16 scan (0 = fp32 scan)
    int64_t last_fallback = 0;       // queries whose first bf16 pass failed the sufficiency check
    int64_t last_exact = 0;          // ... of which the repeat pass failed too: answered by the fp32 scan / the exact pass
    int max_splits = 0;      // 0 = automatic
    int force_kc = 0;        // 0 = automatic candidate count
    int query_parts = 0;     // 0 = automatic; 1 / 2 bf16 parts per query in the bf16 scan
    int small_f32_limit = 192;     // fp32 scans of up to this many queries take scan_f32_small_kernel
    bool bf16_shape16 = true;      // 16-wave scan, one query part: v_mfma_f32_16x16x32_bf16 (measured 2-4 % faster) / 32x32x16
    int f32_helpers = 16;           // fp32 large-tile scan: helper list slots per query block (0 = no work stealing)
    int run_tiles = 0;             // fp32 large-tile scan: corpus tiles per run of the dynamic split (0 = automatic)
    int bf16_pp = 0;               // large-batch bf16 scan, one query part: 1 / 2 = ping-pong kernel (2: with s_setprio)
    int bf16_map = 0;              // large-batch bf16 scan: ScanParams::map
    int bf16_ring = 2;             // 16-wave bf16 scan, one query part: stage-ring depth (2 = default, 3, 4)
    int bf16_occ = 1;              // ... workgroups per CU with the 2-deep ring (1, 2)
    int bf16_variant = 0;          // 16-wave bf16 scan, one query part: issue order behind the barrier (0..3) + 4 * non-temporal corpus pieces
    int bf16_waves = 0;            // large-batch bf16 scan: 8 (scan_bf16_kernel<4>) or 16 waves (scan_bf16_w16_kernel); 0 = automatic
    int margin_post = 1;           // bf16 scan: the after-the-fact margin test (rescore_kernel) and the shorter lists it allows
    bool no_small_batch = false;   // diagnostic: always use the 256-query bf16 scan variant
    bool force_f32_scan = false;   // storage 1: use the fp32 scan although a bf16 copy exists
    size_t tile_bytes() const { return (size_t)nks * ipx::STAGE_A_BYTES; }
    size_t tile_bytes_h() const { return (size_t)nks_h * ipx::HSTAGE_A; }
};

namespace {

int grow_corpus(ipx_index* ix, int64_t need_tiles, hipStream_t st) {
    if (need_tiles <= ix->cap_tiles) return 0;
    int64_t new_tiles = need_tiles;
    if (ix->cap_tiles > 0) new_tiles = std::max(need_tiles, ix->cap_tiles + ix->cap_tiles / 2);
    float* nb = nullptr;
    const size_t nbytes = (size_t)new_tiles * ix->tile_bytes();
    hipError_t e = hipMalloc((void**)&nb, nbytes);
    if (e != hipSuccess && new_tiles > need_tiles) {   // retry with the exact size
        new_tiles = need_tiles;
        e = hipMalloc((void**)&nb, (size_t)new_tiles * ix->tile_bytes());
    }
    if (e != hipSuccess)
        return fail(IPX_ERR_NOMEM, "hipMalloc(corpus, " + std::to_string(nbytes) + " bytes): " + hipGetErrorString(e));
    const size_t old_bytes = (size_t)ix->cap_tiles * ix->tile_bytes();
    const size_t new_bytes = (size_t)new_tiles * ix->tile_bytes();
    hipError_t ce = old_bytes ? hipMemcpyAsync(nb, ix->Xp, old_bytes, hipMemcpyDeviceToDevice, st) : hipSuccess;
    // rows of a partly filled tile must read as zeros
    if (ce == hipSuccess) ce = hipMemsetAsync((char*)nb + old_bytes, 0, new_bytes - old_bytes, st);
    if (ce == hipSuccess && ix->Xp) ce = hipStreamSynchronize(st);
    if (ce != hipSuccess) {
        (void)hipFree(nb);
        return fail(IPX_ERR_HIP, std::string("grow corpus: ") + hipGetErrorString(ce));
    }
    if (ix->Xp) (void)hipFree(ix->Xp);
    ix->Xp = nb;
    ix->cap_tiles = new_tiles;
    return 0;
}

int grow_corpus_h(ipx_index* ix, int64_t need_tiles, hipStream_t st) {
    if (need_tiles <= ix->cap_tiles_h) return 0;
    int64_t new_tiles = need_tiles;
    if (ix->cap_tiles_h > 0) new_tiles = std::max(need_tiles, ix->cap_tiles_h + ix->cap_tiles_h / 2);
    uint16_t* nb = nullptr;
    hipError_t e = hipMalloc((void**)&nb, (size_t)new_tiles * ix->tile_bytes_h());
    if (e != hipSuccess && new_tiles > need_tiles) {
        new_tiles = need_tiles;
        e = hipMalloc((void**)&nb, (size_t)new_tiles * ix->tile_bytes_h());
    }
    if (e != hipSuccess)
        return fail(IPX_ERR_NOMEM, std::string("hipMalloc(bf16 corpus): ") + hipGetErrorString(e));
    const size_t old_bytes = (size_t)ix->cap_tiles_h * ix->tile_bytes_h();
    const size_t new_bytes = (size_t)new_tiles * ix->tile_bytes_h();
    hipError_t ce = old_bytes ? hipMemcpyAsync(nb, ix->Xh, old_bytes, hipMemcpyDeviceToDevice, st) : hipSuccess;
    if (ce == hipSuccess) ce = hipMemsetAsync((char*)nb + old_bytes, 0, new_bytes - old_bytes, st);
    if (ce == hipSuccess && ix->Xh) ce = hipStreamSynchronize(st);
    if (ce != hipSuccess) {
        (void)hipFree(nb);
        return fail(IPX_ERR_HIP, std::string("grow bf16 corpus: ") + hipGetErrorString(ce));
    }
    if (ix->Xh) (void)hipFree(ix->Xh);
    ix->Xh = nb;
    ix->cap_tiles_h = new_tiles;
    return 0;
}

int launch_pack_bf16(const void* src, int dtype, int64_t n, int64_t n_pad, int d, uint16_t* dst, int64_t row0,
                     int rows_per_tile, int nks_h, int parts, hipStream_t st, int swz = 0, void* zero = nullptr,
                     size_t zero_bytes = 0) {
    const int64_t total = n_pad * nks_h * 4;
    if (total == 0) return 0;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(IPX_ERR_ARG, "pack: too many rows in one call");
    hipLaunchKernelGGL(ipx::pack_rows_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dtype, n, n_pad, d,
                       dst, row0, rows_per_tile, nks_h, parts, swz, (uint32_t*)zero, (int64_t)(zero_bytes / 4));
    IPX_HIP(hipGetLastError());
    return 0;
}

int launch_pack(const void* src, int dtype, int64_t n, int64_t n_pad, int d, float* dst, int64_t row0,
                int rows_per_tile, int nks, const float* w, int A, int logits, hipStream_t st) {
    const int64_t total = n_pad * nks * 4;
    if (total == 0) return 0;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(IPX_ERR_ARG, "pack: too many rows in one call");
    hipLaunchKernelGGL(ipx::pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dtype, n, n_pad, d,
                       dst, row0, rows_per_tile, nks, w, A, logits);
    IPX_HIP(hipGetLastError());
    return 0;
}

// one slice of rows, already on the device: every copy this index keeps + the norm bound
int add_device_rows(ipx_index* ix, const void* rows, int dtype, int64_t m, int64_t row0, hipStream_t st) {
    if (ix->storage != 2)
        if (int rc = launch_pack(rows, dtype, m, m, ix->d, ix->Xp, row0, ipx::BD, ix->nks, nullptr, 0, 0, st)) return rc;
    if (ix->storage != 0) {
        if (int rc = launch_pack_bf16(rows, dtype, m, m, ix->d, ix->Xh, row0, ipx::HBD, ix->nks_h, 1, st)) return rc;
        // the margin rule's corpus-side numbers: max |x|_2 of the values the exact score is defined on
        // (the fp32 rows in storage 1, the stored bf16 rows in storage 2) and max |x - bf16(x)|_2
        ipx::OutlierState* outl = ix->storage == 1 ? (ipx::OutlierState*)ix->outl.p : nullptr;
        if (outl && !ix->outl_thr_set) {
            hipLaunchKernelGGL(ipx::outlier_threshold_kernel, dim3(1), dim3(1024), 0, st, rows, dtype, m, ix->d, outl);
            IPX_HIP(hipGetLastError());
            ix->outl_thr_set = true;
        }
        const int blocks = (int)std::min<int64_t>((m + 3) / 4, 4096);
        hipLaunchKernelGGL(ipx::row_sumsq_max_kernel, dim3(blocks), dim3(256), 0, st, rows, dtype, m, ix->d,
                           (uint32_t*)ix->maxsq.p, ix->storage, outl, row0);
        IPX_HIP(hipGetLastError());
        if (outl) {      // listed rows never become candidates of the bf16 scan: their copies turn into NaN
            hipLaunchKernelGGL(ipx::nan_rows_bf16_kernel, dim3(ipx::OUTLIER_MAX), dim3(64), 0, st, ix->Xh, ix->nks_h, outl);
            IPX_HIP(hipGetLastError());
        }
    }
    return 0;
}

// room for n rows in every copy this index keeps
int grow_storage(ipx_index* ix, int64_t n, hipStream_t st) {
    if (ix->storage != 2)
        if (int rc = grow_corpus(ix, (n + ipx::BD - 1) / ipx::BD, st)) return rc;
    if (ix->storage != 0) {
        if (int rc = grow_corpus_h(ix, (n + ipx::HBD - 1) / ipx::HBD, st)) return rc;
        if (!ix->maxsq.p) {
            if (int rc = ix->maxsq.ensure(256)) return rc;
            IPX_HIP(hipMemsetAsync(ix->maxsq.p, 0, 256, st));
        }
        if (ix->storage == 1 && !ix->outl.p) {
            if (int rc = ix->outl.ensure(sizeof(ipx::OutlierState))) return rc;
            IPX_HIP(hipMemsetAsync(ix->outl.p, 0, sizeof(ipx::OutlierState), st));
        }
    }
    return 0;
}

int dtype_size(int dtype) { return dtype == IPX_F32 ? 4 : 2; }

struct Geometry { int nqb, ntiles, S, R, G, cap; };

// Deep lists: the splits' unsorted lists of every query (ScanParams::deepK / deepN) -> sorted top k_out (dD / dI).
int launch_deep_finish(ipx_index* ix, int S, int k_list, int k_out, int64_t nq, float* dD, int64_t* dI, hipStream_t st) {
    const int64_t total = (int64_t)S * k_list;
    const int P_max = (int)std::min<int64_t>(ipx::DEEP_LDS_KEYS, next_pow2((int)std::min<int64_t>(total, ipx::DEEP_LDS_KEYS)));
    const int lds = std::max(P_max, 2) * 8 + (S + 1) * 4 + 256 * 4;
    if (lds > 64 * 1024)
        IPX_HIP(hipFuncSetAttribute((const void*)ipx::deep_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(ipx::deep_finish_kernel, dim3((unsigned)nq), dim3(1024), lds, st, (const uint64_t*)ix->deepK.p,
                       (const int*)ix->deepN.p, S, k_list, k_out, std::max(P_max, 2), ix->id_base, dD, dI);
    IPX_HIP(hipGetLastError());
    return 0;
}

// ---- fp32 scan of a small batch (scan_f32_small_kernel: 32 or 64 queries per workgroup, two workgroups per CU):
// pack -> scan -> merge. Same scores, bit for bit, as the large tile. Geometry, workspace and launch are separate
// so that ipx_search_noalloc can run the same kernels on the static workspace.
struct SmallGeom {
    int qt, bq, nqb, ntiles, S, R, G, cap, ring, nsortwaves;
    size_t qp_bytes, cand_bytes, partD_bytes, partI_bytes, flags_bytes, ctr_bytes, pub_bytes;
    size_t total() const {      // every buffer rounded up to 256 bytes, + the error word
        size_t t = 256;
        for (size_t b : {qp_bytes, cand_bytes, partD_bytes, partI_bytes, flags_bytes, pub_bytes}) t += (b + 255) / 256 * 256;
        return t;
    }
};
struct SmallBufs { void *Qp, *cand, *partD, *partI, *flags, *pub; int* err; };

// `split_cap` > 0 bounds the corpus splits per query block (ipx_set_param "max_splits", or what fits a workspace)
SmallGeom small_geometry(const ipx_index* ix, int64_t nq, int k, int split_cap) {
    SmallGeom g;
    g.qt = nq <= 32 ? 1 : 2;
    g.bq = 32 * g.qt;
    g.nqb = (int)((nq + g.bq - 1) / g.bq);
    g.ntiles = (int)((ix->ntotal + ipx::BD - 1) / ipx::BD);
    const int resident = 2 * ix->num_cus;                   // two 512-thread workgroups per CU
    const int npairs = (g.ntiles + 1) / 2;                  // the kernel steps through pairs of corpus tiles
    int S = std::max(1, resident / g.nqb);
    if (split_cap > 0) S = std::min(S, split_cap);
    g.R = std::max(2, std::min(32, npairs / (S * 64)));
    const int nruns = (npairs + g.R - 1) / g.R;
    g.S = std::min(S, nruns);
    g.G = g.nqb * g.S;
    // queue: k survivors of a compaction + 2 steps (4 tiles) of appends before the next one runs + room to fill
    g.cap = next_pow2(k + 4 * ipx::BD + k / 2);
    g.ring = g.qt == 1 ? ipx::SCfg<1>::RING : ipx::SCfg<2>::RING;
    g.nsortwaves = std::max(1, std::min(8, g.ring / (g.cap * 8)));
    const size_t nout = (size_t)nq * k;
    g.qp_bytes = (size_t)g.nqb * ix->nks * g.bq * 64;
    g.cand_bytes = (size_t)g.G * g.bq * g.cap * sizeof(uint64_t);
    g.partD_bytes = g.S > 1 ? nout * g.S * sizeof(float) : 0;
    g.partI_bytes = g.S > 1 ? nout * g.S * sizeof(int64_t) : 0;
    g.flags_bytes = g.S > 1 ? (size_t)nq * sizeof(int) : 0;
    g.ctr_bytes = ((size_t)g.nqb * sizeof(uint32_t) + 255) / 256 * 256;
    // published bounds: up to IPX_MAX_SUBSPLITS values per split and query (short-list path)
    g.pub_bytes = g.ctr_bytes + (size_t)g.G * g.bq * ipx::IPX_MAX_SUBSPLITS * sizeof(uint32_t);
    return g;
}

// Launches only: no allocation, no synchronisation. `record`: HIP events around the kernels (ipx_set_timing).
int scan_f32_small_launch(ipx_index* ix, const SmallGeom& g, const SmallBufs& b, const float* dq, const float* dw, int A,
                          int logits, int64_t nq, int k, float* dD, int64_t* dI, hipStream_t st, bool record, bool primary,
                          bool set_attr = true) {
    const int S = g.S;
    IPX_HIP(hipMemsetAsync(b.pub, 0, g.pub_bytes, st));
    if (record) IPX_HIP(hipEventRecord(ix->ev[0], st));
    if (int rc = launch_pack(dq, IPX_F32, nq, (int64_t)g.nqb * g.bq, ix->d, (float*)b.Qp, 0, g.bq, ix->nks, dw, A, logits, st))
        return rc;
    if (record) IPX_HIP(hipEventRecord(ix->ev[1], st));

    ipx::ScanParams p;
    p.Xp = ix->Xp;
    p.Qp = (const float*)b.Qp;
    p.cand = (uint64_t*)b.cand;
    p.outD = S > 1 ? (float*)b.partD : dD;
    p.outI = S > 1 ? (int64_t*)b.partI : dI;
    p.err = b.err;
    p.run_ctr = (uint32_t*)b.pub;
    p.pub = (uint32_t*)((char*)b.pub + g.ctr_bytes);
    p.run_tiles = g.R;
    p.r = (k + S - 1) / S;
    p.trace = nullptr;
    if (ix->trace_on && ix->trace.bytes >= (size_t)g.G * 20 * sizeof(uint64_t)) p.trace = (uint64_t*)ix->trace.p;
    p.N = ix->ntotal;
    p.id_base = ix->id_base;
    p.nq = (int)nq;
    p.nks = ix->nks;
    p.k = k;
    p.cap = g.cap;
    p.S = S;
    p.S_out = S;
    p.helpers = 0;
    p.help_ctr = nullptr;
    p.ntiles = g.ntiles;
    const int lds = std::max(g.ring, g.nsortwaves * g.cap * 8) + ipx::LDS_STATE_BYTES;
    {
        void (*kern)(const ipx::ScanParams, const int) = g.qt == 1 ? ipx::scan_f32_small_kernel<1> : ipx::scan_f32_small_kernel<2>;
        if (set_attr) IPX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kern, dim3(g.G), dim3(512), lds, st, p, g.nsortwaves);
    }
    IPX_HIP(hipGetLastError());
    if (record) IPX_HIP(hipEventRecord(ix->ev[2], st));

    if (S > 1) {
        const int P = next_pow2(k);
        const int mlds = P * 24;
        if (mlds > 64 * 1024)
            IPX_HIP(hipFuncSetAttribute((const void*)ipx::merge_topk_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, mlds));
        // many short lists: cut by the union bound and sort once; the fold kernel then only serves
        // queries whose pool overflowed
        const int* todo = nullptr;
        if (S >= 16) {
            hipLaunchKernelGGL(ipx::merge_pool_kernel, dim3((unsigned)nq), dim3(256), 0, st, (const float*)b.partD,
                               (const int64_t*)b.partI, S, k, k, (int64_t)k, (int64_t)S * k, dD, dI, (int*)b.flags, 0);
            IPX_HIP(hipGetLastError());
            todo = (const int*)b.flags;
        }
        hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nq), dim3(256), mlds, st,
                           (const float*)b.partD, (const int64_t*)b.partI, S, nq, k, k, P, 1,
                           (int64_t)k, (int64_t)S * k, dD, dI, todo, (int64_t)0);
        IPX_HIP(hipGetLastError());
        if (record) {
            IPX_HIP(hipEventRecord(ix->ev[3], st));
            ix->ev_merge = true;
        }
    }
    if (primary) {
        ix->ev_valid = record;
        ix->last_wgs = g.G;
        ix->last_S = S;
        ix->last_cap = g.cap;
    }
    return 0;
}

int scan_f32_small_path(ipx_index* ix, const float* dq, const float* dw, int A, int logits, int64_t nq, int k,
                        float* dD, int64_t* dI, hipStream_t st, bool primary) {
    const SmallGeom g = small_geometry(ix, nq, k, ix->max_splits);
    if (int rc = ix->Qp.ensure(g.qp_bytes)) return rc;
    if (int rc = ix->cand.ensure(g.cand_bytes)) return rc;
    if (int rc = ix->err.ensure(256)) return rc;
    if (g.S > 1) {
        if (int rc = ix->partD.ensure(g.partD_bytes)) return rc;
        if (int rc = ix->partI.ensure(g.partI_bytes)) return rc;
        if (int rc = ix->flags.ensure(g.flags_bytes)) return rc;
    }
    if (int rc = ix->pub.ensure(g.pub_bytes)) return rc;
    if (ix->trace_on)
        if (int rc = ix->trace.ensure((size_t)g.G * 20 * sizeof(uint64_t))) return rc;
    const SmallBufs b{ix->Qp.p, ix->cand.p, ix->partD.p, ix->partI.p, ix->flags.p, ix->pub.p, (int*)ix->err.p};
    return scan_f32_small_launch(ix, g, b, dq, dw, A, logits, nq, k, dD, dI, st, primary && ix->timing, primary);
}

// fp32 scan of nq row-major fp32 queries already on the device (A > 0: aspect embeddings + weights,
// fused by the pack kernel): pack -> scan -> merge. Results (device) in dD / dI.
int scan_f32_path(ipx_index* ix, const float* dq, const float* dw, int A, int logits, int64_t nq, int k,
                  float* dD, int64_t* dI, hipStream_t st, bool primary, const float* ceilD = nullptr,
                  const int64_t* ceilI = nullptr) {
    // up to 192 queries: the 32 / 64-query tile does a fraction of the 256-query tile's matrix work
    // (passes of a deep search that carry a ceiling stay on the 256-query kernel, the one that knows ceilings)
    // Lists beyond IPX_MAX_K entries (and every pass that carries a ceiling) take the deep form of the kernel: selections
    // on 4 bytes of LDS per queue slot, the splits' lists left unsorted, deep_finish_kernel behind it.
    const bool deep = k > IPX_MAX_K || ceilI;
    if (k > ipx::DEEP_MAX_K) return fail(IPX_ERR_ARG, "search: one pass serves k <= " + std::to_string(ipx::DEEP_MAX_K));
    if (nq <= ix->small_f32_limit && !ix->no_small_batch && !deep)
        return scan_f32_small_path(ix, dq, dw, A, logits, nq, k, dD, dI, st, primary);
    const bool record = primary && ix->timing;
    const int d = ix->d, nks = ix->nks;
    const int64_t N = ix->ntotal;
    const size_t nout = (size_t)nq * k;
    const int nqb = (int)((nq + ipx::BQ - 1) / ipx::BQ);
    const int ntiles = (int)((N + ipx::BD - 1) / ipx::BD);
    const int cap = std::max(512, next_pow2(2 * k));
    // deep lists: as many selecting waves as fit the LDS of a CU that holds two workgroups (80 KiB each) — or one, when a
    // single wave's selection scratch is larger than that (k > 8192)
    const int sel_bytes = (cap + ipx::SEL_AUX) * 4;
    const int half_cu = 80 * 1024 - ipx::LDS_STATE_BYTES - 64, whole_cu = 160 * 1024 - ipx::LDS_STATE_BYTES - 64;
    const bool two_per_cu = !deep || sel_bytes <= half_cu;
    const int nsel = deep ? std::max(1, std::min(4, (two_per_cu ? half_cu : whole_cu) / sel_bytes)) : 0;
    // geometry: (up to) all workgroups resident at once, 2 per CU; S of them share a query block.
    const int resident = (two_per_cu ? 2 : 1) * ix->num_cus;
    // A query block's S workgroups take runs of R corpus tiles from a shared counter.
    int S = std::max(1, resident / nqb);
    if (ix->max_splits > 0) S = std::min(S, ix->max_splits);
    // (runs of ~1/256 of a workgroup's share: 8.8M rows -> 14 tiles; 59-tile runs measured 0.5 % slower)
    const int R = ix->run_tiles > 0 ? ix->run_tiles : std::max(4, std::min(64, ntiles / (S * 256)));
    const int nruns = (ntiles + R - 1) / R;
    S = std::min(S, nruns);
    // spare list slots per query block for workgroups that finish their own block early (scan_f32_kernel)
    const int H = (S > 1 && nqb > 1 && nruns > 4 * S && !deep) ? ix->f32_helpers : 0;
    const int S_out = S + H;
    const int G = nqb * S;

    if (int rc = ix->Qp.ensure((size_t)nqb * nks * ipx::STAGE_B_BYTES)) return rc;
    if (int rc = ix->cand.ensure((size_t)G * ipx::BQ * cap * sizeof(uint64_t))) return rc;
    if (int rc = ix->err.ensure(256)) return rc;
    if (deep) {
        if (int rc = ix->deepK.ensure(nout * S_out * sizeof(uint64_t))) return rc;
        if (int rc = ix->deepN.ensure((size_t)nq * S_out * sizeof(int))) return rc;
    } else if (S > 1) {
        if (int rc = ix->partD.ensure(nout * S_out * sizeof(float))) return rc;
        if (int rc = ix->partI.ensure(nout * S_out * sizeof(int64_t))) return rc;
    }
    // per search: run counters [nqb] and helper-slot counters [nqb] (each padded to 256 B), then published
    // thresholds [G][256], all zero
    const size_t half_ctr = ((size_t)nqb * sizeof(uint32_t) + 255) / 256 * 256;
    const size_t ctr_bytes = 2 * half_ctr;
    const size_t pub_bytes = ctr_bytes + (size_t)G * ipx::BQ * sizeof(uint32_t);
    if (int rc = ix->pub.ensure(pub_bytes)) return rc;
    IPX_HIP(hipMemsetAsync(ix->pub.p, 0, pub_bytes, st));

    if (record) IPX_HIP(hipEventRecord(ix->ev[0], st));
    if (int rc = launch_pack(dq, IPX_F32, nq, (int64_t)nqb * ipx::BQ, d, (float*)ix->Qp.p, 0, ipx::BQ, nks, dw, A,
                             logits, st))
        return rc;
    if (record) IPX_HIP(hipEventRecord(ix->ev[1], st));

    ipx::ScanParams p;
    p.Xp = ix->Xp;
    p.Qp = (const float*)ix->Qp.p;
    p.cand = (uint64_t*)ix->cand.p;
    p.outD = S > 1 ? (float*)ix->partD.p : dD;
    p.outI = S > 1 ? (int64_t*)ix->partI.p : dI;
    p.err = (int*)ix->err.p;
    p.run_ctr = (uint32_t*)ix->pub.p;
    p.pub = (uint32_t*)((char*)ix->pub.p + ctr_bytes);
    p.run_tiles = R;
    p.r = (k + S - 1) / S;
    p.trace = nullptr;
    if (ix->trace_on) {
        if (int rc = ix->trace.ensure((size_t)G * 20 * sizeof(uint64_t))) return rc;   // 4 per workgroup (+16 in stamp builds)
        p.trace = (uint64_t*)ix->trace.p;
    }
    p.N = N;
    p.id_base = ix->id_base;
    p.nq = (int)nq;
    p.nks = nks;
    p.k = k;
    p.cap = cap;
    p.S = S;
    p.S_out = S > 1 ? S_out : 1;
    p.helpers = H;
    p.help_ctr = (uint32_t*)((char*)ix->pub.p + half_ctr);
    p.ntiles = ntiles;
    if (H > 0) {   // helper slots nobody takes must read as empty lists
        const int64_t nfill = (int64_t)nout * S_out;
        hipLaunchKernelGGL(ipx::fill_empty_kernel, dim3((unsigned)((nfill + 255) / 256)), dim3(256), 0, st, (float*)ix->partD.p,
                           (int64_t*)ix->partI.p, nfill);
        IPX_HIP(hipGetLastError());
    }
    p.ceilD = ceilD;
    p.ceilI = ceilI;
    p.nselect = nsel;
    if (deep) {
        p.deepK = (uint64_t*)ix->deepK.p;
        p.deepN = (int*)ix->deepN.p;
    }
    const int lds = std::max(2 * ipx::STAGE_BYTES, deep ? nsel * sel_bytes : 4 * cap * 8) + ipx::LDS_STATE_BYTES;
    void (*kern)(const ipx::ScanParams) = !deep ? ipx::scan_f32_kernel<false, false>
                                          : ceilI ? ipx::scan_f32_kernel<true, true> : ipx::scan_f32_kernel<false, true>;
    if (lds > 64 * 1024)
        IPX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(G), dim3(ipx::NTHREADS), lds, st, p);
    IPX_HIP(hipGetLastError());
    if (record) IPX_HIP(hipEventRecord(ix->ev[2], st));

    if (deep) {
        if (int rc = launch_deep_finish(ix, S_out, k, k, nq, dD, dI, st)) return rc;
        if (record) {
            IPX_HIP(hipEventRecord(ix->ev[3], st));
            ix->ev_merge = true;
        }
    } else if (S > 1) {
        const int P = next_pow2(k);
        const int mlds = P * 24;
        if (mlds > 64 * 1024)
            IPX_HIP(hipFuncSetAttribute((const void*)ipx::merge_topk_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, mlds));
        // partials are [nq][S][k]: shard stride k, query stride S*k; lists arrive sorted. Many lists (a few query
        // blocks spread over every CU: 512 lists at 256 queries, where folding them one by one took 0.98 ms): cut by
        // the union bound and sort once; the fold kernel then only serves queries whose pool overflowed. (From 64 lists
        // on: with the headline's 34 lists of 6980 queries the fold alone takes 0.29 ms, pool + fold 0.40.)
        const int* todo = nullptr;
        if (S_out >= 64) {
            if (int rc = ix->flags.ensure((size_t)nq * sizeof(int))) return rc;
            hipLaunchKernelGGL(ipx::merge_pool_kernel, dim3((unsigned)nq), dim3(256), 0, st, (const float*)ix->partD.p,
                               (const int64_t*)ix->partI.p, S_out, k, k, (int64_t)k, (int64_t)S_out * k, dD, dI,
                               (int*)ix->flags.p, 0);
            IPX_HIP(hipGetLastError());
            todo = (const int*)ix->flags.p;
        }
        hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nq), dim3(256), mlds, st,
                           (const float*)ix->partD.p, (const int64_t*)ix->partI.p, S_out, nq, k, k, P, 1,
                           (int64_t)k, (int64_t)S_out * k, dD, dI, todo, (int64_t)0);
        IPX_HIP(hipGetLastError());
        if (record) {
            IPX_HIP(hipEventRecord(ix->ev[3], st));
            ix->ev_merge = true;
        }
    }
    if (primary) {
        ix->ev_valid = ix->timing;
        ix->last_wgs = G;
        ix->last_S = S;
        ix->last_cap = cap;
    }
    return 0;
}

// k beyond DEEP_MAX_K (faiss.IndexFlatIP.search takes any k, reference retriever.py:19): one fp32 scan per
// DEEP_MAX_K ranks. Pass j admits only hits strictly after the last hit of pass j-1 in the result order (score
// descending, row ascending), so its top list is ranks [j * DEEP_MAX_K, ...) of the one total order: the
// concatenation is bit for bit what a single pass of depth k would return. Ranks beyond ntotal are (-FLT_MAX, -1)
// without a scan. HIP events (ipx_set_timing) cover the first pass.
int scan_f32_deep(ipx_index* ix, const float* dq, const float* dw, int A, int logits, int64_t nq, int k,
                  float* dD, int64_t* dI, hipStream_t st) {
    const int64_t k_live = std::min<int64_t>(k, ix->ntotal);
    if (k_live < k) {
        const int64_t nout = nq * (int64_t)k;
        hipLaunchKernelGGL(ipx::fill_empty_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, dD, dI, nout);
        IPX_HIP(hipGetLastError());
    }
    const int step = ipx::DEEP_MAX_K;
    if (int rc = ix->deepD.ensure((size_t)nq * step * sizeof(float))) return rc;
    if (int rc = ix->deepI.ensure((size_t)nq * step * sizeof(int64_t))) return rc;
    if (int rc = ix->ceilD.ensure((size_t)nq * sizeof(float))) return rc;
    if (int rc = ix->ceilI.ensure((size_t)nq * sizeof(int64_t))) return rc;
    for (int64_t off = 0; off < k_live; off += step) {
        const int kk = (int)std::min<int64_t>(step, k_live - off);
        if (int rc = scan_f32_path(ix, dq, dw, A, logits, nq, kk, (float*)ix->deepD.p, (int64_t*)ix->deepI.p, st, off == 0,
                                   off ? (const float*)ix->ceilD.p : nullptr, off ? (const int64_t*)ix->ceilI.p : nullptr))
            return rc;
        const int64_t n = nq * (int64_t)kk;
        hipLaunchKernelGGL(ipx::deep_collect_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const float*)ix->deepD.p, (const int64_t*)ix->deepI.p, nq, kk, dD, dI, k, (int)off,
                           (float*)ix->ceilD.p, (int64_t*)ix->ceilI.p, (int*)nullptr);
        IPX_HIP(hipGetLastError());
    }
    return 0;
}

// rescore_kernel over `nblocks` lists: 16 waves per list for the few lists of a small batch (latency), 4 otherwise.
int launch_rescore(const ipx::RescoreParams& rp, int64_t nblocks, hipStream_t st) {
    const int fixed = ((rp.nks * (rp.x_is_bf16 ? 32 : 16) * 4 + 15) & ~15) + rp.P * 8 + (rp.P <= ipx::RS_RANK_SORT ? rp.P * 8 : 0);
    const int per_wave = ipx::RS_KB * 16 * ipx::RS_ROW;
    const int nw = (nblocks <= 128 && fixed + 16 * per_wave <= 160 * 1024) ? 16 : 4;
    const int rlds = fixed + nw * per_wave;
    void (*kern)(const ipx::RescoreParams) = nw == 16 ? ipx::rescore_kernel<16> : ipx::rescore_kernel<4>;
    if (int rc = ensure_dynamic_lds((const void*)kern, rlds)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(nw * 64), rlds, st, rp);
    IPX_HIP(hipGetLastError());
    return 0;
}

// Accumulation-error coefficient of the margin rule (rescore_kernel; the rounding terms are
// measured, not bounded: see row_sumsq_max_kernel): the exact fp32 chain contributes (d+1) 2^-24,
// the MFMA chain d 2^-24 per query part, taken 4x in case the matrix core's internal adds are not
// round-to-nearest; 0.2 % slack for the fp32 norms.
float margin_acc_coef(int d_pad, int qparts) {
    return (float)((1.0 + 4.0 * qparts) * (d_pad + 1) / 16777216.0 * 1.002);
}

// One bf16 pass over nq row-major fp32 queries (device): pack -> bf16 scan keeping kc candidates
// -> merge -> exact rescoring into dD / dI. Synchronises the stream and returns in `flagged` the
// queries whose candidate list could not be proven sufficient (their rows of dD / dI are not final), and in `hard`
// those of them that no bf16 list of any length can serve (rescore_kernel's flag 2: a non-finite norm bound).
int bf16_pass(ipx_index* ix, const float* qf, int64_t nq, int k, int kc, int qparts, float* dD, int64_t* dI,
              hipStream_t st, bool primary, std::vector<int>& flagged, std::vector<int>& hard) {
    const int d = ix->d, nks = ix->nks_h;
    const int64_t N = ix->ntotal;
    const size_t ncand = (size_t)nq * kc;
    const bool timing = primary && ix->timing;

    // candidate lists beyond MAX_KC entries: the 16-wave kernel with its lists left unsorted + deep_finish_kernel
    const bool deep = kc > 4096;
    // up to 64 queries: the 64-query variant (a quarter of the MFMA work per corpus byte)
    const bool small = nq <= 64 && !ix->no_small_batch && !deep;
    const int bq = small ? 64 : 256;
    const int stage_b = bq * 64 * qparts;
    const int nqb = (int)((nq + bq - 1) / bq);
    // large batches, one query part: the 12-wave kernel on 384-row tiles (ipx_bf16_w12_kernels.h)
#ifdef IPX_DIAG
    const bool w12 = !small && !deep && qparts == 1 && ix->bf16_pp == 0 && ix->bf16_waves == 12;
    constexpr int w12_rows = ipx::XCfgBase::TD, w12_threads = ipx::XCfgBase::THREADS;
    const int w12_ring = ix->bf16_ring == 3 ? ipx::XCfg<3>::RING : ipx::XCfg<2>::RING;
#else
    constexpr bool w12 = false;
    constexpr int w12_rows = ipx::HBD, w12_threads = 0, w12_ring = 0;
#endif
    const int tile_rows = w12 ? w12_rows : ipx::HBD;
    const int ntiles = (int)((N + tile_rows - 1) / tile_rows);
    // large batches, one query part: the 16-wave kernel, whose ring depth (and with a 64-KiB ring its workgroups per
    // CU) can be chosen
    const bool w16_1 = !small && (deep || (ix->bf16_pp == 0 &&
                       ((qparts == 1 && (ix->bf16_waves == 16 || ix->bf16_waves == 0) && ix->bf16_shape16) ||
                        (qparts == 2 && (ix->bf16_waves == 16 || ix->bf16_waves == 0)))));
    // Ring depth. Two stages (one in flight) measured fastest on the 8.8M x 768 x 6980-query workload: 84 ms against
    // 93 ms with four — the workgroups that stream the same corpus split then keep each other in step through L2
    // (a workgroup that runs ahead misses L2 and waits on HBM, its followers hit), L2 fetch traffic per launch
    // halves (396 -> 170-190 GB) and the chip holds 2.06 instead of 1.82 GHz.
    const int w16_ring = deep ? 2 : w16_1 && !w12 && ix->bf16_ring != 4 ? ix->bf16_ring : 0;   // 0 = the four-deep ring of WCfg
    const int occ = (w16_ring == 2 && qparts == 1 && ix->bf16_occ == 2 && !deep) ? 2 : 1;
    const int resident = occ * ix->num_cus;                 // one workgroup per CU (two with the 64-KiB ring, if asked)
    int S = std::max(1, resident / nqb);
    if (ix->max_splits > 0) S = std::min(S, ix->max_splits);
    int R = 4;                                               // tiles per run: a power of two in [4, 64]
    while (R < 64 && 2 * R <= ntiles / (S * 64)) R *= 2;
    const int nruns = (ntiles + R - 1) / R;
    S = std::min(S, nruns);
    const int G = nqb * S;
    const bool w16k = !small && (deep || w12 || ((ix->bf16_waves == 16 || ix->bf16_waves == 0) && ix->bf16_pp == 0));
    // queue: kc survivors of a compaction + 2 tiles of appends before the next one runs + room to fill, rounded up to a
    // power of two (a capacity cut to the next multiple of 512 was measured at kc = 3000: 5120 instead of 8192 slots
    // gives the selections two more waves but forces extra episodes — 176 vs 146 ms of scan)
    // (the 12-wave kernel: two 384-row tiles of appends; a multiple of 512 is all its selections need)
    const int cap = w12 ? (kc + 2 * w12_rows + kc / 2 + 511) / 512 * 512 : next_pow2(kc + 2 * ipx::HBD + kc / 2);
    const int cap_sort = next_pow2(cap);
    const int ring = w12 ? w12_ring : small ? (qparts == 2 ? ipx::HCfg<1, 2>::RING : ipx::HCfg<1, 1>::RING)
                     : w16_ring == 2 ? (qparts == 2 ? ipx::WCfg<2, 2>::RING : ipx::WCfg<1, 2>::RING)
                     : (w16_ring == 3 && qparts == 1) ? ipx::WCfg<1, 3>::RING
                           : (qparts == 2 ? ipx::HCfg<4, 2>::RING : ipx::HCfg<4, 1>::RING);
    // LDS for the episodes: the stage ring's bytes, or for the 16-wave kernel (one workgroup per CU, so the CU's LDS
    // is its own) 157 KiB whatever the ring depth
    const int sort_bytes = w16k && occ == 1 ? std::max(ring, 157 * 1024) : ring;
    const int nsortwaves = deep ? 0 : std::max(1, std::min(8, sort_bytes / (cap_sort * 8)));
    const int nselect = w16k ? std::max(1, std::min(w12 ? 12 : 16, sort_bytes / ((cap + ipx::W16_SEL_AUX) * 4))) : 0;

    if (int rc = ix->Qp.ensure((size_t)nqb * nks * stage_b)) return rc;
    if (int rc = ix->cand.ensure((size_t)G * bq * cap * sizeof(uint64_t))) return rc;
    if (int rc = ix->err.ensure(256)) return rc;
    if (int rc = ix->candD.ensure(ncand * sizeof(float))) return rc;
    if (int rc = ix->candI.ensure(ncand * sizeof(int64_t))) return rc;
    if (int rc = ix->flags.ensure((size_t)(nq + 2) * sizeof(int))) return rc;      // + the error word, the outlier count
    if (deep) {
        if (int rc = ix->deepK.ensure(ncand * S * sizeof(uint64_t))) return rc;
        if (int rc = ix->deepN.ensure((size_t)nq * S * sizeof(int))) return rc;
    } else if (S > 1) {
        if (int rc = ix->partD.ensure(ncand * S * sizeof(float))) return rc;
        if (int rc = ix->partI.ensure(ncand * S * sizeof(int64_t))) return rc;
    }
    // per search, zeroed: run counters [nqb] (unused by these kernels), meeting counters [S] (scan_bf16_dx_kernel)
    const size_t ctr_bytes = ((size_t)(nqb + S) * sizeof(uint32_t) + 255) / 256 * 256;
    // published bounds: up to 4 values per split and query on the small-batch short-list path
    const size_t pub_bytes = ctr_bytes + (size_t)G * bq * (small ? 4 : 1) * sizeof(uint32_t);
    if (int rc = ix->pub.ensure(pub_bytes)) return rc;          // (cleared by the pack kernel below)

    // large batches, one query part: the ping-pong kernels (both operands in LDS: any even number of K-stages;
    // corpus operand straight into registers: a multiple of 6)
    const bool pp = !small && !deep && qparts == 1 && (ix->bf16_pp == 1 || ix->bf16_pp == 2) && (nks & 1) == 0;
    const bool dx = !small && !deep && qparts == 1 && (ix->bf16_pp == 3 || ix->bf16_pp == 4) && nks % 6 == 0;
    if (timing) IPX_HIP(hipEventRecord(ix->ev[0], st));
    if (int rc = launch_pack_bf16(qf, IPX_F32, nq, (int64_t)nqb * bq, d, (uint16_t*)ix->Qp.p, 0, bq, nks, qparts, st, dx ? 1 : 0,
                                  ix->pub.p, pub_bytes))
        return rc;
    if (timing) IPX_HIP(hipEventRecord(ix->ev[1], st));

    ipx::ScanParams p;
    p.Xp = (const float*)ix->Xh;
    p.Qp = (const float*)ix->Qp.p;
    p.cand = (uint64_t*)ix->cand.p;
    p.outD = S > 1 ? (float*)ix->partD.p : (float*)ix->candD.p;
    p.outI = S > 1 ? (int64_t*)ix->partI.p : (int64_t*)ix->candI.p;
    p.err = (int*)ix->err.p;
    p.run_ctr = (uint32_t*)ix->pub.p;
    p.pub = (uint32_t*)((char*)ix->pub.p + ctr_bytes);
    p.run_tiles = R;
    p.r = (kc + S - 1) / S;
    p.trace = nullptr;
    if (ix->trace_on && primary) {
        if (int rc = ix->trace.ensure((size_t)G * 20 * sizeof(uint64_t))) return rc;
        IPX_HIP(hipMemsetAsync(ix->trace.p, 0, (size_t)G * 4 * sizeof(uint64_t), st));
        p.trace = (uint64_t*)ix->trace.p;
    }
    p.N = N;
    p.id_base = ix->id_base;
    p.nq = (int)nq;
    p.nks = nks;
    p.k = kc;
    p.cap = cap;
    p.S = S;
    p.S_out = S;
    p.helpers = 0;
    p.help_ctr = nullptr;
    p.ntiles = ntiles;
    p.map = small ? 0 : ix->bf16_map == 1 ? nqb : ix->bf16_map;      // 1 = one group of all query blocks
    p.nselect = nselect;
    if (deep) {
        p.deepK = (uint64_t*)ix->deepK.p;
        p.deepN = (int*)ix->deepN.p;
    }
    const int lds = std::max({ring, nsortwaves * cap_sort * 8, nselect * (cap + ipx::W16_SEL_AUX) * 4}) + ipx::LDS_STATE_BYTES;
    {
        // large batches: the 16-wave kernel (8.8M x 768, 6980 queries, one query part: 93 vs 102 ms for the 8-wave
        // kernel at the same ring depth; hi + lo parts at depth 1000, whole search: 197 ms with the 2-deep ring, 203
        // with 3, 217 on 8 waves)
        const bool w16 = !small && !w12 && (deep || ix->bf16_waves == 16 || ix->bf16_waves == 0);
#ifdef IPX_DIAG
        void (*kern)(const ipx::ScanParams, const int) =
            small ? (qparts == 2 ? ipx::scan_bf16_kernel<1, 2> : ipx::scan_bf16_kernel<1, 1>)
            : w16 ? (qparts == 2 ? ipx::scan_bf16_w16_kernel<2, false>
                                 : (ix->bf16_shape16 ? ipx::scan_bf16_w16_kernel<1, true> : ipx::scan_bf16_w16_kernel<1, false>))
                  : (qparts == 2 ? ipx::scan_bf16_kernel<4, 2> : ipx::scan_bf16_kernel<4, 1>);
        if (w16 && qparts == 1 && !ix->bf16_shape16) {
            switch (ix->bf16_variant) {
                case 1: kern = ipx::scan_bf16_w16_kernel<1, false, 1, false>; break;
                case 2: kern = ipx::scan_bf16_w16_kernel<1, false, 2, false>; break;
                case 3: kern = ipx::scan_bf16_w16_kernel<1, false, 3, false>; break;
                case 4: kern = ipx::scan_bf16_w16_kernel<1, false, 0, true>; break;
                case 5: kern = ipx::scan_bf16_w16_kernel<1, false, 1, true>; break;
                case 6: kern = ipx::scan_bf16_w16_kernel<1, false, 2, true>; break;
                case 7: kern = ipx::scan_bf16_w16_kernel<1, false, 3, true>; break;
                default: break;
            }
        }
        if (deep || (w16 && w16_ring == 2)) kern = qparts == 2 ? ipx::scan_bf16_w16_kernel<2, false, 0, false, 2> : ipx::scan_bf16_w16_kernel<1, true, 0, false, 2>;
        if (w16 && w16_ring == 3 && qparts == 1) kern = ipx::scan_bf16_w16_kernel<1, true, 0, false, 3>;
        if (pp) kern = ix->bf16_pp == 2 ? ipx::scan_bf16_pp_kernel<true> : ipx::scan_bf16_pp_kernel<false>;
        if (dx) kern = ix->bf16_pp == 4 ? ipx::scan_bf16_dx_kernel<true> : ipx::scan_bf16_dx_kernel<false>;
#else
        // (the knobs that select anything else cannot be set in this build: ipx_set_param)
        void (*kern)(const ipx::ScanParams, const int) =
            small ? (qparts == 2 ? ipx::scan_bf16_kernel<1, 2> : ipx::scan_bf16_kernel<1, 1>)
                  : (qparts == 2 ? ipx::scan_bf16_w16_kernel<2, false, 0, false, 2> : ipx::scan_bf16_w16_kernel<1, true, 0, false, 2>);
#endif
#ifdef IPX_DIAG
        if (w12) kern = ix->bf16_ring == 3 ? ipx::scan_bf16_w12_kernel<3> : ipx::scan_bf16_w12_kernel<2>;
#endif
        if (int rc = ensure_dynamic_lds((const void*)kern, std::max(lds, 64 * 1024 + 1))) return rc;
        hipLaunchKernelGGL(kern, dim3(G), dim3((pp || dx) ? 512 : w12 ? w12_threads : w16 ? 1024 : ipx::HTHREADS), lds, st, p, nsortwaves);
    }
    IPX_HIP(hipGetLastError());
    if (timing) IPX_HIP(hipEventRecord(ix->ev[2], st));

    const int* pool_todo = nullptr;
    if (deep) {
        if (int rc = launch_deep_finish(ix, S, kc, kc, nq, (float*)ix->candD.p, (int64_t*)ix->candI.p, st)) return rc;
    } else if (S > 1) {
        const int P = next_pow2(kc);
        const int mlds = P * 24;
        if (mlds > 64 * 1024)
            IPX_HIP(hipFuncSetAttribute((const void*)ipx::merge_topk_kernel,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, mlds));
        // many lists (few queries spread over every CU): cut by the union bound and sort once;
        // the fold kernel then only serves queries whose pool overflowed
        // (a query whose pool overflows — more near-ties at the cut than the pool holds — is folded list by list in the
        // same kernel: no second launch that does nothing in the common case)
        if (S >= 16) {
            if (mlds + 52 * 1024 > 64 * 1024)
                if (int rc = ensure_dynamic_lds((const void*)ipx::merge_pool_kernel, std::max(mlds, 64 * 1024 + 1))) return rc;
            hipLaunchKernelGGL(ipx::merge_pool_kernel, dim3((unsigned)nq), dim3(256), mlds, st, (const float*)ix->partD.p,
                               (const int64_t*)ix->partI.p, S, kc, kc, (int64_t)kc, (int64_t)S * kc,
                               (float*)ix->candD.p, (int64_t*)ix->candI.p, (int*)nullptr, P);
            IPX_HIP(hipGetLastError());
        } else {
            hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nq), dim3(256), mlds, st,
                               (const float*)ix->partD.p, (const int64_t*)ix->partI.p, S, nq, kc, kc, P, 1,
                               (int64_t)kc, (int64_t)S * kc, (float*)ix->candD.p, (int64_t*)ix->candI.p, (const int*)nullptr,
                               (int64_t)0);
            IPX_HIP(hipGetLastError());
        }
    }
    if (timing) {
        IPX_HIP(hipEventRecord(ix->ev[3], st));
        ix->ev_merge = true;
    }

    // the pass's flags (which queries failed the margin check; behind them the scan's error word and the outlier count)
    // are written by the rescoring kernel straight into pinned host memory: one wait, no copy
    if (ix->hflags_n < (size_t)nq + 2) {
        IPX_HIP(hipStreamSynchronize(st));          // (an earlier pass's kernel may still write the old block)
        if (ix->hflags) (void)hipHostFree(ix->hflags);
        ix->hflags = nullptr;
        ix->hflags_n = 0;
        const size_t want = std::max<size_t>((size_t)nq + 2, 4096);
        hipError_t e = hipHostMalloc((void**)&ix->hflags, want * sizeof(int), hipHostMallocDefault);
        if (e != hipSuccess) { ix->hflags = nullptr; return fail(IPX_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
        ix->hflags_n = want;
    }
    ipx::RescoreParams rp;
    rp.candD = (const float*)ix->candD.p;
    rp.candI = (const int64_t*)ix->candI.p;
    rp.todo = pool_todo;
    rp.q = qf;
    rp.qmap = nullptr;
    rp.x_is_bf16 = ix->storage == 2;
    rp.X = rp.x_is_bf16 ? (const void*)ix->Xh : (const void*)ix->Xp;
    rp.nks = rp.x_is_bf16 ? ix->nks_h : ix->nks;
    rp.maxsq = (const uint32_t*)ix->maxsq.p;
    rp.outD = dD;
    rp.outI = dI;
    rp.flags = ix->hflags;
    rp.err = (const int*)ix->err.p;
    rp.outl_count = ix->outl.p ? &((const ipx::OutlierState*)ix->outl.p)->count : nullptr;
    rp.shared_list = 0;
    rp.nq = (int)nq;
    rp.id_base = ix->id_base;
    rp.acc_coef = margin_acc_coef(ix->nks_h * ipx::HBK, qparts);
    rp.qparts = qparts;
    rp.post = ix->margin_post;
    rp.d = d;
    rp.k = k;
    rp.kc = kc;
    rp.P = next_pow2(kc);
    if (int rc = launch_rescore(rp, nq, st)) return rc;
    if (timing) {
        IPX_HIP(hipEventRecord(ix->ev[4], st));
        ix->ev_rescore = true;
    }
    if (primary) {
        ix->ev_valid = ix->timing;
        ix->last_wgs = G;
        ix->last_S = S;
        ix->last_cap = cap;
        ix->last_kc = kc;
    }

    // which queries failed the margin check? (the one host round trip of a pass)
    const int* flags = ix->hflags;
    IPX_HIP(hipStreamSynchronize(st));
    ix->n_outliers = std::min(flags[nq + 1], ipx::OUTLIER_MAX);
    if (flags[nq]) {
        (void)hipMemsetAsync(ix->err.p, 0, 256, st);
        return fail(IPX_ERR_STATE, "search: candidate queue overflow (internal invariant violated)");
    }
    flagged.clear();
    hard.clear();
    for (int64_t i = 0; i < nq; ++i)
        if (flags[(size_t)i]) (flags[(size_t)i] == 2 ? hard : flagged).push_back((int)i);
    return 0;
}

// Rows `sel` of the fp32 queries qf, searched again by `again` (a longer-list bf16 pass or the
// fp32 scan), results written over rows `sel` of dD / dI.
template <class F>
int redo_queries(ipx_index* ix, const float* qf, const std::vector<int>& sel, int k, float* dD, int64_t* dI,
                 hipStream_t st, F again) {
    const int d = ix->d;
    const int64_t nsel = (int64_t)sel.size();
    if (int rc = ix->sel.ensure((size_t)nsel * sizeof(int))) return rc;
    if (int rc = ix->qsel.ensure((size_t)nsel * d * sizeof(float))) return rc;
    if (int rc = ix->selD.ensure((size_t)nsel * k * sizeof(float))) return rc;
    if (int rc = ix->selI.ensure((size_t)nsel * k * sizeof(int64_t))) return rc;
    IPX_HIP(hipMemcpyAsync(ix->sel.p, sel.data(), (size_t)nsel * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ipx::gather_rows_kernel, dim3((unsigned)((nsel * d + 255) / 256)), dim3(256), 0, st, qf,
                       (const int*)ix->sel.p, nsel, d, (float*)ix->qsel.p);
    IPX_HIP(hipGetLastError());
    if (int rc = again((const float*)ix->qsel.p, nsel, (float*)ix->selD.p, (int64_t*)ix->selI.p)) return rc;
    hipLaunchKernelGGL(ipx::scatter_results_kernel, dim3((unsigned)((nsel * k + 255) / 256)), dim3(256), 0, st,
                       (const float*)ix->selD.p, (const int64_t*)ix->selI.p, (const int*)ix->sel.p, nsel, k, dD, dI);
    IPX_HIP(hipGetLastError());
    IPX_HIP(hipStreamSynchronize(st));   // `sel` (host memory) is read by the copy above
    return 0;
}

int merge_deep(const float* dD, const int64_t* dI, int S, int64_t nq, int k_in, int k_out, float* oD, int64_t* oI, hipStream_t st);

// Last resort of an index without fp32 rows (storage 2): queries whose near-ties outnumber the longest candidate
// list (thousands of duplicate rows at the top, say) get the exact score of EVERY stored row — the rescoring kernel
// over lists that enumerate the index, IPX_MAX_K rows per list, then one merge of the lists' top-k. Slow (a full
// pass of scalar fmaf chains per query: milliseconds per million rows) and exact; one query at a time.
int exact_rows_path(ipx_index* ix, const float* qf, const std::vector<int>& sel, int k, float* dD, int64_t* dI, hipStream_t st) {
    const int kc = k > IPX_MAX_K ? ipx::DEEP_MAX_K : IPX_MAX_K;       // rows per list (>= k)
    const int64_t N = ix->ntotal;
    const int64_t nlists = (N + kc - 1) / kc, total = nlists * kc;
    if (nlists > 0x7fffffffLL) return fail(IPX_ERR_ARG, "search: index too large for the exact pass");
    DevBuf& lD = ix->selD;      // the repeat passes' buffers are free here (their results are already scattered)
    DevBuf& lI = ix->selI;
    if (int rc = lD.ensure((size_t)total * sizeof(float))) return rc;
    if (int rc = lI.ensure((size_t)total * sizeof(int64_t))) return rc;
    if (int rc = ix->partD.ensure((size_t)nlists * k * sizeof(float))) return rc;
    if (int rc = ix->partI.ensure((size_t)nlists * k * sizeof(int64_t))) return rc;
    if (int rc = ix->sel.ensure((size_t)nlists * sizeof(int))) return rc;
    if (int rc = ix->deepD.ensure((size_t)nlists * sizeof(int))) return rc;      // the lists' (unused) margin flags
    hipLaunchKernelGGL(ipx::all_rows_lists_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, N, ix->id_base,
                       total, (float*)lD.p, (int64_t*)lI.p);
    IPX_HIP(hipGetLastError());
    ipx::RescoreParams rp;
    rp.candD = (const float*)lD.p;
    rp.candI = (const int64_t*)lI.p;
    rp.q = qf;
    rp.qmap = (const int*)ix->sel.p;
    rp.x_is_bf16 = ix->storage == 2;
    rp.X = rp.x_is_bf16 ? (const void*)ix->Xh : (const void*)ix->Xp;
    rp.nks = rp.x_is_bf16 ? ix->nks_h : ix->nks;
    rp.maxsq = (const uint32_t*)ix->maxsq.p;
    rp.outD = (float*)ix->partD.p;
    rp.outI = (int64_t*)ix->partI.p;
    rp.flags = (int*)ix->deepD.p;
    rp.err = nullptr;
    rp.outl_count = nullptr;
    rp.shared_list = 0;
    rp.nq = 0;
    rp.id_base = ix->id_base;
    rp.acc_coef = 1.f;
    rp.qparts = 1;
    rp.post = 0;
    rp.d = ix->d;
    rp.k = k;
    rp.kc = kc;
    rp.P = next_pow2(kc);
    const int P = next_pow2(k), mlds = P * 24;
    if (mlds > 64 * 1024 && k <= IPX_MAX_K)
        IPX_HIP(hipFuncSetAttribute((const void*)ipx::merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mlds));
    for (int f : sel) {
        IPX_HIP(hipMemsetD32Async((hipDeviceptr_t)ix->sel.p, f, (size_t)nlists, st));
        if (int rc = launch_rescore(rp, nlists, st)) return rc;
        // the lists' top-k (sorted) -> row f of the result
        if (k > IPX_MAX_K) {
            if (int rc = merge_deep((const float*)ix->partD.p, (const int64_t*)ix->partI.p, (int)nlists, 1, k, k,
                                    dD + (size_t)f * k, dI + (size_t)f * k, st))
                return rc;
            continue;
        }
        hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3(1), dim3(256), mlds, st, (const float*)ix->partD.p,
                           (const int64_t*)ix->partI.p, (int)nlists, (int64_t)1, k, k, P, 1, (int64_t)k, (int64_t)0,
                           dD + (size_t)f * k, dI + (size_t)f * k, (const int*)nullptr, (int64_t)0);
        IPX_HIP(hipGetLastError());
    }
    return 0;
}

// The outlier rows (row_sumsq_max_kernel) never appear in a bf16 pass's result: every query scores them exactly on
// the side (the rescoring kernel over the one shared list) and the two sorted lists are merged — (score desc, id asc),
// the order of every other merge. Queries that go on to the fp32 scan afterwards have these rows of dD / dI rewritten.
int merge_outlier_rows(ipx_index* ix, const float* qf, int64_t nq, int k, float* dD, int64_t* dI, hipStream_t st) {
    const size_t n = (size_t)nq * k;
    if (int rc = ix->olist.ensure(ipx::OUTLIER_MAX * sizeof(int64_t))) return rc;
    if (int rc = ix->sideD.ensure(2 * n * sizeof(float))) return rc;
    if (int rc = ix->sideI.ensure(2 * n * sizeof(int64_t))) return rc;
    hipLaunchKernelGGL(ipx::outlier_list_kernel, dim3(1), dim3(256), 0, st, (const ipx::OutlierState*)ix->outl.p,
                       ix->id_base, (int64_t*)ix->olist.p);
    IPX_HIP(hipGetLastError());
    IPX_HIP(hipMemcpyAsync(ix->sideD.p, dD, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    IPX_HIP(hipMemcpyAsync(ix->sideI.p, dI, n * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    ipx::RescoreParams rp;
    rp.candD = nullptr;                      // a shared list is never cut: its approximate scores are not read
    rp.candI = (const int64_t*)ix->olist.p;
    rp.q = qf;
    rp.qmap = nullptr;
    rp.shared_list = 1;
    rp.x_is_bf16 = 0;
    rp.X = (const void*)ix->Xp;
    rp.nks = ix->nks;
    rp.maxsq = (const uint32_t*)ix->maxsq.p;
    rp.outD = (float*)ix->sideD.p + n;
    rp.outI = (int64_t*)ix->sideI.p + n;
    rp.flags = (int*)ix->flags.p;            // the pass's flags are on the host already
    rp.err = nullptr;
    rp.outl_count = nullptr;
    rp.nq = (int)nq;
    rp.id_base = ix->id_base;
    rp.acc_coef = 1.f;
    rp.qparts = 1;
    rp.post = 0;
    rp.d = ix->d;
    rp.k = k;
    rp.kc = ipx::OUTLIER_MAX;
    rp.P = ipx::OUTLIER_MAX;
    if (int rc = launch_rescore(rp, nq, st)) return rc;
    if (k > IPX_MAX_K)       // lists beyond one LDS sort: the two blocks [2][nq][k] through the deep merge
        return merge_deep((const float*)ix->sideD.p, (const int64_t*)ix->sideI.p, 2, nq, k, k, dD, dI, st);
    const int P = next_pow2(k), mlds = P * 24;
    if (mlds > 64 * 1024)
        IPX_HIP(hipFuncSetAttribute((const void*)ipx::merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mlds));
    hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nq), dim3(256), mlds, st, (const float*)ix->sideD.p,
                       (const int64_t*)ix->sideI.p, 2, nq, k, k, P, 1, (int64_t)n, (int64_t)k, dD, dI, (const int*)nullptr,
                       (int64_t)0);
    IPX_HIP(hipGetLastError());
    return 0;
}

// Longest candidate list of a bf16 pass (the API's single-pass depth IPX_MAX_K is about result lists; candidate
// lists may be longer: their queues live in HBM, a selection takes 4 bytes of LDS per queue slot, the final sort 8).
constexpr int MAX_KC = 4096;

// bf16 scan + exact rescoring. Queries that fail the margin check are repeated with the longest
// candidate list, and whatever still fails goes to the fp32 scan (storage 1 keeps fp32 rows) or to the exact pass over
// the bf16 rows (storage 2).
int scan_bf16_path(ipx_index* ix, const float* dq, const float* dw, int A, int logits, int64_t nq, int k,
                   float* dD, int64_t* dI, hipStream_t st) {
    const int d = ix->d;
    // One bf16 part per query and 5k candidates (2k when only the query is rounded) while that
    // list stays short; deeper searches split the query into hi + lo parts — the margin then only
    // carries the rows' rounding (none in storage 2) — and keep 2k (1.25k) candidates.
    // (Gaussian 768-d rows, k = 100: the margin set holds ~270 rows; 4k left 2 of 6980 queries
    // to the repeat pass, 4.5k and up none.)
    // Batches of up to 64 queries are bound by the corpus stream, not the matrix core: the second
    // query part is free there, and its short list (<= the number of splits) lets every split
    // publish just its best score (scan_bf16_kernel, fast1).
    // Measured on the 8.8M x 768 Gaussian workload, storage 1, large batches: one part with 3k candidates
    // beats two parts with 2k (k = 500: 129 vs 190 ms; k = 700 with 2048: 154 vs 199 ms, nothing repeated;
    // at 2.1k every query was). Round 2, candidate lists up to 4096: k = 1000 one part with 3072 188 ms against 202 ms
    // for two parts with 2000, k = 800 185 vs 195 ms, nothing repeated; with 4096 it is slower again (212 ms: the
    // appends and selections grow with the list). So: one part for every depth the bf16 path serves, its list 5k
    // while that stays within 2048, else 3k.
    const int mult1 = ix->storage == 2 ? 2 : 5;
    const bool small = nq <= 64 && !ix->no_small_batch;
    const bool one_part_fits = ix->storage == 2 ? 2 * k <= 1024 : 3 * k <= 3072;
    int qparts = ix->query_parts > 0 ? ix->query_parts : ((small || !one_part_fits) ? 2 : 1);
    int kc = ix->force_kc > 0 ? ix->force_kc
             : qparts == 1 ? std::max(256, ix->storage == 2 ? mult1 * k : std::min(mult1 * k, std::max(2048, 3 * k)))
             // two parts, bf16 rows only (no row rounding in the margin): 1.25k candidates; small batches 1.5k and
             // as few as 128 — a list shorter than the number of splits makes the splits' published bests a far
             // tighter bound (scan_bf16_kernel: the kc-th largest of them instead of the minimum)
             : ix->storage == 2 ? (small ? std::max(128, k + k / 2) : std::max(256, k + k / 4))
             // two parts, fp32 rows: 2k candidates, 3k for shallow lists (measured, small batches: at k = 150 a
             // list of 300 left 2 of 32 queries to the repeat pass, 450 none; from k = 250 on 2k leaves none);
             // 256 (the number of splits: every split publishes just its best score) while that is >= 2.5k
                           : std::max(256, 5 * k <= 2 * 256 ? 256 : (k <= 256 ? 3 * k : 2 * k));
    if (ix->margin_post) {
        // With the after-the-fact margin test (rescore_kernel: the list has to reach e, not 2e, below the k-th score)
        // the lists shrink to about the square root of their old multiple of k. Measured on the 8.8M x 768 Gaussian
        // workload (scripts/margin_probe.py; repeated queries of 6980, or of 20 batches of 64):
        //   fp32 rows + bf16 copy, one part : k = 100 with 256 none (scan 78.9 vs 81.3 ms with 500); k = 1000 with 1800
        //                                     none (scan 95 ms, rescoring 12.7; with 3000: 120 + 17.2)
        //   same, hi + lo, 64 queries       : k = 100 with 150: 36 of 1280, 160: 1, 180 and up: none (scan 2.33 ms
        //                                     against 2.53 with 256); k = 10 with 128: none
        //   bf16 rows only, one part        : k = 1000 with 1500 none (94 ms); 64 queries, k = 100 with 150: 1, 200: none
        //   bf16 rows only, hi + lo         : 64 queries, k = 100 with 110: 118 of 1280, 120 and up: none
        // So: one part for every large batch (2k on fp32 rows, 1.75k on bf16 rows), hi + lo for small ones (2k / 1.5k).
        qparts = ix->query_parts > 0 ? ix->query_parts : (small ? 2 : 1);
        kc = ix->force_kc > 0 ? ix->force_kc
             : qparts == 1 ? std::max(256, ix->storage == 2 ? k + 3 * k / 4 : 2 * k)
             : ix->storage == 2 ? (small ? std::max(128, k + k / 2) : std::max(256, k + k / 4))
                           : (small ? std::max(128, 2 * k) : std::max(256, k <= 256 ? 3 * k : 2 * k));
    }
    if (ix->force_kc == 0 && ix->kc_boost > 8) kc = (int)std::min<int64_t>(ipx::DEEP_MAX_K, (int64_t)kc * ix->kc_boost / 8);
    if (ix->query_parts == 0 && ix->parts_boost) qparts = 2;
    // the longest list: MAX_KC for depths one LDS sort serves, DEEP_MAX_K beyond (lists left unsorted by the scan,
    // deep_finish_kernel; large-batch kernel only)
    const int kc_max = k > IPX_MAX_K ? ipx::DEEP_MAX_K : MAX_KC;
    kc = std::min(std::max(kc, k), kc_max);

    const float* qf = dq;
    if (A > 0) {   // the fused fp32 query is needed row-major for the rescoring
        if (int rc = ix->qfused.ensure((size_t)nq * d * sizeof(float))) return rc;
        const int64_t total = nq * (int64_t)d;
        hipLaunchKernelGGL(ipx::fuse_aspect_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dq, dw, A,
                           logits, nq, d, (float*)ix->qfused.p);
        IPX_HIP(hipGetLastError());
        qf = (const float*)ix->qfused.p;
    }
    // flagged: a longer list may do; hard: no bf16 list can (they skip the repeat pass)
    std::vector<int> flagged, hard;
    if (int rc = bf16_pass(ix, qf, nq, k, kc, qparts, dD, dI, st, true, flagged, hard)) return rc;
    const int64_t nfail = (int64_t)(flagged.size() + hard.size());
    ix->last_fallback = nfail;
    // more than 0.1 % of a large batch needs a longer list: the next searches of this index start with one (a repeat pass
    // is a scan of the whole corpus for a handful of queries: at 1M rows a dozen repeated queries of 6980 cost 40 % of the
    // search, a list x1.25 a few per cent)
    if (ix->force_kc == 0 && nq >= 64 && (int64_t)flagged.size() * 1000 > nq) {
        if (kc < kc_max) ix->kc_boost = std::min(128, ix->kc_boost + (ix->kc_boost + 3) / 4);    // x1.25: 8, 10, 13, 17, 22, ...
        else if (qparts == 1 && ix->query_parts == 0) ix->parts_boost = true;
    }
    // most of a batch failed with a list that is already half the longest one (or for good): the repeat pass would
    // fail too
    const bool hopeless = ix->storage == 1 && nq >= 64 && nfail * 2 > nq && (2 * kc > kc_max || hard.size() * 2 > (size_t)nq) &&
                          ix->force_kc == 0;
    if (hopeless) ix->bf16_fail_k = std::min(ix->bf16_fail_k, k);
    if (!flagged.empty() && !hopeless && (kc < kc_max || qparts == 1)) {    // longest list, hi + lo query
        std::vector<int> still, still_hard;
        int rc = redo_queries(ix, qf, flagged, k, dD, dI, st, [&](const float* q2, int64_t n2, float* D2, int64_t* I2) {
            return bf16_pass(ix, q2, n2, k, kc_max, 2, D2, I2, st, false, still, still_hard);
        });
        if (rc) return rc;
        still.insert(still.end(), still_hard.begin(), still_hard.end());
        for (int& i : still) i = flagged[(size_t)i];
        flagged.swap(still);
        // most of the batch failed with the longest list too: this depth goes to the fp32 scan until rows are added
        if (ix->storage == 1 && nq >= 64 && (int64_t)(flagged.size() + hard.size()) * 2 > nq && ix->force_kc == 0)
            ix->bf16_fail_k = std::min(ix->bf16_fail_k, k);
    }
    flagged.insert(flagged.end(), hard.begin(), hard.end());
    std::sort(flagged.begin(), flagged.end());
    ix->last_exact = (int64_t)flagged.size();
    if (ix->storage == 1 && ix->n_outliers > 0)
        if (int rc = merge_outlier_rows(ix, qf, nq, k, dD, dI, st)) return rc;
    if (flagged.empty()) return 0;
    if (ix->storage != 1) return exact_rows_path(ix, qf, flagged, k, dD, dI, st);
    return redo_queries(ix, qf, flagged, k, dD, dI, st, [&](const float* q2, int64_t n2, float* D2, int64_t* I2) {
        return scan_f32_path(ix, q2, nullptr, 0, 0, n2, k, D2, I2, st, false);
    });
}

// Shared body of ipx_search / ipx_search_aspect. q_src is [nq, d] (A == 0) or [nq, A, d].
int search_impl(ipx_index* ix, const float* q_src, const float* w, int A, int logits, int64_t nq, int k,
                float* out_D, int64_t* out_I, int on_device, void* stream) {
    if (!ix) return fail(IPX_ERR_ARG, "search: null index");
    if (nq < 0) return fail(IPX_ERR_ARG, "search: nq < 0");
    if (k <= 0 || k > IPX_MAX_DEPTH)
        return fail(IPX_ERR_ARG, "search: k=" + std::to_string(k) + " outside [1, " + std::to_string(IPX_MAX_DEPTH) + "]");
    if (nq == 0) return 0;
    if (!q_src || !out_D || !out_I || (A > 0 && !w)) return fail(IPX_ERR_ARG, "search: null pointer");
    if (nq > (int64_t)0x7fffff00) return fail(IPX_ERR_ARG, "search: nq too large for one call");
    IPX_ON_DEVICE(ix->device);
    hipStream_t st = (hipStream_t)stream;
    const int d = ix->d;
    const size_t nout = (size_t)nq * k;

    // inputs / outputs on the device
    const float* dq = q_src;
    const float* dw = w;
    float* dD = out_D;
    int64_t* dI = out_I;
    if (!on_device) {
        const size_t qbytes = (size_t)nq * (A > 0 ? A : 1) * d * sizeof(float);
        if (int rc = ix->stageQ.ensure(qbytes)) return rc;
        IPX_HIP(hipMemcpyAsync(ix->stageQ.p, q_src, qbytes, hipMemcpyHostToDevice, st));
        dq = (const float*)ix->stageQ.p;
        if (A > 0) {
            if (int rc = ix->stageW.ensure((size_t)nq * A * sizeof(float))) return rc;
            IPX_HIP(hipMemcpyAsync(ix->stageW.p, w, (size_t)nq * A * sizeof(float), hipMemcpyHostToDevice, st));
            dw = (const float*)ix->stageW.p;
        }
        if (int rc = ix->outD.ensure(nout * sizeof(float))) return rc;
        if (int rc = ix->outI.ensure(nout * sizeof(int64_t))) return rc;
        dD = (float*)ix->outD.p;
        dI = (int64_t*)ix->outI.p;
    }

    const int64_t N = ix->ntotal;
    if (ix->timing && !ix->ev[0])
        for (auto& e : ix->ev) IPX_HIP(hipEventCreate(&e));
    ix->ev_valid = false;
    ix->ev_merge = false;
    ix->ev_rescore = false;
    ix->last_kc = 0;
    ix->last_fallback = 0;
    ix->last_exact = 0;

    if (N == 0) {
        // empty index: every slot is (-FLT_MAX, -1), as faiss returns for ntotal == 0
        hipLaunchKernelGGL(ipx::fill_empty_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, dD, dI,
                           (int64_t)nout);
        IPX_HIP(hipGetLastError());
    } else {
        // the error word exists since ipx_create and is cleared whenever it was found set: no fill per search
        ix->last_err = (int*)ix->err.p;
        // bf16 scan whenever the index keeps a bf16 copy and the depth is one its candidate lists can serve: 2k
        // candidates on fp32 rows, 1.75k on bf16 rows, at most DEEP_MAX_K entries (and, for lists beyond MAX_KC, a row
        // short enough for the rescoring kernel's LDS: d <= 2048). Deeper searches take the fp32 scan, which bf16-only
        // storage does not have. A forced candidate count ("bf16_candidates") changes the list length, never these
        // limits.
        const int bf16_max_k = ipx::DEEP_MAX_K / 2;
        const bool deep_rows_fit = (ix->storage == 2 ? ix->nks_h * 32 : ix->nks * 16) <= 2048;
        const int bf16_limit = deep_rows_fit ? bf16_max_k : IPX_MAX_K * 4 / 5;
        if (ix->storage == 2 && k > bf16_limit)
            return fail(IPX_ERR_ARG, "search: bf16 storage serves k <= " + std::to_string(bf16_limit) +
                        (deep_rows_fit ? "" : " at this row length (d > 2048)") +
                        " (candidate lists of 1.75 k entries, at most " + std::to_string(ipx::DEEP_MAX_K) +
                        "); use storage 0 or 1 for deeper searches");
        const bool bf16 = ix->storage == 2 ||
                          (ix->storage == 1 && !ix->force_f32_scan && k <= (deep_rows_fit ? bf16_max_k : IPX_MAX_K) &&
                           k < ix->bf16_fail_k);
        int rc = bf16 ? scan_bf16_path(ix, dq, dw, A, logits, nq, k, dD, dI, st)
                 : k > ipx::DEEP_MAX_K ? scan_f32_deep(ix, dq, dw, A, logits, nq, k, dD, dI, st)
                                 : scan_f32_path(ix, dq, dw, A, logits, nq, k, dD, dI, st, true);
        if (rc) return rc;
    }

    if (!on_device) {
        IPX_HIP(hipMemcpyAsync(out_D, dD, nout * sizeof(float), hipMemcpyDeviceToHost, st));
        IPX_HIP(hipMemcpyAsync(out_I, dI, nout * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        int errflag = 0;
        if (N > 0) IPX_HIP(hipMemcpyAsync(&errflag, ix->err.p, sizeof(int), hipMemcpyDeviceToHost, st));
        IPX_HIP(hipStreamSynchronize(st));
        if (errflag) {
            (void)hipMemsetAsync(ix->err.p, 0, 256, st);      // found set: cleared for the next search
            return fail(IPX_ERR_STATE, "search: candidate queue overflow (internal invariant violated)");
        }
    }
    return 0;
}

// Merge with lists or a result beyond IPX_MAX_K entries (device pointers, [S][nq][k_in] -> [nq][k_out]): the input
// lists are cut into chunks of at most IPX_MAX_K entries and merged by merge_topk_kernel, IPX_MAX_K output ranks per
// pass; pass j blanks every entry not strictly after the last entry of pass j-1 (deep_stage_kernel), as the scans of a
// deep search do. Queries go through in batches so the staged copy stays small. Synchronises the stream before it
// frees its buffers. An entry present several times with the same score and id (one document in two shards' lists) is
// returned as often as it is present, as ResultHeap does, also when a pass ends between its copies (cN / eq below).
int merge_deep(const float* dD, const int64_t* dI, int S, int64_t nq, int k_in, int k_out, float* oD, int64_t* oI, hipStream_t st) {
    const int chunk = std::min(k_in, (int)IPX_MAX_K), nchunks = (k_in + chunk - 1) / chunk, step = IPX_MAX_K;
    const int64_t k_live = std::min<int64_t>(k_out, (int64_t)S * k_in);
    if (k_live < k_out) {
        const int64_t nout = nq * (int64_t)k_out;
        hipLaunchKernelGGL(ipx::fill_empty_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, oD, oI, nout);
        IPX_HIP(hipGetLastError());
    }
    if (k_live == 0) return 0;
    const int64_t nb_max = std::min<int64_t>(nq, 512);
    const int64_t Sx = (int64_t)S * nchunks;
    DevBuf sD, sI, tD, tI, cD, cI, cN, eq;
    auto cleanup = [&]() { for (DevBuf* b : {&sD, &sI, &tD, &tI, &cD, &cI, &cN, &eq}) b->release(); };
    int rc = 0;
    if ((rc = sD.ensure((size_t)Sx * nb_max * chunk * sizeof(float))) || (rc = sI.ensure((size_t)Sx * nb_max * chunk * sizeof(int64_t))) ||
        (rc = tD.ensure((size_t)nb_max * step * sizeof(float))) || (rc = tI.ensure((size_t)nb_max * step * sizeof(int64_t))) ||
        (rc = cD.ensure((size_t)nq * sizeof(float))) || (rc = cI.ensure((size_t)nq * sizeof(int64_t))) ||
        (rc = cN.ensure((size_t)nq * sizeof(int))) || (rc = eq.ensure((size_t)nq * sizeof(int)))) {
        cleanup();
        return rc;
    }
    hipError_t e = hipSuccess;
    for (int64_t q0 = 0; q0 < nq && e == hipSuccess; q0 += nb_max) {
        const int64_t nb = std::min(nb_max, nq - q0);
        for (int64_t off = 0; off < k_live && e == hipSuccess; off += step) {
            const int kk = (int)std::min<int64_t>(step, k_live - off);
            const int64_t nst = Sx * nb * chunk;
            if (off) e = hipMemsetAsync((int*)eq.p + q0, 0, (size_t)nb * sizeof(int), st);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(ipx::deep_stage_kernel, dim3((unsigned)((nst + 255) / 256)), dim3(256), 0, st, dD, dI, S, nq, k_in,
                               chunk, nchunks, q0, nb, off ? (const float*)cD.p : nullptr, off ? (const int64_t*)cI.p : nullptr,
                               (const int*)cN.p, (int*)eq.p, (float*)sD.p, (int64_t*)sI.p);
            const int P = next_pow2(std::max(chunk, kk));
            const int mlds = P * 24;
            if (mlds > 64 * 1024)
                e = hipFuncSetAttribute((const void*)ipx::merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mlds);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nb), dim3(256), mlds, st, (const float*)sD.p,
                               (const int64_t*)sI.p, (int)Sx, nb, chunk, kk, P, 0, nb * (int64_t)chunk, (int64_t)chunk,
                               (float*)tD.p, (int64_t*)tI.p, (const int*)nullptr, (int64_t)0);
            const int64_t n = nb * (int64_t)kk;
            hipLaunchKernelGGL(ipx::deep_collect_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)tD.p,
                               (const int64_t*)tI.p, nb, kk, oD + q0 * k_out, oI + q0 * k_out, k_out, (int)off,
                               (float*)cD.p + q0, (int64_t*)cI.p + q0, (int*)cN.p + q0);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) return fail(IPX_ERR_HIP, std::string("merge (deep lists): ") + hipGetErrorString(e));
    return 0;
}

}  // namespace

extern "C" {

// used by ranking_writer.cpp to report through ipx_last_error(); not part of the public ABI
void ipx_set_error_(const char* msg) { g_err = msg ? msg : ""; }

const char* ipx_last_error(void) { return g_err.c_str(); }

const char* ipx_version(void) {
#ifdef IPX_DIAG
    return "ipx 0.1 gfx950 diag";
#else
    return "ipx 0.1 gfx950";
#endif
}

int ipx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ipx_create(int d, int device, ipx_index** out) {
    if (!out) return fail(IPX_ERR_ARG, "create: null out pointer");
    *out = nullptr;
    if (d <= 0 || d > 65536) return fail(IPX_ERR_ARG, "create: dimension " + std::to_string(d) + " outside [1, 65536]");
    int ndev = 0;
    IPX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(IPX_ERR_ARG, "create: device " + std::to_string(device) + " not in [0, " + std::to_string(ndev) + ")");
    ipx_index* ix = new (std::nothrow) ipx_index();
    if (!ix) return fail(IPX_ERR_NOMEM, "create: host allocation failed");
    ix->d = d;
    ix->nks = (d + ipx::BK - 1) / ipx::BK;
    ix->nks_h = (d + ipx::HBK - 1) / ipx::HBK;
    ix->device = device;
    int cus = 0;
    hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess || cus <= 0) {
        delete ix;
        return fail(IPX_ERR_HIP, std::string("create: multiprocessor count: ") + hipGetErrorString(e));
    }
    ix->num_cus = cus;
    {   // set-up of the no-allocation workspace: once per device, with that device current
        DeviceGuard guard(device);
        void* ws = nullptr;
        int rc = guard.err != hipSuccess
                     ? fail(IPX_ERR_HIP, std::string("create: hipSetDevice: ") + hipGetErrorString(guard.err))
                     : scratch_ptr(device, &ws);
        // the error word of this index's searches, and the launch attributes ipx_search_noalloc's kernels need: that
        // entry point itself then neither allocates nor touches the module
        if (!rc) rc = ix->err.ensure(256);
        if (!rc && hipMemset(ix->err.p, 0, 256) != hipSuccess) rc = fail(IPX_ERR_HIP, "create: hipMemset");
        if (!rc) {
            const int l1 = ipx::SCfg<1>::RING + ipx::LDS_STATE_BYTES, l2 = ipx::SCfg<2>::RING + ipx::LDS_STATE_BYTES;
            if (hipFuncSetAttribute((const void*)ipx::scan_f32_small_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, l1) != hipSuccess ||
                hipFuncSetAttribute((const void*)ipx::scan_f32_small_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, l2) != hipSuccess)
                rc = fail(IPX_ERR_HIP, "create: hipFuncSetAttribute");
        }
        if (rc) {
            ix->err.release();
            delete ix;
            return rc;
        }
    }
    ix->last_err = (int*)ix->err.p;
    *out = ix;
    return 0;
}

int ipx_destroy(ipx_index* ix) {
    if (!ix) return 0;
    DeviceGuard guard(ix->device);
    if (ix->Xp) (void)hipFree(ix->Xp);
    if (ix->Xh) (void)hipFree(ix->Xh);
    for (DevBuf* b : {&ix->Qp, &ix->cand, &ix->partD, &ix->partI, &ix->err, &ix->stageIn, &ix->stageQ, &ix->stageW,
                      &ix->outD, &ix->outI, &ix->trace, &ix->pub, &ix->maxsq, &ix->outl, &ix->olist, &ix->sideD, &ix->sideI, &ix->qfused, &ix->candD, &ix->candI, &ix->deepD, &ix->deepI, &ix->ceilD, &ix->ceilI, &ix->deepK, &ix->deepN,
                      &ix->flags, &ix->sel, &ix->qsel, &ix->selD, &ix->selI, &ix->poolH, &ix->poolW, &ix->poolQ})
        b->release();
    for (auto& e : ix->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : ix->ev_stage)
        if (e) (void)hipEventDestroy(e);
    if (ix->pinned) (void)hipHostFree(ix->pinned);
    if (ix->hflags) (void)hipHostFree(ix->hflags);
    delete ix;
    return 0;
}

int ipx_dim(const ipx_index* ix) { return ix ? ix->d : 0; }
int64_t ipx_ntotal(const ipx_index* ix) { return ix ? ix->ntotal : 0; }

int ipx_reserve(ipx_index* ix, int64_t n) {
    if (!ix || n < 0) return fail(IPX_ERR_ARG, "reserve: bad argument");
    IPX_ON_DEVICE(ix->device);
    return grow_storage(ix, n, nullptr);
}

int ipx_set_id_base(ipx_index* ix, int64_t id_base) {
    if (!ix) return fail(IPX_ERR_ARG, "set_id_base: null index");
    ix->id_base = id_base;
    return 0;
}

int ipx_add(ipx_index* ix, const void* rows, int64_t n, int dtype, int on_device, void* stream) {
    if (!ix) return fail(IPX_ERR_ARG, "add: null index");
    if (n < 0) return fail(IPX_ERR_ARG, "add: n < 0");
    if (dtype != IPX_F32 && dtype != IPX_F16 && dtype != IPX_BF16) return fail(IPX_ERR_ARG, "add: unknown dtype");
    if (n == 0) return 0;
    if (!rows) return fail(IPX_ERR_ARG, "add: null rows");
    if (ix->ntotal + n > (int64_t)0x7fffff00) return fail(IPX_ERR_ARG, "add: more than 2^31 rows in one index");
    IPX_ON_DEVICE(ix->device);
    hipStream_t st = (hipStream_t)stream;
    ix->bf16_fail_k = INT32_MAX;          // new rows, new norm bounds
    ix->kc_boost = 8;
    ix->parts_boost = false;
    if (int rc = grow_storage(ix, ix->ntotal + n, st)) return rc;
    const size_t row_bytes = (size_t)ix->d * dtype_size(dtype);
    if (on_device) {
        // pack in slices so one launch's grid stays within 2^31 blocks
        const int64_t slice = std::max<int64_t>(1, (int64_t)(1ull << 30) / ((int64_t)ix->nks * 4));
        for (int64_t r0 = 0; r0 < n; r0 += slice) {
            const int64_t m = std::min(slice, n - r0);
            if (int rc = add_device_rows(ix, (const char*)rows + r0 * row_bytes, dtype, m, ix->ntotal + r0, st)) return rc;
        }
    } else {
        // host rows (pageable memory, e.g. an unpickled ndarray): two pinned 64-MiB buffers and two
        // device staging buffers; while slice i travels over PCIe and is packed, host threads copy
        // slice i+1 into the other pinned buffer. The caller's array is read once, never pinned.
        const int64_t slice = std::max<int64_t>(1, (int64_t)(64ull << 20) / (int64_t)row_bytes);
        const size_t slice_bytes = (size_t)std::min(slice, n) * row_bytes;
        if (int rc = ix->stageIn.ensure(2 * slice_bytes)) return rc;
        if (!ix->pinned || ix->pinned_bytes < 2 * slice_bytes) {
            if (ix->pinned) (void)hipHostFree(ix->pinned);
            ix->pinned = nullptr;
            ix->pinned_bytes = 0;
            hipError_t e = hipHostMalloc(&ix->pinned, 2 * slice_bytes, hipHostMallocDefault);
            if (e != hipSuccess) { ix->pinned = nullptr; return fail(IPX_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
            ix->pinned_bytes = 2 * slice_bytes;
        }
        for (auto& e : ix->ev_stage)
            if (!e) IPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const int nthreads = (int)std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
        int64_t i = 0;
        for (int64_t r0 = 0; r0 < n; r0 += slice, ++i) {
            const int64_t m = std::min(slice, n - r0);
            const int b = (int)(i & 1);
            char* pin = (char*)ix->pinned + (size_t)b * slice_bytes;
            char* dev = (char*)ix->stageIn.p + (size_t)b * slice_bytes;
            if (i >= 2) IPX_HIP(hipEventSynchronize(ix->ev_stage[b]));   // the copy out of this pinned buffer is done
            const char* src = (const char*)rows + r0 * row_bytes;
            const size_t bytes = (size_t)m * row_bytes;
            if (nthreads == 1 || bytes < (8u << 20)) {
                memcpy(pin, src, bytes);
            } else {
                std::vector<std::thread> th;
                const size_t per = (bytes / nthreads + 4095) & ~(size_t)4095;
                for (int t = 0; t < nthreads; ++t) {
                    const size_t lo = std::min(bytes, (size_t)t * per), hi = std::min(bytes, lo + per);
                    if (hi > lo) th.emplace_back([=] { memcpy(pin + lo, src + lo, hi - lo); });
                }
                for (auto& t : th) t.join();
            }
            IPX_HIP(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st));
            IPX_HIP(hipEventRecord(ix->ev_stage[b], st));
            if (int rc = add_device_rows(ix, dev, dtype, m, ix->ntotal + r0, st)) return rc;
        }
        IPX_HIP(hipStreamSynchronize(st));   // host-pointer calls are synchronous
    }
    ix->ntotal += n;
    return 0;
}

int ipx_search(ipx_index* ix, const float* q, int64_t nq, int k, float* out_D, int64_t* out_I, int on_device,
               void* stream) {
    return search_impl(ix, q, nullptr, 0, 0, nq, k, out_D, out_I, on_device, stream);
}

// The one entry point that allocates nothing: ipx_search on device pointers, small batches, fp32 rows, with every
// intermediate buffer (packed queries, candidate queues, per-split lists, counters, the error word) carved from the
// static workspace `scratch`. No hipMalloc / hipFree / event creation; stream-ordered, returns without synchronising.
int ipx_search_noalloc(ipx_index* ix, const float* q, int64_t nq, int k, float* out_D, int64_t* out_I, void* stream) {
    if (!ix) return fail(IPX_ERR_ARG, "search_noalloc: null index");
    if (nq < 0 || nq > IPX_NOALLOC_MAX_QUERIES)
        return fail(IPX_ERR_ARG, "search_noalloc: nq=" + std::to_string(nq) + " outside [0, " +
                    std::to_string(IPX_NOALLOC_MAX_QUERIES) + "] (use ipx_search for larger batches)");
    if (k <= 0 || k > IPX_MAX_K)
        return fail(IPX_ERR_ARG, "search_noalloc: k=" + std::to_string(k) + " outside [1, " + std::to_string(IPX_MAX_K) + "]");
    if (ix->storage == 2) return fail(IPX_ERR_STATE, "search_noalloc: needs fp32 rows (storage 0 or 1)");
    if (nq == 0) return 0;
    if (!q || !out_D || !out_I) return fail(IPX_ERR_ARG, "search_noalloc: null pointer");
    DeviceGuard guard(ix->device);
    if (guard.err != hipSuccess) return fail(IPX_ERR_HIP, std::string("search_noalloc: hipSetDevice: ") + hipGetErrorString(guard.err));
    hipStream_t st = (hipStream_t)stream;
    ix->ev_valid = ix->ev_merge = ix->ev_rescore = false;
    ix->last_kc = 0;
    ix->last_fallback = 0;
    ix->last_exact = 0;
    if (ix->ntotal == 0) {
        const int64_t nout = nq * k;
        hipLaunchKernelGGL(ipx::fill_empty_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, out_D, out_I, nout);
        IPX_HIP(hipGetLastError());
        ix->last_err = nullptr;
        return 0;
    }
    void* ws = nullptr;
    if (int rc = scratch_ptr(ix->device, &ws)) return rc;
    // as many corpus splits as the workspace holds (each split brings its candidate queues and its partial lists)
    SmallGeom g = small_geometry(ix, nq, k, ix->max_splits);
    while (g.total() > (size_t)IPX_NOALLOC_SCRATCH_BYTES && g.S > 1) {
        const int fit = (int)((double)g.S * IPX_NOALLOC_SCRATCH_BYTES / (double)g.total());
        g = small_geometry(ix, nq, k, std::max(1, std::min(g.S - 1, fit)));
    }
    if (g.total() > (size_t)IPX_NOALLOC_SCRATCH_BYTES)
        return fail(IPX_ERR_ARG, "search_noalloc: nq=" + std::to_string(nq) + ", k=" + std::to_string(k) + ", d=" +
                    std::to_string(ix->d) + " need " + std::to_string(g.total()) + " bytes of workspace, more than the " +
                    std::to_string((size_t)IPX_NOALLOC_SCRATCH_BYTES) + " this entry point owns (use ipx_search)");
    char* at = (char*)ws;
    auto carve = [&](size_t bytes) { char* r = at; at += (bytes + 255) / 256 * 256; return (void*)r; };
    SmallBufs b;
    b.err = (int*)ix->err.p;          // per index (allocated by ipx_create): ipx_check(idx) reports this index's searches
    b.Qp = carve(g.qp_bytes);
    b.cand = carve(g.cand_bytes);
    b.partD = carve(g.partD_bytes);
    b.partI = carve(g.partI_bytes);
    b.flags = carve(g.flags_bytes);
    b.pub = carve(g.pub_bytes);
    // the workspace keeps the previous call's contents: what must start at zero is cleared in the launch (run counters
    // and published bounds); queues, lists and flags are written before they are read
    ix->last_err = b.err;
    const bool record = ix->timing && ix->ev[0] != nullptr;     // events are never created here
    return scan_f32_small_launch(ix, g, b, q, nullptr, 0, 0, nq, k, out_D, out_I, st, record, true, /*set_attr=*/false);
}

int ipx_search_aspect(ipx_index* ix, const float* aspect_emb, const float* w, int A, int w_is_logits, int64_t nq,
                      int k, float* out_D, int64_t* out_I, int on_device, void* stream) {
    if (A <= 0 || A > 64) return fail(IPX_ERR_ARG, "search_aspect: A outside [1, 64]");
    return search_impl(ix, aspect_emb, w, A, w_is_logits ? 1 : 0, nq, k, out_D, out_I, on_device, stream);
}

// MURAL 'att' pooling of n sequences into ix->poolQ (fp32 [n, d]); attention weights behind W and b in ix->poolW.
static int pool_impl(ipx_index* ix, const char* who, const float* hidden, int64_t seq_len, const float* W, const float* b,
                     int A, int begin_id, int64_t n, float* out_att, int on_device, hipStream_t st) {
    if (!ix) return fail(IPX_ERR_ARG, std::string(who) + ": null index");
    if (A <= 0 || A > 64) return fail(IPX_ERR_ARG, std::string(who) + ": A outside [1, 64]");
    if (begin_id < 0 || begin_id + (int64_t)A > seq_len)
        return fail(IPX_ERR_ARG, std::string(who) + ": aspect tokens outside the sequence");
    if (n < 0 || n > 0x7fffffff) return fail(IPX_ERR_ARG, std::string(who) + ": n outside [0, 2^31)");
    if (n == 0) return 0;
    if (!hidden || !W || !b) return fail(IPX_ERR_ARG, std::string(who) + ": null pointer");
    IPX_ON_DEVICE(ix->device);
    const int d = ix->d;
    const hipMemcpyKind in = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // keep [CLS] and the A aspect rows of every sequence: [n][1 + A][d]
    const size_t row = (size_t)d * sizeof(float), kept = (size_t)(1 + A) * row, pitch = (size_t)seq_len * row;
    if (int rc = ix->poolH.ensure((size_t)n * kept)) return rc;
    if (int rc = ix->poolW.ensure((size_t)A * row + (size_t)A * sizeof(float) + (size_t)n * A * sizeof(float))) return rc;
    if (int rc = ix->poolQ.ensure((size_t)n * row)) return rc;
    IPX_HIP(hipMemcpy2DAsync(ix->poolH.p, kept, hidden, pitch, row, (size_t)n, in, st));
    IPX_HIP(hipMemcpy2DAsync((char*)ix->poolH.p + row, kept, (const char*)hidden + (size_t)begin_id * row, pitch,
                             (size_t)A * row, (size_t)n, in, st));
    float* dW = (float*)ix->poolW.p;
    float* db = dW + (size_t)A * d;
    float* datt = db + A;
    IPX_HIP(hipMemcpyAsync(dW, W, (size_t)A * row, in, st));
    IPX_HIP(hipMemcpyAsync(db, b, (size_t)A * sizeof(float), in, st));
    hipLaunchKernelGGL(ipx::mural_pool_kernel, dim3((unsigned)n), dim3(256), 0, st, (const float*)ix->poolH.p, dW, db, A, n,
                       d, (float*)ix->poolQ.p, datt);
    IPX_HIP(hipGetLastError());
    if (out_att)
        IPX_HIP(hipMemcpyAsync(out_att, datt, (size_t)n * A * sizeof(float),
                               on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    return 0;
}

int ipx_pool(ipx_index* ix, const float* hidden, int64_t seq_len, const float* W, const float* b, int A, int begin_id,
             int64_t n, float* out_rep, float* out_att, int on_device, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n > 0 && !out_rep) return fail(IPX_ERR_ARG, "pool: null pointer");
    if (int rc = pool_impl(ix, "pool", hidden, seq_len, W, b, A, begin_id, n, out_att, on_device, st)) return rc;
    if (n == 0) return 0;
    IPX_HIP(hipMemcpyAsync(out_rep, ix->poolQ.p, (size_t)n * ix->d * sizeof(float),
                           on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    if (!on_device) IPX_HIP(hipStreamSynchronize(st));
    return 0;
}

int ipx_add_pooled(ipx_index* ix, const float* hidden, int64_t seq_len, const float* W, const float* b, int A,
                   int begin_id, int64_t n, int on_device, void* stream) {
    if (int rc = pool_impl(ix, "add_pooled", hidden, seq_len, W, b, A, begin_id, n, nullptr, on_device, (hipStream_t)stream))
        return rc;
    if (n == 0) return 0;
    if (int rc = ipx_add(ix, ix->poolQ.p, n, IPX_F32, 1, stream)) return rc;
    if (!on_device) IPX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int ipx_search_pooled(ipx_index* ix, const float* hidden, int64_t seq_len, const float* W, const float* b, int A,
                      int begin_id, int64_t nq, int k, float* out_D, int64_t* out_I, float* out_att, int on_device,
                      void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (k <= 0 || k > IPX_MAX_DEPTH) return fail(IPX_ERR_ARG, "search_pooled: k outside [1, IPX_MAX_DEPTH]");
    if (nq > 0 && (!out_D || !out_I)) return fail(IPX_ERR_ARG, "search_pooled: null pointer");
    if (int rc = pool_impl(ix, "search_pooled", hidden, seq_len, W, b, A, begin_id, nq, out_att, on_device, st)) return rc;
    if (nq == 0) return 0;
    if (on_device) return search_impl(ix, (const float*)ix->poolQ.p, nullptr, 0, 0, nq, k, out_D, out_I, 1, stream);
    // host outputs: search on the pooled queries (already on the device), then copy the lists back
    const size_t nout = (size_t)nq * k;
    if (int rc = ix->outD.ensure(nout * sizeof(float))) return rc;
    if (int rc = ix->outI.ensure(nout * sizeof(int64_t))) return rc;
    if (int rc = search_impl(ix, (const float*)ix->poolQ.p, nullptr, 0, 0, nq, k, (float*)ix->outD.p, (int64_t*)ix->outI.p, 1,
                             stream))
        return rc;
    IPX_HIP(hipMemcpyAsync(out_D, ix->outD.p, nout * sizeof(float), hipMemcpyDeviceToHost, st));
    IPX_HIP(hipMemcpyAsync(out_I, ix->outI.p, nout * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    return ipx_check(ix, stream);
}

int ipx_merge_topk(const float* D, const int64_t* I, int S, int64_t nq, int k_in, int k_out, float* out_D,
                   int64_t* out_I, int on_device, int device, void* stream) {
    if (S < 0 || nq < 0 || k_in <= 0 || k_out <= 0) return fail(IPX_ERR_ARG, "merge: bad sizes");
    if (k_in > IPX_MAX_DEPTH || k_out > IPX_MAX_DEPTH)
        return fail(IPX_ERR_ARG, "merge: k outside [1, " + std::to_string(IPX_MAX_DEPTH) + "]");
    if (nq == 0) return 0;
    if ((S > 0 && (!D || !I)) || !out_D || !out_I) return fail(IPX_ERR_ARG, "merge: null pointer");
    if (nq > 0x7fffffffLL) return fail(IPX_ERR_ARG, "merge: nq too large");
    IPX_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const size_t nin = (size_t)S * nq * k_in, nout = (size_t)nq * k_out;
    const float* dD = D;
    const int64_t* dI = I;
    float* oD = out_D;
    int64_t* oI = out_I;
    DevBuf bD, bI, boD, boI;
    auto cleanup = [&]() { bD.release(); bI.release(); boD.release(); boI.release(); };
    if (!on_device) {
        int rc = 0;
        if (nin) {
            if ((rc = bD.ensure(nin * sizeof(float))) || (rc = bI.ensure(nin * sizeof(int64_t)))) { cleanup(); return rc; }
        }
        if ((rc = boD.ensure(nout * sizeof(float))) || (rc = boI.ensure(nout * sizeof(int64_t)))) { cleanup(); return rc; }
        if (nin) {
            hipError_t e = hipMemcpyAsync(bD.p, D, nin * sizeof(float), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(bI.p, I, nin * sizeof(int64_t), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) { cleanup(); return fail(IPX_ERR_HIP, std::string("merge H2D: ") + hipGetErrorString(e)); }
        }
        dD = (const float*)bD.p; dI = (const int64_t*)bI.p; oD = (float*)boD.p; oI = (int64_t*)boI.p;
    }
    if (k_in > IPX_MAX_K || k_out > IPX_MAX_K) {      // lists beyond what one workgroup sorts in LDS
        int rc = merge_deep(dD, dI, S, nq, k_in, k_out, oD, oI, st);
        hipError_t e = hipSuccess;
        if (!rc && !on_device) {
            e = hipMemcpyAsync(out_D, oD, nout * sizeof(float), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(out_I, oI, nout * sizeof(int64_t), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (!on_device) cleanup();
        if (rc) return rc;
        if (e != hipSuccess) return fail(IPX_ERR_HIP, std::string("merge: ") + hipGetErrorString(e));
        return 0;
    }
    const int P = next_pow2(std::max(k_in, k_out));
    const int mlds = P * 24;
    hipError_t e = hipSuccess;
    if (mlds > 64 * 1024)
        e = hipFuncSetAttribute((const void*)ipx::merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mlds);
    if (e == hipSuccess) {
        // inputs are [S][nq][k_in]: shard stride nq*k_in, query stride k_in; order not assumed
        hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nq), dim3(256), mlds, st, dD, dI, S, nq, k_in, k_out,
                           P, 0, (int64_t)nq * k_in, (int64_t)k_in, oD, oI, (const int*)nullptr, (int64_t)0);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !on_device) {
        e = hipMemcpyAsync(out_D, oD, nout * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(out_I, oI, nout * sizeof(int64_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (!on_device) cleanup();
    if (e != hipSuccess) return fail(IPX_ERR_HIP, std::string("merge: ") + hipGetErrorString(e));
    return 0;
}

size_t ipx_packed_block_bytes(int64_t nq, int k) {
    if (nq < 0 || k <= 0) return 0;
    return ((size_t)nq * (size_t)k * 12 + 15) / 16 * 16;
}

int ipx_merge_topk_packed(const void* packed, int S, int64_t nq, int k_in, int k_out, float* out_D, int64_t* out_I,
                          int device, void* stream) {
    if (S < 0 || nq < 0 || k_in <= 0 || k_out <= 0) return fail(IPX_ERR_ARG, "merge_packed: bad sizes");
    if (k_in > IPX_MAX_K || k_out > IPX_MAX_K)
        return fail(IPX_ERR_ARG, "merge_packed: k outside [1, " + std::to_string(IPX_MAX_K) + "] (deeper lists: ipx_merge_topk)");
    if (nq == 0) return 0;
    if ((S > 0 && !packed) || !out_D || !out_I) return fail(IPX_ERR_ARG, "merge_packed: null pointer");
    if (nq > 0x7fffffffLL) return fail(IPX_ERR_ARG, "merge_packed: nq too large");
    IPX_ON_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const size_t block = ipx_packed_block_bytes(nq, k_in);
    const int64_t* I = (const int64_t*)packed;
    const float* D = (const float*)((const char*)packed + (size_t)nq * k_in * sizeof(int64_t));
    const int P = next_pow2(std::max(k_in, k_out));
    const int mlds = P * 24;
    if (mlds > 64 * 1024)
        IPX_HIP(hipFuncSetAttribute((const void*)ipx::merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mlds));
    hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nq), dim3(256), mlds, st, D, I, S, nq, k_in, k_out, P, 0,
                       (int64_t)(block / sizeof(float)), (int64_t)k_in, out_D, out_I, (const int*)nullptr,
                       (int64_t)(block / sizeof(int64_t)));
    IPX_HIP(hipGetLastError());
    return 0;
}

int ipx_set_param(ipx_index* ix, const char* name, int64_t value) {
    if (!ix || !name) return fail(IPX_ERR_ARG, "set_param: null argument");
    ix->bf16_fail_k = INT32_MAX;
    ix->kc_boost = 8;
    ix->parts_boost = false;
    if (!strcmp(name, "max_splits")) {
        if (value < 0 || value > 65536) return fail(IPX_ERR_ARG, "set_param: max_splits outside [0, 65536]");
        ix->max_splits = (int)value;
        return 0;
    }
    if (!strcmp(name, "storage")) {
        if (value < 0 || value > 2) return fail(IPX_ERR_ARG, "set_param: storage must be 0 (f32), 1 (f32 + bf16 scan) or 2 (bf16)");
        if (ix->ntotal > 0 || ix->cap_tiles > 0 || ix->cap_tiles_h > 0)
            return fail(IPX_ERR_STATE, "set_param: storage can only be chosen while the index is empty and unreserved");
        if (value != 0 && ix->d > 8192) return fail(IPX_ERR_ARG, "set_param: bf16 storage needs d <= 8192");
        ix->storage = (int)value;
        return 0;
    }
    if (!strcmp(name, "f32_helpers")) {
        if (value < 0 || value > 64) return fail(IPX_ERR_ARG, "set_param: f32_helpers outside [0, 64]");
        ix->f32_helpers = (int)value;
        return 0;
    }
    if (!strcmp(name, "run_tiles")) {
        if (value < 0 || value > 4096) return fail(IPX_ERR_ARG, "set_param: run_tiles outside [0, 4096]");
        ix->run_tiles = (int)value;
        return 0;
    }
#ifndef IPX_DIAG
    {   // kernel variants that only a diagnostic build carries: this build accepts their default values
        static const struct { const char* name; int64_t dflt, alt; } diag_only[] = {
            {"bf16_shape16", 1, 1}, {"bf16_pp", 0, 0}, {"bf16_map", 0, 0}, {"bf16_ring", 0, 2}, {"bf16_occ", 1, 1},
            {"bf16_variant", 0, 0}, {"bf16_waves", 0, 16}};
        for (const auto& e : diag_only)
            if (!strcmp(name, e.name)) {
                if (value == e.dflt || value == e.alt) return 0;
                return fail(IPX_ERR_ARG, std::string("set_param: ") + name + " selects a kernel variant that only the diagnostic "
                            "build has (libipx_diag.so, -DIPX_DIAG: __graft_entry__.build_diag())");
            }
    }
#endif
    if (!strcmp(name, "bf16_shape16")) {
        ix->bf16_shape16 = value != 0;
        return 0;
    }
    if (!strcmp(name, "bf16_pp")) {
        if (value < 0 || value > 4) return fail(IPX_ERR_ARG, "set_param: bf16_pp outside [0, 4]");
        ix->bf16_pp = (int)value;
        return 0;
    }
    if (!strcmp(name, "bf16_map")) {
        if (value < 0 || value > 4096) return fail(IPX_ERR_ARG, "set_param: bf16_map outside [0, 4096]");
        ix->bf16_map = (int)value;
        return 0;
    }
    if (!strcmp(name, "bf16_ring")) {
        if (value != 0 && (value < 2 || value > 4)) return fail(IPX_ERR_ARG, "set_param: bf16_ring is 0 (default), 2, 3 or 4");
        ix->bf16_ring = value == 0 ? 2 : (int)value;
        return 0;
    }
    if (!strcmp(name, "bf16_occ")) {
        if (value < 1 || value > 2) return fail(IPX_ERR_ARG, "set_param: bf16_occ is 1 or 2");
        ix->bf16_occ = (int)value;
        return 0;
    }
    if (!strcmp(name, "bf16_variant")) {
        if (value < 0 || value > 7) return fail(IPX_ERR_ARG, "set_param: bf16_variant outside [0, 7]");
        ix->bf16_variant = (int)value;
        return 0;
    }
    if (!strcmp(name, "bf16_waves")) {
        if (value != 0 && value != 8 && value != 12 && value != 16) return fail(IPX_ERR_ARG, "set_param: bf16_waves must be 0, 8, 12 or 16");
        ix->bf16_waves = (int)value;
        return 0;
    }
    if (!strcmp(name, "small_f32_limit")) {
        ix->small_f32_limit = (int)value;
        return 0;
    }
    if (!strcmp(name, "margin_post")) {
        ix->margin_post = value != 0;
        return 0;
    }
    if (!strcmp(name, "no_small_batch")) {
        ix->no_small_batch = value != 0;
        return 0;
    }
    if (!strcmp(name, "query_parts")) {
        if (value < 0 || value > 2) return fail(IPX_ERR_ARG, "set_param: query_parts must be 0 (automatic), 1 or 2");
        ix->query_parts = (int)value;
        return 0;
    }
    if (!strcmp(name, "f32_scan")) {
        if (value && ix->storage == 2) return fail(IPX_ERR_STATE, "set_param: f32_scan needs fp32 rows (storage 0 or 1)");
        ix->force_f32_scan = value != 0;
        return 0;
    }
    if (!strcmp(name, "bf16_candidates")) {
        if (value < 0 || value > ipx::DEEP_MAX_K)
            return fail(IPX_ERR_ARG, "set_param: bf16_candidates outside [0, " + std::to_string(ipx::DEEP_MAX_K) + "]");
        ix->force_kc = (int)value;
        return 0;
    }
    if (!strcmp(name, "trace")) {
        ix->trace_on = value != 0;
        return 0;
    }
    return fail(IPX_ERR_ARG, std::string("set_param: unknown parameter ") + name);
}

int ipx_read_trace(ipx_index* ix, uint64_t* out, int max_workgroups) {
    if (!ix || !out) return fail(IPX_ERR_ARG, "read_trace: null argument");
    if (!ix->trace_on || !ix->trace.p) return fail(IPX_ERR_STATE, "read_trace: tracing is off (set_param trace 1)");
    IPX_ON_DEVICE(ix->device);
    const int n = std::min(max_workgroups, ix->last_wgs);
    IPX_HIP(hipDeviceSynchronize());
    IPX_HIP(hipMemcpy(out, ix->trace.p, (size_t)n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (ix->last_kc > 0 && max_workgroups >= 2 * ix->last_wgs)   // bf16 scan: episode split follows
        IPX_HIP(hipMemcpy(out + (size_t)n * 4, (uint64_t*)ix->trace.p + (size_t)ix->last_wgs * 4,
                          (size_t)n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int ipx_check(ipx_index* ix, void* stream) {
    if (!ix) return fail(IPX_ERR_ARG, "check: null index");
    IPX_ON_DEVICE(ix->device);
    hipStream_t st = (hipStream_t)stream;
    int errflag = 0;
    if (ix->last_err) IPX_HIP(hipMemcpyAsync(&errflag, ix->last_err, sizeof(int), hipMemcpyDeviceToHost, st));
    IPX_HIP(hipStreamSynchronize(st));
    if (errflag) {
        (void)hipMemsetAsync(ix->last_err, 0, sizeof(int), st);
        return fail(IPX_ERR_STATE, "search: candidate queue overflow (internal invariant violated)");
    }
    return 0;
}

int ipx_set_timing(ipx_index* ix, int enabled) {
    if (!ix) return fail(IPX_ERR_ARG, "set_timing: null index");
    ix->timing = enabled != 0;
    return 0;
}

int ipx_last_timing(ipx_index* ix, float* pack_ms, float* scan_ms, float* merge_ms) {
    if (!ix) return fail(IPX_ERR_ARG, "last_timing: null index");
    if (!ix->ev_valid) return fail(IPX_ERR_STATE, "last_timing: no timed search recorded (call ipx_set_timing first)");
    IPX_ON_DEVICE(ix->device);
    IPX_HIP(hipEventSynchronize(ix->ev[ix->ev_rescore ? 4 : ix->ev_merge ? 3 : 2]));
    float a = 0.f, b = 0.f, c = 0.f;
    IPX_HIP(hipEventElapsedTime(&a, ix->ev[0], ix->ev[1]));
    IPX_HIP(hipEventElapsedTime(&b, ix->ev[1], ix->ev[2]));
    if (ix->ev_merge) IPX_HIP(hipEventElapsedTime(&c, ix->ev[2], ix->ev[3]));
    if (pack_ms) *pack_ms = a;
    if (scan_ms) *scan_ms = b;
    if (merge_ms) *merge_ms = c;
    return 0;
}

int ipx_last_bf16_stats(ipx_index* ix, float* rescore_ms, int* candidates, int64_t* fallback_queries) {
    if (!ix) return fail(IPX_ERR_ARG, "last_bf16_stats: null index");
    float r = 0.f;
    if (rescore_ms && ix->ev_valid && ix->ev_rescore) {
        IPX_ON_DEVICE(ix->device);
        IPX_HIP(hipEventSynchronize(ix->ev[4]));
        IPX_HIP(hipEventElapsedTime(&r, ix->ev[3], ix->ev[4]));
    }
    if (rescore_ms) *rescore_ms = r;
    if (candidates) *candidates = ix->last_kc;
    if (fallback_queries) *fallback_queries = ix->last_fallback;
    return 0;
}

int ipx_outlier_rows(const ipx_index* ix) { return ix ? ix->n_outliers : 0; }

int64_t ipx_last_exact_queries(const ipx_index* ix) { return ix ? ix->last_exact : 0; }

int ipx_last_geometry(const ipx_index* ix, int* workgroups, int* splits, int* queue_cap) {
    if (!ix) return fail(IPX_ERR_ARG, "last_geometry: null index");
    if (workgroups) *workgroups = ix->last_wgs;
    if (splits) *splits = ix->last_S;
    if (queue_cap) *queue_cap = ix->last_cap;
    return 0;
}

}  // extern "C"
