// ipx.hip — host side of the C ABI declared in include/ipx.h.
//
// Owns the device memory of a flat inner-product index (packed layout, see ipx_kernels.h)
// and launches the gfx950 kernels. Every entry point cites the reference call it replaces
// in include/ipx.h. No CPU fallback: without a HIP device every compute call fails with
// IPX_ERR_HIP.
#include "ipx_kernels.h"
#include "../../include/ipx.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

namespace {

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

int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace

struct ipx_index {
    int d = 0, nks = 0, device = 0;
    int64_t ntotal = 0;
    int64_t cap_tiles = 0;   // tiles of 128 rows the corpus buffer holds
    int64_t id_base = 0;
    float* Xp = nullptr;     // packed corpus
    DevBuf Qp, cand, partD, partI, err, stageIn, stageQ, stageW, outD, outI, trace, pub;
    bool trace_on = false;
    bool timing = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false, ev_merge = false;
    int last_wgs = 0, last_S = 0, last_cap = 0;
    int max_splits = 0;      // 0 = automatic
    size_t tile_bytes() const { return (size_t)nks * ipx::STAGE_A_BYTES; }
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
    if (old_bytes) IPX_HIP(hipMemcpyAsync(nb, ix->Xp, old_bytes, hipMemcpyDeviceToDevice, st));
    // rows of a partly filled tile must read as zeros
    IPX_HIP(hipMemsetAsync((char*)nb + old_bytes, 0, new_bytes - old_bytes, st));
    if (ix->Xp) {
        IPX_HIP(hipStreamSynchronize(st));
        (void)hipFree(ix->Xp);
    }
    ix->Xp = nb;
    ix->cap_tiles = new_tiles;
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

int dtype_size(int dtype) { return dtype == IPX_F32 ? 4 : 2; }

// Shared body of ipx_search / ipx_search_aspect. q_src is [nq, d] (A == 0) or [nq, A, d].
int search_impl(ipx_index* ix, const float* q_src, const float* w, int A, int logits, int64_t nq, int k,
                float* out_D, int64_t* out_I, int on_device, void* stream) {
    if (!ix) return fail(IPX_ERR_ARG, "search: null index");
    if (nq < 0) return fail(IPX_ERR_ARG, "search: nq < 0");
    if (k <= 0 || k > IPX_MAX_K)
        return fail(IPX_ERR_ARG, "search: k=" + std::to_string(k) + " outside [1, " + std::to_string(IPX_MAX_K) + "]");
    if (nq == 0) return 0;
    if (!q_src || !out_D || !out_I || (A > 0 && !w)) return fail(IPX_ERR_ARG, "search: null pointer");
    if (nq > (int64_t)0x7fffff00) return fail(IPX_ERR_ARG, "search: nq too large for one call");
    IPX_HIP(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    const int d = ix->d, nks = ix->nks;
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

    if (N == 0) {
        // empty index: every slot is (-FLT_MAX, -1), as faiss returns for ntotal == 0
        hipLaunchKernelGGL(ipx::fill_empty_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, dD, dI,
                           (int64_t)nout);
        IPX_HIP(hipGetLastError());
    } else {
        const int nqb = (int)((nq + ipx::BQ - 1) / ipx::BQ);
        const int ntiles = (int)((N + ipx::BD - 1) / ipx::BD);
        // geometry: (up to) all workgroups resident at once, 2 per CU; S of them share a query block.
        hipDeviceProp_t prop;
        IPX_HIP(hipGetDeviceProperties(&prop, ix->device));
        const int resident = 2 * prop.multiProcessorCount;
        // A query block's S workgroups take runs of R corpus tiles from a shared counter.
        int S = std::max(1, resident / nqb);
        if (ix->max_splits > 0) S = std::min(S, ix->max_splits);
        const int R = std::max(4, std::min(64, ntiles / (S * 64)));
        const int nruns = (ntiles + R - 1) / R;
        S = std::min(S, nruns);
        const int G = nqb * S;
        const int cap = std::max(512, next_pow2(2 * k));

        if (int rc = ix->Qp.ensure((size_t)nqb * nks * ipx::STAGE_B_BYTES)) return rc;
        if (int rc = ix->cand.ensure((size_t)G * ipx::BQ * cap * sizeof(uint64_t))) return rc;
        if (int rc = ix->err.ensure(256)) return rc;
        if (S > 1) {
            if (int rc = ix->partD.ensure(nout * S * sizeof(float))) return rc;
            if (int rc = ix->partI.ensure(nout * S * sizeof(int64_t))) return rc;
        }
        IPX_HIP(hipMemsetAsync(ix->err.p, 0, 256, st));
        // per search: run counters [nqb] (padded to 256 B) then published thresholds [G][256], all zero
        const size_t ctr_bytes = ((size_t)nqb * sizeof(uint32_t) + 255) / 256 * 256;
        const size_t pub_bytes = ctr_bytes + (size_t)G * ipx::BQ * sizeof(uint32_t);
        if (int rc = ix->pub.ensure(pub_bytes)) return rc;
        IPX_HIP(hipMemsetAsync(ix->pub.p, 0, pub_bytes, st));

        if (ix->timing) IPX_HIP(hipEventRecord(ix->ev[0], st));
        if (int rc = launch_pack(dq, IPX_F32, nq, (int64_t)nqb * ipx::BQ, d, (float*)ix->Qp.p, 0, ipx::BQ, nks, dw, A,
                                 logits, st))
            return rc;
        if (ix->timing) IPX_HIP(hipEventRecord(ix->ev[1], st));

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
        p.ntiles = ntiles;
        const int lds = std::max(2 * ipx::STAGE_BYTES, 4 * cap * 8) + ipx::LDS_STATE_BYTES;
        if (lds > 64 * 1024)
            IPX_HIP(hipFuncSetAttribute((const void*)ipx::scan_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(ipx::scan_f32_kernel, dim3(G), dim3(ipx::NTHREADS), lds, st, p);
        IPX_HIP(hipGetLastError());
        if (ix->timing) IPX_HIP(hipEventRecord(ix->ev[2], st));

        if (S > 1) {
            const int P = next_pow2(k);
            const int mlds = P * 24;
            if (mlds > 64 * 1024)
                IPX_HIP(hipFuncSetAttribute((const void*)ipx::merge_topk_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, mlds));
            // partials are [nq][S][k]: shard stride k, query stride S*k; lists arrive sorted
            hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nq), dim3(256), mlds, st,
                               (const float*)ix->partD.p, (const int64_t*)ix->partI.p, S, nq, k, k, P, 1,
                               (int64_t)k, (int64_t)S * k, dD, dI);
            IPX_HIP(hipGetLastError());
            if (ix->timing) IPX_HIP(hipEventRecord(ix->ev[3], st));
            ix->ev_merge = true;
        }
        ix->ev_valid = ix->timing;
        ix->last_wgs = G;
        ix->last_S = S;
        ix->last_cap = cap;
    }

    if (!on_device) {
        IPX_HIP(hipMemcpyAsync(out_D, dD, nout * sizeof(float), hipMemcpyDeviceToHost, st));
        IPX_HIP(hipMemcpyAsync(out_I, dI, nout * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        int errflag = 0;
        if (N > 0) IPX_HIP(hipMemcpyAsync(&errflag, ix->err.p, sizeof(int), hipMemcpyDeviceToHost, st));
        IPX_HIP(hipStreamSynchronize(st));
        if (errflag) return fail(IPX_ERR_STATE, "search: candidate queue overflow (internal invariant violated)");
    }
    return 0;
}

}  // namespace

extern "C" {

// used by ranking_writer.cpp to report through ipx_last_error(); not part of the public ABI
void ipx_set_error_(const char* msg) { g_err = msg ? msg : ""; }

const char* ipx_last_error(void) { return g_err.c_str(); }

const char* ipx_version(void) { return "ipx 0.1 gfx950"; }

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
    ix->device = device;
    *out = ix;
    return 0;
}

int ipx_destroy(ipx_index* ix) {
    if (!ix) return 0;
    (void)hipSetDevice(ix->device);
    if (ix->Xp) (void)hipFree(ix->Xp);
    for (DevBuf* b : {&ix->Qp, &ix->cand, &ix->partD, &ix->partI, &ix->err, &ix->stageIn, &ix->stageQ, &ix->stageW,
                      &ix->outD, &ix->outI, &ix->trace, &ix->pub})
        b->release();
    for (auto& e : ix->ev)
        if (e) (void)hipEventDestroy(e);
    delete ix;
    return 0;
}

int ipx_dim(const ipx_index* ix) { return ix ? ix->d : 0; }
int64_t ipx_ntotal(const ipx_index* ix) { return ix ? ix->ntotal : 0; }

int ipx_reserve(ipx_index* ix, int64_t n) {
    if (!ix || n < 0) return fail(IPX_ERR_ARG, "reserve: bad argument");
    IPX_HIP(hipSetDevice(ix->device));
    return grow_corpus(ix, (n + ipx::BD - 1) / ipx::BD, nullptr);
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
    IPX_HIP(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    if (int rc = grow_corpus(ix, (ix->ntotal + n + ipx::BD - 1) / ipx::BD, st)) return rc;
    const size_t row_bytes = (size_t)ix->d * dtype_size(dtype);
    if (on_device) {
        // pack in slices so one launch's grid stays within 2^31 blocks
        const int64_t slice = std::max<int64_t>(1, (int64_t)(1ull << 30) / ((int64_t)ix->nks * 4));
        for (int64_t r0 = 0; r0 < n; r0 += slice) {
            const int64_t m = std::min(slice, n - r0);
            if (int rc = launch_pack((const char*)rows + r0 * row_bytes, dtype, m, m, ix->d, ix->Xp, ix->ntotal + r0,
                                     ipx::BD, ix->nks, nullptr, 0, 0, st))
                return rc;
        }
    } else {
        // host rows: stage through a bounded device buffer (256 MiB), one copy per slice
        const int64_t slice = std::max<int64_t>(1, (int64_t)(256ull << 20) / (int64_t)row_bytes);
        if (int rc = ix->stageIn.ensure((size_t)std::min(slice, n) * row_bytes)) return rc;
        for (int64_t r0 = 0; r0 < n; r0 += slice) {
            const int64_t m = std::min(slice, n - r0);
            IPX_HIP(hipMemcpyAsync(ix->stageIn.p, (const char*)rows + r0 * row_bytes, (size_t)m * row_bytes,
                                   hipMemcpyHostToDevice, st));
            if (int rc = launch_pack(ix->stageIn.p, dtype, m, m, ix->d, ix->Xp, ix->ntotal + r0, ipx::BD, ix->nks,
                                     nullptr, 0, 0, st))
                return rc;
            IPX_HIP(hipStreamSynchronize(st));   // the staging buffer is reused by the next slice
        }
    }
    ix->ntotal += n;
    return 0;
}

int ipx_search(ipx_index* ix, const float* q, int64_t nq, int k, float* out_D, int64_t* out_I, int on_device,
               void* stream) {
    return search_impl(ix, q, nullptr, 0, 0, nq, k, out_D, out_I, on_device, stream);
}

int ipx_search_aspect(ipx_index* ix, const float* aspect_emb, const float* w, int A, int w_is_logits, int64_t nq,
                      int k, float* out_D, int64_t* out_I, int on_device, void* stream) {
    if (A <= 0 || A > 64) return fail(IPX_ERR_ARG, "search_aspect: A outside [1, 64]");
    return search_impl(ix, aspect_emb, w, A, w_is_logits ? 1 : 0, nq, k, out_D, out_I, on_device, stream);
}

int ipx_merge_topk(const float* D, const int64_t* I, int S, int64_t nq, int k_in, int k_out, float* out_D,
                   int64_t* out_I, int on_device, int device, void* stream) {
    if (S < 0 || nq < 0 || k_in <= 0 || k_out <= 0) return fail(IPX_ERR_ARG, "merge: bad sizes");
    if (k_in > IPX_MAX_K || k_out > IPX_MAX_K)
        return fail(IPX_ERR_ARG, "merge: k outside [1, " + std::to_string(IPX_MAX_K) + "]");
    if (nq == 0) return 0;
    if ((S > 0 && (!D || !I)) || !out_D || !out_I) return fail(IPX_ERR_ARG, "merge: null pointer");
    if (nq > 0x7fffffffLL) return fail(IPX_ERR_ARG, "merge: nq too large");
    IPX_HIP(hipSetDevice(device));
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
    const int P = next_pow2(std::max(k_in, k_out));
    const int mlds = P * 24;
    hipError_t e = hipSuccess;
    if (mlds > 64 * 1024)
        e = hipFuncSetAttribute((const void*)ipx::merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mlds);
    if (e == hipSuccess) {
        // inputs are [S][nq][k_in]: shard stride nq*k_in, query stride k_in; order not assumed
        hipLaunchKernelGGL(ipx::merge_topk_kernel, dim3((unsigned)nq), dim3(256), mlds, st, dD, dI, S, nq, k_in, k_out,
                           P, 0, (int64_t)nq * k_in, (int64_t)k_in, oD, oI);
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

int ipx_set_param(ipx_index* ix, const char* name, int64_t value) {
    if (!ix || !name) return fail(IPX_ERR_ARG, "set_param: null argument");
    if (!strcmp(name, "max_splits")) {
        if (value < 0 || value > 65536) return fail(IPX_ERR_ARG, "set_param: max_splits outside [0, 65536]");
        ix->max_splits = (int)value;
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
    IPX_HIP(hipSetDevice(ix->device));
    const int n = std::min(max_workgroups, ix->last_wgs);
    IPX_HIP(hipDeviceSynchronize());
    IPX_HIP(hipMemcpy(out, ix->trace.p, (size_t)n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
#ifdef IPX_STAMPS
    // diagnostic build: per-wave cycle sums follow the workgroup records
    if (max_workgroups >= 5 * ix->last_wgs)
        IPX_HIP(hipMemcpy(out + (size_t)n * 4, (uint64_t*)ix->trace.p + (size_t)ix->last_wgs * 4,
                          (size_t)n * 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
#endif
    return 0;
}

int ipx_check(ipx_index* ix, void* stream) {
    if (!ix) return fail(IPX_ERR_ARG, "check: null index");
    IPX_HIP(hipSetDevice(ix->device));
    hipStream_t st = (hipStream_t)stream;
    int errflag = 0;
    if (ix->err.p) IPX_HIP(hipMemcpyAsync(&errflag, ix->err.p, sizeof(int), hipMemcpyDeviceToHost, st));
    IPX_HIP(hipStreamSynchronize(st));
    if (errflag) return fail(IPX_ERR_STATE, "search: candidate queue overflow (internal invariant violated)");
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
    IPX_HIP(hipSetDevice(ix->device));
    IPX_HIP(hipEventSynchronize(ix->ev[ix->ev_merge ? 3 : 2]));
    float a = 0.f, b = 0.f, c = 0.f;
    IPX_HIP(hipEventElapsedTime(&a, ix->ev[0], ix->ev[1]));
    IPX_HIP(hipEventElapsedTime(&b, ix->ev[1], ix->ev[2]));
    if (ix->ev_merge) IPX_HIP(hipEventElapsedTime(&c, ix->ev[2], ix->ev[3]));
    if (pack_ms) *pack_ms = a;
    if (scan_ms) *scan_ms = b;
    if (merge_ms) *merge_ms = c;
    return 0;
}

int ipx_last_geometry(const ipx_index* ix, int* workgroups, int* splits, int* queue_cap) {
    if (!ix) return fail(IPX_ERR_ARG, "last_geometry: null index");
    if (workgroups) *workgroups = ix->last_wgs;
    if (splits) *splits = ix->last_S;
    if (queue_cap) *queue_cap = ix->last_cap;
    return 0;
}

}  // extern "C"
