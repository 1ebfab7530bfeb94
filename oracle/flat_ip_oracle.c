/*
 * flat_ip_oracle.c — CPU restatement of the arithmetic behind tevatron.faiss_retriever.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (mural_amd/, tevatron/, libipx.so)
 * may import, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the third-party library faiss
 * (faiss-cpu==1.7.2 per /root/reference/docs/index.md:32), which is neither vendored in
 * the reference nor installable here, and the reference has no tests or golden vectors
 * for it. What follows restates faiss 1.7.2's published algorithm for the calls the
 * reference makes:
 *   faiss.IndexFlatIP(d).add / .search     src/tevatron/faiss_retriever/retriever.py:12,16,19
 *     -> knn_inner_product: blocked sgemm (4096 queries x 1024 database rows) in
 *        oracle.py + HeapResultHandler<CMin<float,int64>> below (strict '>' admission,
 *        binary min-heap of k, heap_reorder at the end, (-FLT_MAX, -1) in unused slots)
 *   faiss.ResultHeap(nq,k).add_result / .finalize   src/tevatron/faiss_retriever/reducer.py:15-17
 *     -> float_maxheap_array_t::addn_with_ids + reorder (keeps the k smallest values)
 * plus one helper that has no faiss counterpart: orc_scores_mfma_order reproduces the
 * exact fp32 summation order of the gfx950 kernel (v_mfma_f32_32x32x2_f32 is a k-ordered
 * fmaf chain) so the GPU result can also be checked bit for bit.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads the parallel loops below use (1 when built without OpenMP) */
int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- binary heap over (value, id) pairs, 1-based sifting as in faiss utils/Heap.h ----
 * keep_largest != 0 : CMin ordering (root = smallest kept value; used for inner product)
 * keep_largest == 0 : CMax ordering (root = largest kept value; used by ResultHeap)      */
static inline int before(int keep_largest, float a, float b) {
    return keep_largest ? (a < b) : (a > b);
}

static void sift_from_root(int keep_largest, int64_t k, float* val, int64_t* ids, float v, int64_t id) {
    float* hv = val - 1; /* 1-based */
    int64_t* hi = ids - 1;
    int64_t i = 1;
    for (;;) {
        int64_t c1 = i << 1, c2 = c1 + 1;
        if (c1 > k) break;
        if (c2 == k + 1 || before(keep_largest, hv[c1], hv[c2])) {
            if (before(keep_largest, v, hv[c1])) break;
            hv[i] = hv[c1]; hi[i] = hi[c1]; i = c1;
        } else {
            if (before(keep_largest, v, hv[c2])) break;
            hv[i] = hv[c2]; hi[i] = hi[c2]; i = c2;
        }
    }
    hv[i] = v; hi[i] = id;
}

static float neutral(int keep_largest) { return keep_largest ? -FLT_MAX : FLT_MAX; }

/* heap_heapify with no initial values: every slot neutral / id -1 */
void orc_heap_init(int keep_largest, int64_t nq, int64_t k, float* D, int64_t* I) {
    for (int64_t i = 0; i < nq * k; ++i) { D[i] = neutral(keep_largest); I[i] = -1; }
}

/* HeapResultHandler::add_results / HeapArray::addn_with_ids:
 * vals is [nq][nj] with row stride ld; ids == NULL means id = j0 + column (IndexFlat),
 * otherwise ids is [nq][nj] with the same row stride (ResultHeap). Columns in order;
 * a value enters only if it is strictly better than the root. */
void orc_heap_addn(int keep_largest, int64_t nq, int64_t k, float* D, int64_t* I,
                   const float* vals, const int64_t* ids, int64_t nj, int64_t ld, int64_t j0) {
    /* faiss runs this loop "#pragma omp parallel for" over the queries (HeapResultHandler::add_results);
     * each query's heap is private, so the result does not depend on the thread count */
#pragma omp parallel for schedule(static) if (nq > 1)
    for (int64_t q = 0; q < nq; ++q) {
        float* hd = D + q * k;
        int64_t* hi = I + q * k;
        const float* row = vals + q * ld;
        for (int64_t j = 0; j < nj; ++j) {
            float v = row[j];
            if (before(keep_largest, hd[0], v))
                sift_from_root(keep_largest, k, hd, hi, v, ids ? ids[q * ld + j] : j0 + j);
        }
    }
}

/* heap_reorder: pop everything, best first; unused slots go last as (neutral, -1) */
void orc_heap_reorder(int keep_largest, int64_t nq, int64_t k, float* D, int64_t* I) {
#pragma omp parallel for schedule(static) if (nq > 1)
    for (int64_t q = 0; q < nq; ++q) {
        float* hd = D + q * k;
        int64_t* hi = I + q * k;
        int64_t filled = 0;
        for (int64_t i = 0; i < k; ++i) {
            float v = hd[0];
            int64_t id = hi[0];
            int64_t n = k - i; /* heap_pop on the first n slots: re-sift the last one from the root */
            sift_from_root(keep_largest, n, hd, hi, hd[n - 1], hi[n - 1]);
            hd[k - filled - 1] = v;
            hi[k - filled - 1] = id;
            if (id != -1) ++filled;
        }
        memmove(hd, hd + k - filled, filled * sizeof(float));
        memmove(hi, hi + k - filled, filled * sizeof(int64_t));
        for (int64_t i = filled; i < k; ++i) { hd[i] = neutral(keep_largest); hi[i] = -1; }
    }
}

/* Scores in the summation order of the gfx950 scan kernel (mural_amd/csrc/ipx_kernels.h):
 * the feature axis is zero-padded to a multiple of 16; within every group of 8 features the
 * MFMA chain adds k = 0,4,1,5,2,6,3,7, each step one fmaf with a single rounding.
 * out[q][j] = <Q[q], X[j]>, row stride of out = nb. */
void orc_scores_mfma_order(const float* Q, const float* X, int64_t nq, int64_t nb, int64_t d, float* out) {
    static const int ord[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    const int64_t dpad = (d + 15) / 16 * 16;
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nq; ++q) {
        const float* qv = Q + q * d;
        for (int64_t j = 0; j < nb; ++j) {
            const float* xv = X + j * d;
            float acc = 0.0f;
            for (int64_t g = 0; g < dpad; g += 8)
                for (int e = 0; e < 8; ++e) {
                    int64_t kk = g + ord[e];
                    float a = kk < d ? xv[kk] : 0.0f;
                    float b = kk < d ? qv[kk] : 0.0f;
                    acc = fmaf(a, b, acc);
                }
            out[q * nb + j] = acc;
        }
    }
}
