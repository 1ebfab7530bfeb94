/*
 * ipx.h — C ABI of the MI355X (gfx950) exact inner-product search engine.
 *
 * This is the drop-in boundary for the retrieval hot path of
 * tevatron.faiss_retriever (reference paths are relative to /root/reference/src):
 * each entry point below replaces one call the reference makes into faiss.
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success and a non-zero ipx_status on failure;
 *     ipx_last_error() then returns a thread-local, human-readable message.
 *     Nothing here aborts the process or throws across the boundary.
 *   - `on_device` != 0 means the data pointers are device (HBM) pointers on the
 *     index's GPU; 0 means host pointers (the library stages the copies itself).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Calls with host pointers are synchronous. Calls with device pointers are
 *     stream-ordered; on an index with fp32 rows only (storage 0) they return
 *     without synchronising. With a bf16 copy (storage 1 or 2) ipx_search /
 *     ipx_search_aspect / ipx_search_pooled read back one flag word per query
 *     after the rescoring pass (which queries the margin rule sends to a repeat
 *     pass) and so synchronise the stream once per pass before they return;
 *     ipx_merge_topk synchronises when a list exceeds IPX_MAX_K entries.
 *   - an index serves one call at a time: its scratch buffers, timing events
 *     and last-search statistics are per index, not per call. Concurrent
 *     searches need one index per caller (or external locking); different
 *     indexes on one GPU may run concurrently on different streams (except
 *     through ipx_search_noalloc, see there).
 *   - doc ids are row numbers in insertion order (faiss IndexFlat semantics) plus
 *     the index's id_base; empty result slots hold id -1 and score -FLT_MAX.
 *   - result order: score descending, ties by ascending id (deterministic;
 *     faiss 1.7.2 leaves the order of exactly-equal scores unspecified).
 */
#ifndef IPX_H_
#define IPX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ipx_index ipx_index;

enum ipx_status {
    IPX_OK = 0,
    IPX_ERR_ARG = 1,     /* bad argument (dimension mismatch, k out of range, null pointer) */
    IPX_ERR_HIP = 2,     /* a HIP runtime call failed */
    IPX_ERR_NOMEM = 3,   /* device or host allocation failed */
    IPX_ERR_STATE = 4,   /* internal invariant violated (candidate queue overflow) */
    IPX_ERR_IO = 5       /* file could not be written */
};

enum ipx_dtype { IPX_F32 = 0, IPX_F16 = 1, IPX_BF16 = 2 };

/* Depths. faiss.IndexFlatIP.search takes any k (retriever.py:19, --depth of __main__.py:58), and so do ipx_search /
 * ipx_search_aspect / ipx_search_pooled / ipx_merge_topk, up to IPX_MAX_DEPTH:
 *   k <= IPX_MAX_K     : a query's list is sorted in LDS by the scan itself (the fast path of every kernel; the limit of
 *                        ipx_search_noalloc and ipx_merge_topk_packed).
 *   k <= 16384         : still ONE scan of the corpus — every corpus split keeps its best k keys unsorted in HBM and a
 *                        finishing kernel selects and sorts across the splits (8.8M x 768, 6980 queries: k = 5000 costs
 *                        1.14 x the k = 1000 search). The bf16 scan serves k <= 8192 this way (candidate lists of up to
 *                        16384 entries); an index that keeps only bf16 rows (storage 2) therefore serves k <= 8192
 *                        (k <= 1638 when rows are longer than 2048 elements).
 *   beyond             : one fp32 scan per 16384 ranks, each admitting only hits after the previous scan's last one in
 *                        the result order; the concatenation is bit for bit the one sorted list (needs fp32 rows:
 *                        storage 0 or 1). ipx_merge_topk does the same over its input lists, 2048 ranks per pass. */
#define IPX_MAX_K 2048
#define IPX_MAX_DEPTH (1 << 20)

/* Thread-local message for the last failing call on this thread. */
const char* ipx_last_error(void);

/* Library build tag, e.g. "ipx 0.1 gfx950". */
const char* ipx_version(void);

/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int ipx_device_count(void);

/* Replaces faiss.IndexFlatIP(d)                       retriever.py:12
 * Creates an empty flat inner-product index of dimension d on GPU `device`. */
int ipx_create(int d, int device, ipx_index** out);

/* Releases the index and all device memory it owns (object GC in the reference). */
int ipx_destroy(ipx_index* idx);

/* index.d / index.ntotal of the faiss object          retriever.py:13 (self.index) */
int     ipx_dim(const ipx_index* idx);
int64_t ipx_ntotal(const ipx_index* idx);

/* Pre-sizes device storage for n rows in total (optional; add grows on demand). */
int ipx_reserve(ipx_index* idx, int64_t n);

/* Offset added to every returned row id (row-sharded multi-GPU search; the
 * reference gets global ids from per-shard lookup tables, docs/retrieval.md:34-45). */
int ipx_set_id_base(ipx_index* idx, int64_t id_base);

/* Replaces index.add(p_reps)                          retriever.py:16
 * Appends n rows of dimension d. `rows` is C-contiguous [n, d] of `dtype`
 * (ipx_dtype); f16/bf16 rows are widened to fp32 on the device, which is what
 * `p_reps.astype('float32')` does on the host in __main__.py:79. */
/* (A call that fails part-way — a HIP error in the middle of a multi-slice host copy — leaves ntotal and every
 * search result as they were before the call; the norm bounds of the bf16 margin rule may already include the
 * rows that were not added, which only makes the rule more cautious.) */
int ipx_add(ipx_index* idx, const void* rows, int64_t n, int dtype, int on_device, void* stream);

/* Replaces index.search(q_reps, k)                    retriever.py:19
 * q: fp32 [nq, d]; out_D: fp32 [nq, k]; out_I: int64 [nq, k]. */
int ipx_search(ipx_index* idx, const float* q, int64_t nq, int k,
               float* out_D, int64_t* out_I, int on_device, void* stream);

/* No counterpart in the reference: ipx_search for callers that cannot allocate device memory while they run (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc). Same results as ipx_search, bit
 * for bit. Device pointers only (q, out_D, out_I on the index's GPU), stream-ordered, returns without
 * synchronising (ipx_check(idx) reports an invariant violation flagged by THIS index's searches: the error word is
 * per index, allocated by ipx_create); the index must keep fp32 rows (storage 0 or 1: an index with a bf16 copy is
 * scanned on its fp32 rows here). Calls no hipMalloc / hipFree, creates no events and sets no function attributes
 * (ipx_create resolves the workspace address and the kernels' launch attributes once per device): every intermediate
 * buffer lives in a static device array of
 * IPX_NOALLOC_SCRATCH_BYTES that each GPU the library runs on holds until the process ends. Limits: nq <=
 * IPX_NOALLOC_MAX_QUERIES, k <= IPX_MAX_K, and (nq, k, d) whose buffers fit the array with at least one corpus
 * split (the corpus is cut into as many splits as fit: fewer than ipx_search uses, so it is slower on large
 * indexes); anything beyond is IPX_ERR_ARG with the sizes in the message. All calls on one GPU share that GPU's
 * array: two calls that overlap in time on one GPU (different streams or threads, any indexes) race; calls on one
 * stream are ordered and safe. */
#define IPX_NOALLOC_SCRATCH_BYTES (32u << 20)
#define IPX_NOALLOC_MAX_QUERIES 192
int ipx_search_noalloc(ipx_index* idx, const float* q, int64_t nq, int k,
                       float* out_D, int64_t* out_I, void* stream);

/* MURAL multi-aspect query, fused on the device before the scan:
 *   q[i] = sum_a w[i][a] * aspect_emb[i][a][:]        modeling/mul_grain.py:97-110
 * aspect_emb: fp32 [nq, A, d]; w: fp32 [nq, A]. If w_is_logits != 0 the weights are
 * soft-maxed over a first (mul_grain.py:103-105). The result equals ipx_search on
 * the pre-fused query vectors (inner product is linear in q). */
int ipx_search_aspect(ipx_index* idx, const float* aspect_emb, const float* w, int A,
                      int w_is_logits, int64_t nq, int k,
                      float* out_D, int64_t* out_I, int on_device, void* stream);

/* The MURAL pooler on the device: what MULGRAIN_Pooler.forward ('att' pooling) computes from the
 * encoder's last hidden states                         modeling/mul_grain.py:97-110 (queries), :117-132 (docs)
 *   att = softmax(W h[:,0] + b)        W: fp32 [A, d], b: fp32 [A]   (cls_linear_q / cls_linear_d of pooler.pt)
 *   rep = sum_a att[a] * h[:, begin_id + a]
 * hidden: fp32 [n, seq_len, d], d = ipx_dim(idx) (only row 0 and rows begin_id .. begin_id+A-1 of each
 * sequence are read); out_att (may be NULL): fp32 [n, A], the weights the reference returns as q_att / doc_att.
 * The reference pools at encode time (driver/encode.py:275-278, modeling/encoder.py) and stores rep in the
 * pickles; with these three the encoder's output goes straight into the index or the search.
 *   ipx_pool          rep -> out_rep fp32 [n, d]
 *   ipx_add_pooled    rep appended like ipx_add
 *   ipx_search_pooled rep searched like ipx_search */
int ipx_pool(ipx_index* idx, const float* hidden, int64_t seq_len, const float* W, const float* b, int A,
             int begin_id, int64_t n, float* out_rep, float* out_att, int on_device, void* stream);
int ipx_add_pooled(ipx_index* idx, const float* hidden, int64_t seq_len, const float* W, const float* b, int A,
                   int begin_id, int64_t n, int on_device, void* stream);
int ipx_search_pooled(ipx_index* idx, const float* hidden, int64_t seq_len, const float* W, const float* b, int A,
                      int begin_id, int64_t nq, int k, float* out_D, int64_t* out_I, float* out_att,
                      int on_device, void* stream);

/* Replaces faiss.ResultHeap(nq,k).add_result(-D_s, I_s) for s in shards; finalize()
 *                                                      reducer.py:15-18
 * D: fp32 [S, nq, k_in], I: int64 [S, nq, k_in] (one block per shard, ids already
 * global, id -1 marks an empty slot). Writes the global top k_out per query, sorted
 * (score desc, id asc), padded with (-FLT_MAX, -1). `device` selects the GPU. */
int ipx_merge_topk(const float* D, const int64_t* I, int S, int64_t nq, int k_in, int k_out,
                   float* out_D, int64_t* out_I, int on_device, int device, void* stream);

/* ipx_merge_topk for the exchange format of the row-sharded search (mural_amd/sharded.py; the reference's
 * per-shard result pickles + reducer.py:15-18 play this role): every shard's lists for all queries sit in ONE
 * buffer, so one all-gather moves scores and ids together. Block s of `packed` = int64 ids [nq, k_in] followed
 * by fp32 scores [nq, k_in]; blocks are ipx_packed_block_bytes(nq, k_in) bytes apart (a multiple of 16).
 * Device pointers only, stream-ordered, k_in and k_out <= IPX_MAX_K; the result is ipx_merge_topk's. */
size_t ipx_packed_block_bytes(int64_t nq, int k);
int ipx_merge_topk_packed(const void* packed, int S, int64_t nq, int k_in, int k_out,
                          float* out_D, int64_t* out_I, int device, void* stream);

/* Index options and tuning knobs. Search RESULTS never depend on any of them, with one
 * documented exception: storage 2 defines the exact score on the bf16-rounded rows.
 *   "storage"    : what the index keeps of the added rows; only while the index is empty.
 *                  0 (default) fp32 rows, fp32 MFMA scan.
 *                  1 fp32 rows + a bf16 copy: bf16 MFMA scan for candidates, exact fp32
 *                    rescoring; same scores and ids as storage 0, bit for bit (queries whose
 *                    candidate list cannot be proven sufficient are repeated on the fp32 scan).
 *                  2 bf16 rows only (rows given as bf16 are kept exactly, fp32 / f16 rows are
 *                    rounded to nearest even): bf16 scan, then exact fp32 scores of
 *                    <fp32 query, stored bf16 row>; half the HBM of storage 0; k <= 8192.
 *                  Queries whose candidate list cannot be proven sufficient (more near-ties than the longest
 *                  list holds: thousands of duplicate rows, a degenerate embedding) are repeated — storage 1: on
 *                  the fp32 scan; storage 2: by an exact pass over every stored row, slow but never an error.
 *                  A zero query needs no repeat (all scores are exactly 0: rows in ascending order).
 *                  Storage 1 keeps rows far longer than the rest (|x|^2 above 4 x the mean of the first rows added;
 *                  NaN / inf rows too; at most 256 of them, more count as ordinary rows) out of the bf16 scan: every
 *                  search scores them exactly on the side and merges them in, so they do not widen the other rows'
 *                  margin. Same results, bit for bit.
 *                  A candidate list is only ever trusted while the norm bound of the margin rule is finite: rows with
 *                  inf / NaN elements (or norms that overflow fp32) outside the side list — storage 2 has none, storage 1
 *                  keeps 256 — send every query to the exact path (the bf16 score of such a row can be NaN where the
 *                  exact one is +inf). Rows whose norms spread smoothly (no clear outliers) widen the margins instead:
 *                  the index then learns a longer automatic list from its repeat passes (x1.25 per large batch that
 *                  repeated more than 0.1 % of its queries; forgotten when rows are added or a parameter is set).
 *   "bf16_candidates" : candidates per query the bf16 scan keeps (0 = automatic; at most 16384; never shorter than k,
 *                  and never a way around the depth limits above).
 *   "margin_post": 1 (default) = a candidate list that fails the a-priori sufficiency test (it reaches 2e below
 *                  the k-th approximate score) is still accepted when, after exact rescoring, its last approximate
 *                  score + e lies below the k-th best exact score of its candidates — equally provable, about half
 *                  the margin in practice, so the automatic lists are shorter (2k / 1.75k candidates); 0 = the
 *                  a-priori test alone with its longer lists. Results do not depend on it.
 *   "query_parts": bf16 parts per query in the bf16 scan: 1 = bf16(q); 2 = hi + lo (16 bits of
 *                  the query, twice the MFMA work, a far smaller margin); 0 = automatic.
 *   "f32_scan"   : 1 = storage 1 searches with the fp32 scan (A/B measurements on one index).
 *   "max_splits" : upper bound on the corpus splits per query block (0 = automatic).
 *   "small_f32_limit" : fp32 scans of up to this many queries use the 32/64-query tile that
 *                  streams the corpus at the HBM rate (default 192; results do not depend on it).
 *   "no_small_batch" : 1 = always use the 256-query tiles (A/B measurements, tests).
 *   "bf16_waves", "bf16_shape16", "bf16_ring", "bf16_occ", "bf16_variant", "bf16_map", "bf16_pp" : kernels and kernel
 *                  variants of the large-batch bf16 scan that only a diagnostic build carries (libipx_diag.so: the same
 *                  sources with -DIPX_DIAG, __graft_entry__.build_diag(); see csrc/ipx_bf16_*_kernels.h). The shipped
 *                  library has one large-batch bf16 kernel (16 waves, 2-deep ring; 16x16x32 MFMA with one query part,
 *                  32x32x16 with hi + lo parts) and accepts only these knobs' default values.
 *   "run_tiles"  : fp32 large-tile scan: corpus tiles per run of the dynamic split (0 = automatic).
 *   "trace"      : 1 = every scan workgroup records {start, end (100 MHz ticks), HW_ID,
 *                  compactions}; read back with ipx_read_trace (profiling aid). */
int ipx_set_param(ipx_index* idx, const char* name, int64_t value);
int ipx_read_trace(ipx_index* idx, uint64_t* out, int max_workgroups);

/* Stream-ordered (on_device) searches return before the kernels finish; this waits for
 * `stream` and reports an invariant violation the kernels flagged (IPX_ERR_STATE), if any.
 * Host-pointer searches do this themselves. */
int ipx_check(ipx_index* idx, void* stream);

/* Timing of the kernels of the most recent search call on this index, measured with
 * HIP events on the stream they were launched on. ipx_set_timing(idx,1) enables it.
 * ipx_last_timing synchronises on the recorded events. Any output may be NULL. */
int ipx_set_timing(ipx_index* idx, int enabled);
int ipx_last_timing(ipx_index* idx, float* pack_ms, float* scan_ms, float* merge_ms);

/* bf16 scan bookkeeping of the most recent search: time of the rescoring kernel (needs
 * ipx_set_timing), candidates kept per query (0 = the fp32 scan ran), and how many queries
 * failed the sufficiency check of the first pass and were repeated. Any output may be NULL. */
int ipx_last_bf16_stats(ipx_index* idx, float* rescore_ms, int* candidates, int64_t* fallback_queries);

/* Rows of a storage-1 index the bf16 scan leaves to the side list (see "storage"), as of the most recent bf16
 * search. No counterpart in the reference; bookkeeping for tests. */
int ipx_outlier_rows(const ipx_index* idx);

/* Queries of the most recent bf16 search that no candidate list could serve (the repeat pass with the longest list
 * failed too, or the norm bound is not finite) and that were answered by the fp32 scan (storage 1) or the exact pass
 * over every stored row (storage 2). No counterpart in the reference; bookkeeping for tests and the bench. */
int64_t ipx_last_exact_queries(const ipx_index* idx);

/* Launch geometry chosen by the most recent search: workgroups, corpus splits per
 * query block, candidate-queue capacity. For bench / DESIGN.md bookkeeping. */
int ipx_last_geometry(const ipx_index* idx, int* workgroups, int* splits, int* queue_cap);

/* Replaces the Python loop of write_ranking            __main__.py:32-38
 * Host-only. For each query i (in order) sorts its k hits by score descending (stable,
 * like Python's sorted(..., reverse=True) on the score alone) and writes
 *   f"{qid}\t{docid}\t{score}\n"
 * where score is the float32 widened to double and printed with Python's repr(float).
 * Ids arrive as numpy '<U' cells: UCS-4 code points, zero padded to the given width
 * (qids [nq][qid_width], docids [nq*k][docid_width]) and are written as UTF-8; when
 * docids is NULL, docids_i64 [nq*k] is written in decimal (the reducer's int64 ids). */
int ipx_write_ranking(const char* path, const uint32_t* qids, int qid_width,
                      const uint32_t* docids, int docid_width, const int64_t* docids_i64,
                      const float* scores, int64_t nq, int k);

/* The same file straight from what the search returns            __main__.py:27-28 + :32-38
 * (search_queries' id mapping and write_ranking's loop in one pass, for text output): `rows` int64 [nq*k] are
 * positions in the CLI's lookup list, whose entries arrive as one UTF-8 blob: entry r is
 * doc_blob[doc_offsets[r] .. doc_offsets[r+1] - sep) (n_docs + 1 offsets; sep = bytes of separator behind every
 * entry, 0 or more: "\n".join(ids) is such a blob with sep = 1). Negative rows count from the end, as the
 * reference's list indexing does (a -1 padding id names the last document); rows outside [-n_docs, n_docs) are
 * IPX_ERR_ARG ("list index out of range"). The [nq, k] array of id strings is never built. */
int ipx_write_ranking_lookup(const char* path, const uint32_t* qids, int qid_width, const char* doc_blob,
                             const int64_t* doc_offsets, int sep, int64_t n_docs, const int64_t* rows,
                             const float* scores, int64_t nq, int k);

/* Python repr(float(np.float32(x))) into buf (>= 32 bytes). Returns the length. */
int ipx_format_score(float x, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* IPX_H_ */
