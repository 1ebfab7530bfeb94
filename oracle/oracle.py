"""CPU oracle for the tevatron.faiss_retriever hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker; the product (mural_amd/, tevatron/, libipx.so) never does.

PARITY UNPINNED: faiss (faiss-cpu==1.7.2, /root/reference/docs/index.md:32) is a
third-party dependency that is absent from the reference tree and from this image, and
the reference holds no tests or golden vectors for this path. The classes below restate
the published faiss 1.7.2 algorithm for exactly the calls the reference makes:

  FlatIPOracle        faiss.IndexFlatIP + its numpy wrapper      retriever.py:12,16,19
  ResultHeapOracle    faiss.ResultHeap (python class)             reducer.py:15-18
  mural_pool_att      MULGRAIN_Pooler.forward, 'att' pooling      modeling/mul_grain.py:97-132
                      (this one IS pinned: tests/golden/pooler_att.npz comes from the reference module)

The heap code is C (flat_ip_oracle.c, built by `make -C oracle`); the blocked sgemm that
faiss delegates to BLAS is numpy's matmul (OpenBLAS here).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

FLT_MAX = float(np.finfo(np.float32).max)

# faiss 1.7.2 globals (utils/distances.cpp): queries per sgemm block, database rows per
# sgemm block, and the query count below which it skips BLAS.
BLAS_QUERY_BS = 4096
BLAS_DATABASE_BS = 1024
DISTANCE_COMPUTE_BLAS_THRESHOLD = 20


def build():
    """Compile liboracle.so with gcc (see oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _c():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        i64, vp, i32 = C.c_int64, C.c_void_p, C.c_int
        L.orc_heap_init.argtypes = [i32, i64, i64, vp, vp]
        L.orc_heap_addn.argtypes = [i32, i64, i64, vp, vp, vp, vp, i64, i64, i64]
        L.orc_heap_reorder.argtypes = [i32, i64, i64, vp, vp]
        L.orc_scores_mfma_order.argtypes = [vp, vp, i64, i64, i64, vp]
        L.orc_num_threads.restype = i32
        _lib = L
    return _lib


def num_threads():
    """OpenMP threads the heap stages run on (the sgemm's threads are OpenBLAS's own)."""
    return int(_c().orc_num_threads())


class FlatIPOracle:
    """faiss.IndexFlatIP as the reference uses it (retriever.py:12,16,19)."""

    def __init__(self, d):
        self.d = int(d)
        self._blocks = []
        self.ntotal = 0

    def add(self, x):
        # faiss python wrapper (replacement_add): n, d = x.shape; assert d == self.d;
        # x = np.ascontiguousarray(x, dtype='float32')
        n, d = x.shape
        assert d == self.d
        x = np.ascontiguousarray(x, dtype="float32")
        self._blocks.append(x.copy())
        self.ntotal += n

    def _xb(self):
        if len(self._blocks) > 1:
            self._blocks = [np.concatenate(self._blocks, axis=0)]
        return self._blocks[0] if self._blocks else np.zeros((0, self.d), np.float32)

    def search(self, x, k):
        # wrapper (replacement_search): n, d = x.shape; assert d == self.d; assert k > 0
        n, d = x.shape
        assert d == self.d
        assert k > 0
        x = np.ascontiguousarray(x, dtype="float32")
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        L = _c()
        xb = self._xb()
        L.orc_heap_init(1, n, k, D.ctypes.data, I.ctypes.data)
        if n < DISTANCE_COMPUTE_BLAS_THRESHOLD:
            # exhaustive_inner_product_seq: one query at a time against every row
            for i in range(n):
                for j0 in range(0, self.ntotal, 65536):
                    s = np.ascontiguousarray((xb[j0:j0 + 65536] @ x[i])[None, :], dtype=np.float32)
                    L.orc_heap_addn(1, 1, k, D[i:i + 1].ctypes.data, I[i:i + 1].ctypes.data,
                                    s.ctypes.data, None, s.shape[1], s.shape[1], j0)
        else:
            # exhaustive_inner_product_blas: 4096-query x 1024-row sgemm blocks, heap after each
            for i0 in range(0, n, BLAS_QUERY_BS):
                i1 = min(i0 + BLAS_QUERY_BS, n)
                for j0 in range(0, self.ntotal, BLAS_DATABASE_BS):
                    j1 = min(j0 + BLAS_DATABASE_BS, self.ntotal)
                    s = np.ascontiguousarray(x[i0:i1] @ xb[j0:j1].T, dtype=np.float32)
                    L.orc_heap_addn(1, i1 - i0, k, D[i0:i1].ctypes.data, I[i0:i1].ctypes.data,
                                    s.ctypes.data, None, j1 - j0, j1 - j0, j0)
        L.orc_heap_reorder(1, n, k, D.ctypes.data, I.ctypes.data)
        return D, I


class ResultHeapOracle:
    """faiss.ResultHeap(nq, k) with keep_max=False: keeps the k smallest values (reducer.py:15)."""

    def __init__(self, nq, k):
        self.nq, self.k = int(nq), int(k)
        self.D = np.zeros((nq, k), dtype=np.float32)
        self.I = np.zeros((nq, k), dtype=np.int64)
        _c().orc_heap_init(0, self.nq, self.k, self.D.ctypes.data, self.I.ctypes.data)

    def add_result(self, D, I):
        nq, kd = D.shape
        D = np.ascontiguousarray(D, dtype="float32")
        I = np.ascontiguousarray(I, dtype="int64")   # '<U' id arrays are parsed as integers here
        assert I.shape == (nq, kd)
        assert nq == self.nq
        _c().orc_heap_addn(0, nq, self.k, self.D.ctypes.data, self.I.ctypes.data,
                           D.ctypes.data, I.ctypes.data, kd, kd, 0)

    def finalize(self):
        _c().orc_heap_reorder(0, self.nq, self.k, self.D.ctypes.data, self.I.ctypes.data)


def combine_results(results):
    """reducer.combine_faiss_results (reducer.py:10-20) on the oracle heap."""
    rh = None
    for scores, indices in results:
        if rh is None:
            rh = ResultHeapOracle(scores.shape[0], scores.shape[1])
        rh.add_result(-scores, indices)
    rh.finalize()
    return -rh.D, rh.I


# ---- references that are not faiss restatements --------------------------------------------

def mural_pool_att(hidden, W, b, begin_id=1):
    """MULGRAIN_Pooler.forward, 'att' pooling (modeling/mul_grain.py:97-110 for queries, :117-132 for
    passages — the same arithmetic): att = softmax(h[:,0] W^T + b); rep = att @ h[:, begin:begin+A].
    Pinned by tests/golden/pooler_att.npz, which the reference module itself produced."""
    hidden = np.asarray(hidden, dtype=np.float32)
    W = np.asarray(W, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    A = W.shape[0]
    logits = hidden[:, 0] @ W.T + b
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    att = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    asp = hidden[:, begin_id:begin_id + A]
    rep = np.einsum("na,nad->nd", att, asp).astype(np.float32)
    return rep, att


def exact_topk_f64(Q, X, k):
    """Exact scores in float64; order (score desc, row asc). Returns (D64 [nq,kk], I [nq,kk])."""
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    kk = min(k, X.shape[0])
    idx = np.argsort(-S, axis=1, kind="stable")[:, :kk]
    return np.take_along_axis(S, idx, axis=1), idx


def scores_mfma_order(Q, X):
    """fp32 scores with the summation order of the gfx950 kernel (bit-exact reference)."""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    X = np.ascontiguousarray(X, dtype=np.float32)
    out = np.empty((Q.shape[0], X.shape[0]), dtype=np.float32)
    _c().orc_scores_mfma_order(Q.ctypes.data, X.ctypes.data, Q.shape[0], X.shape[0], Q.shape[1], out.ctypes.data)
    return out


def topk_total_order(S, k, id_base=0):
    """Top-k of fp32 score rows under the engine's total order: score desc, row asc; only
    scores > -FLT_MAX (NaN excluded) are admitted; (-FLT_MAX, -1) padding."""
    nq, n = S.shape
    D = np.full((nq, k), -FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        row = S[q]
        ok = np.nonzero(row > -FLT_MAX)[0]
        order = ok[np.argsort(-row[ok], kind="stable")][:k]
        D[q, :len(order)] = row[order]
        I[q, :len(order)] = order + id_base
    return D, I


def merge_total_order(Ds, Is, k_out):
    """Reference for ipx_merge_topk: union of S lists, order (score desc, id asc)."""
    S, nq, _ = Ds.shape
    D = np.full((nq, k_out), -FLT_MAX, dtype=np.float32)
    I = np.full((nq, k_out), -1, dtype=np.int64)
    for q in range(nq):
        d = Ds[:, q, :].reshape(-1)
        i = Is[:, q, :].reshape(-1)
        ok = d > -FLT_MAX
        d, i = d[ok], i[ok]
        order = np.lexsort((i, -d.astype(np.float64)))[:k_out]
        D[q, :len(order)] = d[order]
        I[q, :len(order)] = i[order]
    return D, I


def check_topk(D, I, Q, X, k, rtol=1e-4, id_base=0):
    """The north-star acceptance rule (BASELINE.json): fp32 scores within 1e-4 relative error
    of the exact ones, and the top-k id list identical to the exact one except where two
    scores fall within that tolerance of each other.

    Formalised per query as: (a) ids are distinct, in range, padding is (-FLT_MAX, -1);
    (b) |D[r] - exact(q, I[r])| <= tol; (c) exact(q, I[r]) is within 2*tol of the exact
    r-th best score, i.e. the list is the exact ranking up to permutations among near-ties.
    tol = rtol*|score| + 2e-6 * sum_i |q_i x_i| (the second term is the fp32 rounding scale of
    a d-term dot product; it only matters for scores near zero, where a relative bound is
    meaningless). Returns a list of violation strings (empty = pass)."""
    Q64, X64 = Q.astype(np.float64), X.astype(np.float64)
    S = Q64 @ X64.T
    Sabs = np.abs(Q64) @ np.abs(X64).T
    kk = min(k, X.shape[0])
    order = np.argsort(-S, axis=1, kind="stable")[:, :kk]
    refD = np.take_along_axis(S, order, axis=1)
    errs = []
    if not (I[:, kk:] == -1).all() or not (D[:, kk:] == np.float32(-FLT_MAX)).all():
        errs.append("padding slots are not (-FLT_MAX, -1)")
    for q in range(D.shape[0]):
        ids = I[q, :kk] - id_base
        if kk and (ids.min() < 0 or ids.max() >= X.shape[0]):
            errs.append(f"q{q}: id out of range")
            continue
        if len(set(ids.tolist())) != kk:
            errs.append(f"q{q}: duplicate ids")
            continue
        exact = S[q, ids]
        tol = rtol * np.abs(exact) + 2e-6 * Sabs[q, ids]
        if not np.all(np.abs(D[q, :kk] - exact) <= tol):
            errs.append(f"q{q}: score error {np.abs(D[q, :kk] - exact).max():.3e}")
        if not np.all(np.abs(exact - refD[q]) <= 2 * tol):
            bad = int(np.argmax(np.abs(exact - refD[q]) > 2 * tol))
            errs.append(f"q{q}: rank {bad} holds doc {ids[bad]} ({exact[bad]:.6f}) but exact is "
                        f"{order[q, bad]} ({refD[q, bad]:.6f})")
        if not np.all(D[q, :kk][:-1] >= D[q, :kk][1:]):
            errs.append(f"q{q}: scores not sorted descending")
        if len(errs) > 5:
            break
    return errs
