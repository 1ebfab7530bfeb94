"""GPU parity tests: every call goes through the C ABI of libipx.so (mural_amd/_lib.py) and is
checked against the CPU oracle on the same seeded inputs.

Two bars, both written here:
  * BIT-EXACT against the oracle's kernel-order scores (oracle.scores_mfma_order) and the
    engine's total order (score desc, row asc): ids identical, score bits identical.
  * the north-star tolerance rule against the faiss-algorithm restatement (FlatIPOracle) and
    float64 ground truth: scores within 1e-4 relative, id lists identical except where two
    scores fall within that tolerance of each other (oracle.check_topk).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, gaussian
from oracle import oracle as O

pytestmark = pytest.mark.gpu

FLT_MAX = np.float32(np.finfo(np.float32).max)


@pytest.fixture(scope="module")
def R():
    from mural_amd.faiss_retriever import retriever
    from mural_amd import _lib
    assert _lib.lib().ipx_device_count() >= 1, "no HIP device visible: the search path has no CPU fallback"
    return retriever


def build(R, X, adds=1, dtype=None):
    r = R.BaseFaissIPRetriever(X)
    step = -(-X.shape[0] // adds) if X.shape[0] else 1
    for lo in range(0, X.shape[0], step):
        part = X[lo:lo + step]
        r.add(part.astype(dtype) if dtype else part)
    assert r.index.ntotal == X.shape[0] and r.index.d == X.shape[1]
    return r


def assert_bit_exact(D, I, Q, X, k, id_base=0):
    rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k, id_base=id_base)
    assert np.array_equal(I, rI), f"ids differ at {np.argwhere(I != rI)[:3].tolist()}"
    assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), "score bits differ"


# ---- search: bit-exact + tolerance rule ------------------------------------------------------
@pytest.mark.parametrize("n,d,nq,k,adds", [
    (4096, 768, 32, 1, 1), (4096, 768, 32, 10, 1), (4096, 768, 32, 100, 1), (4096, 768, 32, 1000, 1),
    (5000, 768, 300, 100, 1),          # two query blocks, ragged last corpus tile
    (20000, 100, 513, 100, 3),         # d not a multiple of 16, three adds, 3 query blocks
    (9000, 17, 64, 2048, 2),           # largest k, odd d
    (777, 768, 1, 100, 1), (777, 768, 19, 10, 1), (777, 768, 20, 10, 1), (777, 768, 21, 10, 1),
    (128, 16, 256, 128, 1), (129, 16, 257, 5, 1), (1, 8, 3, 4, 1),
    # ragged last query block: 160 / 200 / 97 real queries = 3+2 / 4+3 / 2+2 live 32-query groups per wave column
    (6000, 48, 416, 50, 1), (6000, 48, 456, 7, 2), (3000, 96, 353, 100, 1),
])
def test_search_matches_oracle(R, n, d, nq, k, adds):
    X, Q = gaussian(1234, (n, d)), gaussian(4321, (nq, d))
    r = build(R, X, adds)
    D, I = r.search(Q, k)
    assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (nq, k) == I.shape
    assert_bit_exact(D, I, Q, X, k)
    assert O.check_topk(D, I, Q, X, k) == []
    # the faiss-algorithm restatement: same scores within 1e-4 relative
    fo = O.FlatIPOracle(d)
    fo.add(X)
    fD, fI = fo.search(Q, k)
    kk = min(k, n)
    assert np.all(np.abs(fD[:, :kk] - D[:, :kk]) <= 1e-4 * np.abs(fD[:, :kk]) + 1e-4)
    assert np.array_equal(fI[:, kk:], I[:, kk:]) and np.array_equal(fD[:, kk:], D[:, kk:])
    assert (fI == I).mean() > 0.99


def test_long_scan_many_compactions(R):
    # one workgroup per query block walks the whole corpus: the candidate queues fill and are
    # compacted many times, thresholds rise tile by tile
    X, Q = gaussian(99, (150000, 64)), gaussian(98, (300, 64))
    r = build(R, X)
    S = O.scores_mfma_order(Q, X)
    for max_splits in (1, 3, 0):
        r.index.set_param("max_splits", max_splits)
        for k in (100, 1000):
            D, I = r.search(Q, k)
            rD, rI = O.topk_total_order(S, k)
            assert np.array_equal(I, rI) and np.array_equal(D, rD), (max_splits, k)
    assert r.index.last_geometry()["splits"] > 3


def test_ascending_scores_worst_case_for_the_filter(R):
    # every new row beats the running threshold: maximum queue pressure
    n, d = 40000, 16
    X = np.zeros((n, d), np.float32)
    X[:, 0] = np.arange(n, dtype=np.float32)
    Q = np.zeros((5, d), np.float32)
    Q[:, 0] = np.array([1, 2, 0.5, -1, 0], np.float32)
    r = build(R, X)
    r.index.set_param("max_splits", 1)
    for k in (1, 100, 2048):
        D, I = r.search(Q, k)
        assert_bit_exact(D, I, Q, X, k)
    assert (I[0, :3] == [n - 1, n - 2, n - 3]).all() and (I[3, :3] == [0, 1, 2]).all()
    assert (I[4, :5] == np.arange(5)).all()       # all-equal scores: rows ascending


def test_k_larger_than_ntotal_and_empty_index(R):
    X, Q = gaussian(1, (37, 64)), gaussian(2, (25, 64))
    r = build(R, X)
    D, I = r.search(Q, 100)
    assert (I[:, 37:] == -1).all() and (D[:, 37:] == -FLT_MAX).all()
    assert_bit_exact(D, I, Q, X, 100)
    e = R.BaseFaissIPRetriever(X)                 # constructor does not add init_reps
    assert e.index.ntotal == 0
    D, I = e.search(Q, 7)
    assert (I == -1).all() and (D == -FLT_MAX).all()
    D, I = r.search(Q[:0], 5)
    assert D.shape == (0, 5) and I.shape == (0, 5)


def test_exact_ties_break_on_row_id(R):
    base = gaussian(3, (300, 32))
    X = np.concatenate([base, base, base[:50]], axis=0)
    Q = gaussian(4, (40, 32))
    r = build(R, X)
    D, I = r.search(Q, 9)
    assert_bit_exact(D, I, Q, X, 9)
    assert (I[:, 0] < 300).all() and np.array_equal(D[:, 0], D[:, 1])
    assert O.check_topk(D, I, Q, X, 9) == []


def test_nan_and_inf_rows(R):
    X = gaussian(7, (500, 16))
    X[5, 0] = np.nan
    X[9, :] = 0
    X[9, 0] = np.inf
    Q = np.abs(gaussian(8, (30, 16))) + 0.1
    r = build(R, X)
    for q in (Q, -Q):
        D, I = r.search(q, 500)
        assert_bit_exact(D, I, q, X, 500)
        fo = O.FlatIPOracle(16)
        fo.add(X)
        fD, fI = fo.search(q, 500)
        assert np.array_equal(np.sort(fI, axis=1), np.sort(I, axis=1))   # same admitted set as the heap
    assert not (I == 5).any() and not (I == 9).any() and (I[:, -2:] == -1).all()


def test_float16_shards_are_widened_on_device(R):
    X16 = gaussian(11, (3000, 64)).astype(np.float16)
    Q = gaussian(12, (33, 64))
    r = build(R, X16.astype(np.float32))
    r16 = R.BaseFaissIPRetriever(X16)
    r16.add(X16)
    a, b = r.search(Q, 50), r16.search(Q, 50)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_dimension_mismatch_and_k_errors(R):
    r = build(R, gaussian(0, (10, 8)))
    with pytest.raises(AssertionError):
        r.add(np.zeros((3, 9), np.float32))
    with pytest.raises(AssertionError):
        r.search(np.zeros((3, 9), np.float32), 5)
    with pytest.raises(AssertionError):
        r.search(np.zeros((3, 8), np.float32), 0)
    with pytest.raises(ValueError):
        r.search(np.zeros((3, 8), np.float32), (1 << 20) + 1)       # IPX_MAX_DEPTH


# ---- k beyond one LDS sort (IPX_MAX_K = 2048): faiss takes any k (reference retriever.py:19, --depth) ------
# up to DEEP_MAX_K = 16384 in one scan (lists left unsorted per split, deep_finish_kernel), beyond that one scan per
# 16384 ranks under a ceiling
@pytest.mark.parametrize("n,d,nq,k,storage", [
    (20000, 64, 37, 5000, "f32"),          # one query block spread over every CU
    (20000, 64, 300, 5000, "f32"),         # two query blocks
    (20000, 64, 300, 4096, "bf16scan"),    # an index that also keeps a bf16 copy
    (3000, 32, 70, 5000, "f32"),           # k > ntotal: padding
    (2049, 16, 5, 2049, "f32"),            # one rank beyond a sorted list
    (40000, 16, 70, 16384, "f32"),         # the deepest single pass: one workgroup per CU
    (40000, 16, 9, 16385, "f32"),          # one rank beyond it: a second pass under a ceiling
    (50000, 16, 5, 40000, "f32"),          # three passes
    (9000, 24, 260, 9000, "f32"),          # every row, sorted
])
def test_search_deeper_than_one_pass(R, n, d, nq, k, storage):
    X, Q = gaussian(77, (n, d)), gaussian(78, (nq, d))
    r = R.BaseFaissIPRetriever(X)
    r.index = R.FlatIPIndex(d, storage=storage)
    r.add(X)
    D, I = r.search(Q, k)
    assert D.shape == (nq, k) == I.shape
    assert_bit_exact(D, I, Q, X, k)
    kk = min(k, n)
    assert (I[:, kk:] == -1).all() and (D[:, kk:] == -FLT_MAX).all()
    srt = np.sort(I[:, :kk], axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                         # no row twice across the pass boundaries
    # a pass boundary does not show: the single-pass list is the prefix
    D1, I1 = r.search(Q, 2048)
    assert np.array_equal(I1, I[:, :2048]) and np.array_equal(D1, D[:, :2048])


def test_deep_search_ties_across_the_pass_boundary(R):
    # 6000 rows, three copies of 2000: every score appears three times, the triples straddle rank 2048 and 4096
    base = gaussian(5, (2000, 32))
    X = np.concatenate([base, base, base], axis=0)
    Q = gaussian(6, (270, 32))
    r = build(R, X)
    D, I = r.search(Q, 6000)
    assert_bit_exact(D, I, Q, X, 6000)
    assert np.array_equal(D[:, 0::3], D[:, 1::3]) and (I[:, 0::3] < I[:, 1::3]).all() and (I[:, 1::3] < I[:, 2::3]).all()
    # all-equal scores: rows ascending through every pass
    Z = np.zeros((5000, 16), np.float32)
    rz = build(R, Z)
    Dz, Iz = rz.search(np.ones((3, 16), np.float32), 5000)
    assert (Dz == 0).all() and (Iz == np.arange(5000)).all()
    # the same across the boundary of two scans (rank 16384 = 3 * 5461 + 1: inside a triple), and 20000 all-equal scores
    base = gaussian(7, (7000, 16))
    X = np.concatenate([base, base, base], axis=0)
    Q = gaussian(8, (5, 16))
    r = build(R, X)
    D, I = r.search(Q, 21000)
    assert_bit_exact(D, I, Q, X, 21000)
    assert np.array_equal(D[:, 0::3], D[:, 1::3]) and (I[:, 0::3] < I[:, 1::3]).all() and (I[:, 1::3] < I[:, 2::3]).all()
    Z = np.zeros((20000, 8), np.float32)
    rz = build(R, Z)
    Dz, Iz = rz.search(np.ones((2, 8), np.float32), 20000)
    assert (Dz == 0).all() and (Iz == np.arange(20000)).all()


def test_deep_search_skips_nan_rows(R):
    X = gaussian(7, (5000, 16))
    X[5, 0] = np.nan
    X[4000, 1] = np.nan
    Q = gaussian(8, (9, 16))
    r = build(R, X)
    D, I = r.search(Q, 5000)
    assert_bit_exact(D, I, Q, X, 5000)
    assert (I[:, -2:] == -1).all() and not (I == 5).any() and not (I == 4000).any()


def test_batch_search_equals_one_call_bitwise(R):
    # a score here is one fmaf chain per (query, row): it cannot depend on the batch it is in
    X, Q = gaussian(21, (6000, 128)), gaussian(22, (301, 128))
    r = build(R, X)
    D, I = r.search(Q, 64)
    for bs in (128, 7, 1000):
        bD, bI = r.batch_search(Q, 64, bs, quiet=True)
        assert np.array_equal(bD, D) and np.array_equal(bI, I)


def test_topk_prefix_property(R):
    X, Q = gaussian(31, (30000, 96)), gaussian(32, (130, 96))
    r = build(R, X)
    D1000, I1000 = r.search(Q, 1000)
    for k in (1, 10, 100, 999):
        D, I = r.search(Q, k)
        assert np.array_equal(D, D1000[:, :k]) and np.array_equal(I, I1000[:, :k])


# ---- merge / reducer -------------------------------------------------------------------------
def gpu_merge(Ds, Is, k_out):
    from mural_amd import _lib
    S, nq, k_in = Ds.shape
    Ds, Is = np.ascontiguousarray(Ds, np.float32), np.ascontiguousarray(Is, np.int64)
    oD, oI = np.empty((nq, k_out), np.float32), np.empty((nq, k_out), np.int64)
    _lib.check(_lib.lib().ipx_merge_topk(C.c_void_p(Ds.ctypes.data), C.c_void_p(Is.ctypes.data), S, nq, k_in, k_out,
                                         C.c_void_p(oD.ctypes.data), C.c_void_p(oI.ctypes.data), 0, 0, None))
    return oD, oI


@pytest.mark.parametrize("S,nq,k_in,k_out", [(1, 5, 10, 10), (4, 33, 100, 100), (8, 20, 1000, 1000),
                                             (20, 7, 100, 100), (3, 9, 50, 20), (3, 9, 20, 50), (2, 3, 2048, 2048),
                                             (5, 4, 1, 1),
                                             # beyond one pass: long lists, long results, both; more queries than a batch
                                             (3, 9, 5000, 5000), (8, 4, 5000, 100), (4, 6, 1500, 5000), (2, 3, 2049, 6000),
                                             (2, 700, 2100, 2100)])
def test_merge_topk_matches_reference(R, S, nq, k_in, k_out):
    rng = np.random.default_rng(S * 1000 + k_in)
    Ds = rng.standard_normal((S, nq, k_in)).astype(np.float32)          # unsorted on purpose
    Is = rng.permutation(S * nq * k_in).reshape(S, nq, k_in).astype(np.int64) * 7919 - 5
    Ds[0, 0, : min(3, k_in)] = -FLT_MAX                                    # empty slots
    Is[0, 0, : min(3, k_in)] = -1
    if k_in > 4:
        Ds[:, 1, 4] = 0.25                                                 # exact ties across shards
        Ds[0, 2, 2] = np.nan
    oD, oI = gpu_merge(Ds, Is, k_out)
    rD, rI = O.merge_total_order(Ds, Is, k_out)
    assert np.array_equal(oI, rI) and np.array_equal(oD, rD)


def test_deep_merge_keeps_duplicate_entries_across_a_pass_boundary(R):
    """One document in two shards' lists (same score, same id) is returned twice by ResultHeap (reducer.py:15-18) and
    by the single-pass merge; a merge beyond IPX_MAX_K entries must do the same when its pass boundary (rank 2048,
    4096, ...) falls between the copies — it once returned the entry a single time there."""
    rng = np.random.default_rng(12)
    S, nq, k_in, k_out = 3, 4, 3000, 6000
    Ds = np.sort(rng.standard_normal((S, nq, k_in)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    Is = rng.permutation(S * nq * k_in).reshape(S, nq, k_in).astype(np.int64)
    for q in range(nq):
        order = np.argsort(-Ds[:, q, :].ravel(), kind="stable")
        # an entry near rank 2047 (the last of pass 1) gets copies in the other shards' lists (written over their tail
        # entries, far down the order): q = 0, 1: one copy, the pair sits on ranks 2047 / 2048; q = 2: two copies, ranks
        # 2047..2049; q = 3: the entry of rank 2045 four times more, ranks 2045..2049 (three copies returned by pass 1,
        # two left for pass 2). Then the entry that ends pass 2 (rank 4095 after the shift) gets one copy.
        r1, c1 = {0: (2047, 1), 1: (2047, 1), 2: (2047, 2), 3: (2045, 4)}[q]
        for rank, copies, skip in ((r1, c1, 0), (4095 - c1, 1, 8)):
            s0, j0 = divmod(int(order[rank]), k_in)
            for c in range(copies):
                s1 = (s0 + 1 + c % (S - 1)) % S
                j1 = k_in - 1 - c - skip
                Ds[s1, q, j1], Is[s1, q, j1] = Ds[s0, q, j0], Is[s0, q, j0]
    oD, oI = gpu_merge(Ds, Is, k_out)
    rD, rI = O.merge_total_order(Ds, Is, k_out)
    assert np.array_equal(oI, rI), np.argwhere(oI != rI)[:4].tolist()
    assert np.array_equal(oD, rD)
    for q in range(nq):                     # the construction holds: copies sit on both sides of ranks 2048 and 4096
        assert rI[q, 2047] == rI[q, 2048] and rD[q, 2047] == rD[q, 2048]
        assert rI[q, 4095] == rI[q, 4096] and rD[q, 4095] == rD[q, 4096]


def test_merge_equals_result_heap_oracle(R):
    # the faiss ResultHeap restatement keeps the same set; order among exact ties may differ
    rng = np.random.default_rng(3)
    Ds = np.sort(rng.standard_normal((6, 40, 100)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    Is = rng.permutation(6 * 40 * 100).reshape(6, 40, 100).astype(np.int64)
    oD, oI = gpu_merge(Ds, Is, 100)
    cD, cI = O.combine_results(zip(Ds, Is))
    assert np.array_equal(oD, cD) and np.array_equal(oI, cI)


def test_combine_faiss_results_golden(R, golden, capsys):
    from mural_amd.faiss_retriever.reducer import combine_faiss_results
    shards = list(zip(golden["D_shard_scores"], golden["D_shard_ids"]))     # '<U' int-like ids
    cD, cI = combine_faiss_results(iter(shards))
    assert capsys.readouterr().out == str(golden["D_stdout"])
    assert cD.dtype == np.float32 and cI.dtype == np.int64
    assert np.array_equal(cD, golden["D_expected_scores"])
    eI = golden["D_expected_ids"]
    for q in range(cD.shape[0]):                   # ids equal up to order inside exact-tie groups
        for s in np.unique(cD[q]):
            m = cD[q] == s
            assert sorted(cI[q][m]) == sorted(eI[q][m])
    with pytest.raises(ValueError):
        combine_faiss_results([(np.zeros((1, 2), np.float32), np.array([["B0048", "x"]]))])
    # more shards than one merge call takes, different k per shard
    rng = np.random.default_rng(8)
    many = [(rng.standard_normal((5, 6 + (s % 3))).astype(np.float32), rng.integers(0, 10 ** 6, (5, 6 + (s % 3))))
            for s in range(19)]
    cD, cI = combine_faiss_results(iter(many))
    eD, eI = O.combine_results(many)
    assert np.array_equal(cD, eD) and np.array_equal(cI, eI)


def test_logical_shards_merge_to_the_unsharded_result(R):
    # G row shards -> local top-k with global ids -> merge == one index
    X, Q = gaussian(41, (10007, 80)), gaussian(42, (150, 80))
    full = build(R, X)
    from mural_amd.sharded import shard_bounds
    for k in (10, 1000):
        D, I = full.search(Q, k)
        for G in (1, 2, 4, 8):
            Ds, Is = [], []
            for g in range(G):
                lo, hi = shard_bounds(X.shape[0], G, g)
                r = build(R, X[lo:hi])
                r.index.set_id_base(lo)
                d, i = r.search(Q, k)
                Ds.append(d), Is.append(i)
            mD, mI = gpu_merge(np.stack(Ds), np.stack(Is), k)
            assert np.array_equal(mD, D) and np.array_equal(mI, I), (k, G)


# ---- MURAL aspect fusion ---------------------------------------------------------------------
def test_aspect_fused_query_equals_prefused_search(R):
    n, d, nq, A = 5000, 768, 70, 3
    X = gaussian(51, (n, d))
    E = gaussian(52, (nq, A, d))
    logits = gaussian(53, (nq, A))
    w = np.exp(logits - logits.max(1, keepdims=True))
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    r = build(R, X)
    fused = (w[:, :, None].astype(np.float64) * E.astype(np.float64)).sum(1).astype(np.float32)
    D0, I0 = r.search(fused, 100)
    for D, I in (r.index.search_aspect(E, w, 100), r.index.search_aspect(E, logits, 100, logits=True)):
        assert O.check_topk(D, I, fused, X, 100) == []
        assert np.all(np.abs(D - D0) <= 1e-4 * np.abs(D0))
        assert (I == I0).mean() > 0.99
    # exactly representable weights: the fused vector is exact, results are bit-identical
    w2 = np.tile(np.array([0.5, 0.25, 0.25], np.float32), (nq, 1))
    E2 = np.round(E * 64) / 64
    fused2 = (w2[:, :, None] * E2).sum(1).astype(np.float32)
    a, b = r.index.search_aspect(E2, w2, 50), r.search(fused2, 50)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- CLI end to end against the reference's own output ---------------------------------------
def _parse(text):
    rows = [ln.split("\t") for ln in text.decode().splitlines()]
    return [(q, d, float(s)) for q, d, s in rows]


def test_cli_end_to_end_golden(R, golden, tmp_path, monkeypatch):
    from mural_amd.faiss_retriever import __main__ as drv
    out = tmp_path / "rank.txt"
    drv.main(["--query_reps", os.path.join(GOLDEN_DIR, "E_query_emb.pkl"),
              "--passage_reps", os.path.join(GOLDEN_DIR, "E_corpus_emb*.pkl"),
              "--depth", "10", "--batch_size", "4", "--save_ranking_to", str(out), "--save_text", "--quiet"])
    got, exp = _parse(out.read_bytes()), _parse(golden["E_expected_ranking"].tobytes())
    assert len(got) == len(exp) == 90
    assert [g[0] for g in got] == [e[0] for e in exp]
    for g, e in zip(got, exp):
        assert abs(g[2] - e[2]) <= 1e-4 * abs(e[2]) + 1e-6
        assert g[1] == e[1] or abs(g[2] - e[2]) <= 1e-4 * abs(e[2])
    assert sum(g[1] == e[1] for g, e in zip(got, exp)) >= 88
    # pickle output (no --save_text) feeds the reducer: (scores f32, ids '<U')
    outp = tmp_path / "rank.pkl"
    drv.main(["--query_reps", os.path.join(GOLDEN_DIR, "E_query_emb.pkl"),
              "--passage_reps", os.path.join(GOLDEN_DIR, "E_corpus_emb0.pkl"),
              "--depth", "10", "--batch_size", "-100", "--save_ranking_to", str(outp), "--quiet"])
    import pickle
    with open(outp, "rb") as f:                     # written by this test a moment ago
        s, p = pickle.load(f)
    assert s.dtype == np.float32 and s.shape == (9, 10) and p.dtype.kind == "U" and p.shape == (9, 10)
    assert all(x.startswith("doc") for x in p.reshape(-1))


# ---- torch device-pointer path, single-rank sharded wrapper ----------------------------------
def test_device_pointer_api_and_single_rank_sharded(R):
    import torch
    X, Q = gaussian(61, (7000, 128)), gaussian(62, (90, 128))
    r = build(R, X)
    D, I = r.search(Q, 32)
    dev = torch.device("cuda:0")
    idx = R.FlatIPIndex(128, device=0)
    idx.add(torch.from_numpy(X[:4000]).to(dev))
    idx.add(torch.from_numpy(X[4000:]).to(dev).to(torch.bfloat16).float())   # f32 values on a bf16 grid
    Xb = X.copy()
    Xb[4000:] = torch.from_numpy(X[4000:]).to(torch.bfloat16).float().numpy()
    tD, tI = idx.search_device(torch.from_numpy(Q).to(dev), 32)
    idx.check()
    assert_bit_exact(tD.cpu().numpy(), tI.cpu().numpy(), Q, Xb, 32)
    # bf16 rows given as bf16 are widened on the device to the same values
    idx2 = R.FlatIPIndex(128, device=0)
    idx2.add(torch.from_numpy(X[:4000]).to(dev))
    idx2.add(torch.from_numpy(X[4000:]).to(dev).to(torch.bfloat16))
    uD, uI = idx2.search_device(torch.from_numpy(Q).to(dev), 32)
    assert torch.equal(uD, tD) and torch.equal(uI, tI)
    from mural_amd.sharded import ShardedFlatIP
    sh = ShardedFlatIP(128, row_lo=1000, device=0)
    sh.add_local(torch.from_numpy(X).to(dev))
    sD, sI = sh.search(torch.from_numpy(Q).to(dev), 32)
    assert np.array_equal(sD.cpu().numpy(), D) and np.array_equal(sI.cpu().numpy(), I + 1000)
    # the exchange forced on one rank (no process group: packed block -> merge; deep lists: per-array path)
    for k in (32, 2100):
        eD, eI = sh.search(torch.from_numpy(Q).to(dev), k, exchange=True)
        pD, pI = sh.search(torch.from_numpy(Q).to(dev), k)
        assert torch.equal(eD, pD) and torch.equal(eI, pI)


def test_merge_packed_equals_merge(R):
    """ipx_merge_topk_packed reads the shards' (ids, scores) blocks as one all-gather delivers them."""
    import torch
    from mural_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(17)
    for S, nq, k_in, k_out in ((1, 1, 1, 1), (8, 1, 100, 100), (8, 64, 100, 100), (3, 7, 33, 50), (4, 5, 2048, 2048), (2, 9, 7, 3)):
        Ds = rng.standard_normal((S, nq, k_in)).astype(np.float32)
        Is = rng.permutation(S * nq * k_in).reshape(S, nq, k_in).astype(np.int64) * 4099 + (1 << 33)
        Ds[0, 0, :1] = -FLT_MAX
        Is[0, 0, :1] = -1
        if k_in > 4:
            Ds[:, 0, 4] = 0.5                                               # ties across shards
        block = L.ipx_packed_block_bytes(nq, k_in)
        assert block % 16 == 0 and block >= nq * k_in * 12
        buf = np.zeros((S, block), np.uint8)
        for s in range(S):
            buf[s, : nq * k_in * 8] = Is[s].reshape(-1).view(np.uint8)
            buf[s, nq * k_in * 8: nq * k_in * 12] = Ds[s].reshape(-1).view(np.uint8)
        t = torch.from_numpy(buf).to(dev)
        oD = torch.empty((nq, k_out), dtype=torch.float32, device=dev)
        oI = torch.empty((nq, k_out), dtype=torch.int64, device=dev)
        _lib.check(L.ipx_merge_topk_packed(C.c_void_p(t.data_ptr()), S, nq, k_in, k_out, C.c_void_p(oD.data_ptr()),
                                           C.c_void_p(oI.data_ptr()), 0, None))
        torch.cuda.synchronize()
        rD, rI = O.merge_total_order(Ds, Is, k_out)
        assert np.array_equal(oI.cpu().numpy(), rI) and np.array_equal(oD.cpu().numpy(), rD), (S, nq, k_in, k_out)
    assert L.ipx_merge_topk_packed(C.c_void_p(t.data_ptr()), 2, 9, 5000, 3, C.c_void_p(oD.data_ptr()), C.c_void_p(oI.data_ptr()), 0, None) == 1


# ---- BASELINE.json configs[1] size: properties that need no CPU reference --------------------
def test_config2_scale_properties(R):
    """1.8M x 768 fp32 corpus, 8192 queries, top-100 on one GPU (BASELINE.json configs[1]).
    Size-independent checks: sharded == unsharded, top-100 is the prefix of top-1000 for a
    query slice, planted rows are found at rank 0, scores equal a float64 recomputation."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    n, d, nq = 1_800_000, 768, 8192
    X = torch.randn((n, d), generator=g, device=dev, dtype=torch.float32)
    g.manual_seed(4321)
    Q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)
    plant = torch.arange(0, nq, 64, device=dev)
    rows = (plant * 211 + 17) % n
    X[rows] = Q[plant] * 3.0                         # a row no random row can beat
    idx = R.FlatIPIndex(d, device=0)
    idx.reserve(n)
    for lo in range(0, n, 300_000):
        idx.add(X[lo:lo + 300_000])
    D, I = idx.search_device(Q, 100)
    idx.check()
    assert torch.equal(I[plant, 0], rows)
    assert bool((D[:, :-1] >= D[:, 1:]).all()) and int(I.min()) >= 0 and int(I.max()) < n
    # scores of the returned ids, recomputed in float64
    qs = torch.arange(0, nq, 257, device=dev)
    ref = torch.einsum("qkd,qd->qk", X[I[qs]].double(), Q[qs].double())
    assert bool(((D[qs].double() - ref).abs() <= 1e-4 * ref.abs()).all())
    # nothing outside the returned list beats the k-th score (beyond tolerance)
    full = Q[qs[:8]] @ X.T
    kth = D[qs[:8], -1]
    above = (full > (kth * (1 + 1e-4))[:, None]).sum(1)
    assert bool((above <= 100).all())
    # top-100 is the prefix of top-1000
    D1k, I1k = idx.search_device(Q[:512], 1000)
    assert torch.equal(D1k[:, :100], D[:512]) and torch.equal(I1k[:, :100], I[:512])
    # 4 logical row shards merged on the GPU == the unsharded result
    from mural_amd.sharded import shard_bounds, _HipEngine
    parts_D, parts_I = [], []
    for s in range(4):
        lo, hi = shard_bounds(n, 4, s)
        e = _HipEngine(d, 0)
        e.set_id_base(lo)
        e.add(X[lo:hi])
        pd, pi = e.search(Q, 100)
        parts_D.append(pd), parts_I.append(pi)
        torch.cuda.synchronize()
        del e
    mD, mI = _HipEngine(d, 0).merge(torch.stack(parts_D), torch.stack(parts_I), 100)
    assert torch.equal(mD, D) and torch.equal(mI, I)


def test_cli_torchrun_mode_one_rank_equals_plain_cli(R, tmp_path, monkeypatch):
    # the multi-process CLI path (_main_sharded) with a 1-rank RCCL group on this GPU: HIP engine wiring,
    # host rows / host queries in, byte-identical ranking out
    import socket
    from mural_amd.faiss_retriever import __main__ as drv
    argv = ["--query_reps", os.path.join(GOLDEN_DIR, "E_query_emb.pkl"),
            "--passage_reps", os.path.join(GOLDEN_DIR, "E_corpus_emb*.pkl"),
            "--depth", "10", "--batch_size", "4", "--save_text", "--quiet", "--save_ranking_to"]
    drv.main(argv + [str(tmp_path / "plain.tsv")])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    for key, val in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"),
                     ("MASTER_PORT", str(port))):
        monkeypatch.setenv(key, val)
    drv._main_sharded(drv._parse_args(argv + [str(tmp_path / "sharded.tsv")]))
    assert (tmp_path / "sharded.tsv").read_bytes() == (tmp_path / "plain.tsv").read_bytes()


def test_merge_negative_zero_ties_with_zero(R):
    # -0.0 == +0.0 under float compares (faiss's order): the tie breaks on the id, not on the sign bit
    Ds = np.array([[[0.0, -0.0, 1.0, -0.0]], [[-0.0, 0.0, -1.0, 0.0]]], np.float32)
    Is = np.array([[[40, 10, 5, 30]], [[20, 50, 6, 0]]], np.int64)
    oD, oI = gpu_merge(Ds, Is, 8)
    assert oI.tolist() == [[5, 0, 10, 20, 30, 40, 50, 6]]
    assert np.array_equal(oD, np.array([[1, 0, 0, 0, 0, 0, 0, -1]], np.float32))


def test_randomised_merge_and_entry_point_cases(R):
    """A bounded run of scripts/fuzz_merge.py (merge kernel: random list counts, depths, ties, holes, NaN;
    id base, aspect-fused and plain queries on both storage modes), bit for bit against the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_merge.py"), "100", "5"],
                         capture_output=True, text=True, timeout=250)
    assert out.returncode == 0 and "all 100 cases ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_helper_workgroups_join_other_query_blocks(R):
    """scan_f32_kernel: workgroups whose query block has no runs left join the block with the most runs left
    (a list of their own in a spare output slot). Few splits and several query blocks make most workgroups
    helpers at some point; results must not depend on it."""
    X, Q = gaussian(61, (160000, 16)), gaussian(62, (700, 16))        # 3 query blocks, 1250 corpus tiles
    ix = R.FlatIPIndex(16)
    ix.add(X)
    ix.set_param("run_tiles", 4)
    S = O.scores_mfma_order(Q, X)
    for k in (10, 300):
        rD, rI = O.topk_total_order(S, k)
        for max_splits, helpers in ((3, 8), (3, 0), (20, 2), (0, 8)):
            ix.set_param("max_splits", max_splits)
            ix.set_param("f32_helpers", helpers)
            D, I = ix.search(Q, k)
            assert np.array_equal(I, rI) and np.array_equal(D, rD), (k, max_splits, helpers)
