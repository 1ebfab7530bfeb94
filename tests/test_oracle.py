"""The CPU oracle (oracle/) against exact references and the faiss semantics SURVEY.md §2a
lists. Parity with faiss itself is UNPINNED (no faiss here, no reference fixtures): these
tests pin the restatement against float64 ground truth and the documented edge behaviour."""
import numpy as np
import pytest

from conftest import gaussian
from oracle import oracle as O

FLT_MAX = np.float32(np.finfo(np.float32).max)


@pytest.mark.parametrize("n,d,nq,k", [(4096, 768, 32, 1), (4096, 768, 32, 10), (4096, 768, 32, 100),
                                      (4096, 768, 32, 1000), (3000, 100, 19, 50), (3000, 100, 20, 50),
                                      (3000, 100, 21, 50), (1500, 768, 1, 100)])
def test_flat_ip_matches_exact(n, d, nq, k):
    X, Q = gaussian(1234, (n, d)), gaussian(4321, (nq, d))
    ix = O.FlatIPOracle(d)
    ix.add(X)
    D, I = ix.search(Q, k)
    assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (nq, k)
    assert O.check_topk(D, I, Q, X, k) == []


def test_padding_when_k_exceeds_ntotal():
    X, Q = gaussian(1, (37, 64)), gaussian(2, (25, 64))
    ix = O.FlatIPOracle(64)
    ix.add(X)
    D, I = ix.search(Q, 100)
    assert (I[:, 37:] == -1).all() and (D[:, 37:] == -FLT_MAX).all()
    assert (np.sort(I[:, :37], axis=1) == np.arange(37)).all()
    assert O.check_topk(D, I, Q, X, 100) == []


def test_empty_index_and_incremental_add():
    ix = O.FlatIPOracle(16)
    D, I = ix.search(gaussian(0, (3, 16)), 5)
    assert (I == -1).all() and (D == -FLT_MAX).all()
    X = gaussian(5, (1000, 16))
    for lo in range(0, 1000, 333):
        ix.add(X[lo:lo + 333])
    assert ix.ntotal == 1000
    one = O.FlatIPOracle(16)
    one.add(X)
    Q = gaussian(6, (40, 16))
    a, b = ix.search(Q, 20), one.search(Q, 20)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_strict_admission_keeps_earlier_row_on_ties():
    # duplicated rows: exactly equal scores; the heap admits only strictly larger values, so
    # among equals contesting the last slot the earlier row stays
    base = gaussian(3, (50, 32))
    X = np.concatenate([base, base], axis=0)
    Q = gaussian(4, (24, 32))
    ix = O.FlatIPOracle(32)
    ix.add(X)
    D, I = ix.search(Q, 1)
    assert (I[:, 0] < 50).all()
    D, I = ix.search(Q, 7)
    assert O.check_topk(D, I, Q, X, 7) == []


def test_nan_and_inf_rows():
    X = gaussian(7, (200, 16))
    X[5, 0] = np.nan       # NaN score never enters the heap
    X[9, :] = 0
    X[9, 0] = np.inf       # +-inf scores (query component sign decides)
    Q = np.abs(gaussian(8, (30, 16))) + 0.1
    ix = O.FlatIPOracle(16)
    ix.add(X)
    D, I = ix.search(Q, 200)
    assert not (I == 5).any()
    assert (I[:, 0] == 9).all() and np.isinf(D[:, 0]).all()
    assert (I[:, -1] == -1).all()          # the NaN row leaves one slot unfilled
    Qn = -Q
    D, I = ix.search(Qn, 200)
    assert not (I == 9).any()              # -inf is not > -FLT_MAX: never admitted


def test_wrapper_assertions():
    ix = O.FlatIPOracle(8)
    with pytest.raises(AssertionError):
        ix.add(np.zeros((3, 9), np.float32))
    with pytest.raises(ValueError):
        ix.add(np.zeros((8,), np.float32))
    with pytest.raises(AssertionError):
        ix.search(np.zeros((3, 8), np.float32), 0)


def test_result_heap_is_topk_of_union():
    rng = np.random.default_rng(11)
    S, nq, k = 5, 12, 16
    Ds = np.sort(rng.standard_normal((S, nq, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    Is = rng.permutation(S * nq * k).reshape(S, nq, k).astype(np.int64)
    cD, cI = O.combine_results(zip(Ds, Is))
    mD, mI = O.merge_total_order(Ds, Is, k)
    assert np.array_equal(cD, mD) and np.array_equal(cI, mI)


def test_result_heap_golden_and_string_ids(golden):
    cD, cI = O.combine_results(zip(golden["D_shard_scores"], golden["D_shard_ids"]))
    assert np.array_equal(cD, golden["D_expected_scores"])
    assert np.array_equal(cI, golden["D_expected_ids"])
    with pytest.raises(ValueError):
        O.combine_results([(np.zeros((1, 2), np.float32), np.array([["B0048", "x"]]))])


def test_mfma_order_scores_close_to_exact():
    X, Q = gaussian(21, (500, 100)), gaussian(22, (9, 100))
    S = O.scores_mfma_order(Q, X)
    ref = Q.astype(np.float64) @ X.astype(np.float64).T
    assert np.abs(S - ref).max() < 1e-4
    # a feature count that is a multiple of 16 needs no padding: same chain, same bits
    X2, Q2 = gaussian(23, (64, 32)), gaussian(24, (4, 32))
    ord8 = [0, 4, 1, 5, 2, 6, 3, 7]
    acc = np.zeros((4, 64), dtype=np.float32)
    for g in range(0, 32, 8):
        for e in ord8:
            acc = np.array([[np.float32(np.fma(X2[j, g + e], Q2[q, g + e], acc[q, j])) if hasattr(np, "fma")
                             else np.float32(np.float64(X2[j, g + e]) * np.float64(Q2[q, g + e]) + np.float64(acc[q, j]))
                             for j in range(64)] for q in range(4)], dtype=np.float32)
    assert np.array_equal(acc, O.scores_mfma_order(Q2, X2))


# ---- MURAL pooler: this restatement IS pinned, by outputs of the reference's own module -------------
@pytest.mark.parametrize("tag", ["small", "bert", "wide"])
def test_mural_pool_matches_reference_module_outputs(tag):
    import os
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, "pooler_att.npz"))
    W, b, begin = g[f"{tag}_W"], g[f"{tag}_b"], int(g[f"{tag}_begin"])
    for side in "qp":
        rep, att = O.mural_pool_att(g[f"{tag}_h{side}"], W, b, begin)
        assert np.allclose(att, g[f"{tag}_{side}_att"], rtol=1e-5, atol=1e-7)
        assert np.allclose(rep, g[f"{tag}_{side}_rep"], rtol=1e-5, atol=1e-6)
        assert np.allclose(att.sum(1), 1.0, atol=1e-6)
