"""GPU parity for the small-batch fp32 scan (scan_f32_small_kernel: <= 192 queries on an fp32 index) through
the C ABI. Bar: BIT-EXACT — ids and score bits — against the oracle's kernel-order scores
(oracle.scores_mfma_order) and against the 256-query tile kernel on the same index: a score may not depend on
the batch it was computed in. Every case is run through both kernels, so the large tile keeps its coverage
for small batches too."""
import numpy as np
import pytest

from conftest import gaussian
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from mural_amd.faiss_retriever import retriever
    from mural_amd import _lib
    assert _lib.lib().ipx_device_count() >= 1, "no HIP device visible: the search path has no CPU fallback"
    return retriever


def both_kernels(ix, Q, k):
    ix.set_param("no_small_batch", 1)
    big = ix.search(Q, k)
    ix.set_param("no_small_batch", 0)
    small = ix.search(Q, k)
    assert np.array_equal(big[0], small[0]) and np.array_equal(big[1], small[1])
    return small


@pytest.mark.parametrize("n,d,nq,k", [
    (30000, 768, 1, 100), (30000, 768, 32, 100), (30000, 768, 33, 100), (30000, 768, 64, 1000),
    (30000, 768, 65, 10), (30000, 768, 150, 100), (30000, 768, 192, 1),
    (127, 16, 5, 10), (128, 16, 5, 200), (129, 16, 40, 10), (257, 24, 7, 300), (385, 100, 100, 2048),
    (1, 8, 3, 4), (70000, 17, 64, 2048),
])
def test_small_tile_equals_large_tile_and_oracle(R, n, d, nq, k):
    X, Q = gaussian(77 + n, (n, d)), gaussian(78 + nq, (nq, d))
    ix = R.FlatIPIndex(d)
    ix.add(X)
    D, I = both_kernels(ix, Q, k)
    rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k)
    assert np.array_equal(I, rI) and np.array_equal(D, rD)
    assert O.check_topk(D, I, Q, X, k) == []


def test_small_tile_long_scan_every_split_count(R):
    # few splits: each workgroup walks many tile pairs, queues fill, compaction episodes and the
    # union threshold run; S = 512 splits: the publish / poll path of short lists
    X, Q = gaussian(5, (200000, 32)), gaussian(6, (48, 32))
    ix = R.FlatIPIndex(32)
    ix.add(X)
    S = O.scores_mfma_order(Q, X)
    for max_splits in (1, 2, 7, 0):
        ix.set_param("max_splits", max_splits)
        for k in (1, 100, 700):
            D, I = ix.search(Q, k)
            rD, rI = O.topk_total_order(S, k)
            assert np.array_equal(I, rI) and np.array_equal(D, rD), (max_splits, k)
    assert ix.last_geometry()["splits"] > 7


def test_small_tile_ascending_scores_and_ties(R):
    # every new row beats the running threshold (maximum queue pressure), then exact ties
    n, d = 60000, 16
    X = np.zeros((n, d), np.float32)
    X[:, 0] = np.arange(n, dtype=np.float32)
    Q = np.zeros((5, d), np.float32)
    Q[:, 0] = np.array([1, 2, 0.5, -1, 0], np.float32)
    ix = R.FlatIPIndex(d)
    ix.add(X)
    for max_splits in (1, 0):
        ix.set_param("max_splits", max_splits)
        for k in (1, 100, 2048):
            D, I = both_kernels(ix, Q, k)
            rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k)
            assert np.array_equal(I, rI) and np.array_equal(D, rD)
    assert (I[4, :5] == np.arange(5)).all()       # all-equal scores: rows ascending
    base = gaussian(3, (300, 32))
    Xt = np.concatenate([base] * 40)              # every row 40 times
    ix2 = R.FlatIPIndex(32)
    ix2.add(Xt)
    Qt = gaussian(4, (9, 32))
    D, I = both_kernels(ix2, Qt, 100)
    rD, rI = O.topk_total_order(O.scores_mfma_order(Qt, Xt), 100)
    assert np.array_equal(I, rI) and np.array_equal(D, rD)


def test_small_tile_nan_inf_and_id_base(R):
    X = gaussian(7, (5000, 16))
    X[17, 3] = np.nan
    X[99, 0] = np.inf
    X[100, 0] = -np.inf
    Q = gaussian(8, (6, 16))
    ix = R.FlatIPIndex(16)
    ix.add(X)
    ix.set_id_base(1_000_000_000_000)
    D, I = both_kernels(ix, Q, 50)
    assert not np.isnan(D).any()
    assert ((I >= 1_000_000_000_000) | (I == -1)).all()
    assert not (I == 1_000_000_000_017).any()


def test_small_tile_configs1_scale_properties(R):
    # BASELINE configs[1] rows (1.8M x 768) at batch 16: prefix property and self-retrieval
    import torch
    dev = torch.device("cuda:0")
    n, d = 1_800_000, 768
    ix = R.FlatIPIndex(d)
    ix.reserve(n)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    keep = None
    for lo in range(0, n, 300000):
        x = torch.randn((min(300000, n - lo), d), generator=g, device=dev)
        if keep is None:
            keep = x[:16].clone()
        ix.add(x)
    D, I = ix.search_device(keep, 100)
    D10, I10 = ix.search_device(keep, 10)
    assert torch.equal(I[:, 0].cpu(), torch.arange(16))           # each row finds itself first
    assert torch.equal(D[:, :10], D10) and torch.equal(I[:, :10], I10)
    assert bool((D[:, :-1] >= D[:, 1:]).all())
    ix.set_param("no_small_batch", 1)
    Db, Ib = ix.search_device(keep, 100)
    assert torch.equal(D, Db) and torch.equal(I, Ib)
