"""GPU parity of the bf16 scan + exact fp32 rescoring modes (include/ipx.h, "storage" 1 and 2).

The bar is the fp32 path's own: storage 1 ("bf16scan") must return the SAME BITS as the fp32 scan
(scores and ids), whatever the candidate-list length, because every query either passes the
sufficiency check of rescore_kernel or is repeated on a longer list / the fp32 scan. Storage 2
("bf16") is exact on the bf16-rounded rows: bit-identical to the oracle's kernel-order scores of
<fp32 query, rounded row>.
"""
import numpy as np
import pytest

from conftest import diag_build, gaussian
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FLT_MAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def R():
    from mural_amd.faiss_retriever import retriever
    from mural_amd import _lib
    assert _lib.lib().ipx_device_count() >= 1, "no HIP device visible: the search path has no CPU fallback"
    return retriever


def bf16_round(x):
    """round-to-nearest-even fp32 -> bf16 -> fp32 (finite inputs)"""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def index(R, X, storage, adds=1):
    ix = R.FlatIPIndex(X.shape[1], storage=storage)
    step = -(-X.shape[0] // adds) if X.shape[0] else 1
    for lo in range(0, X.shape[0], step):
        ix.add(X[lo:lo + step])
    assert ix.ntotal == X.shape[0]
    return ix


def assert_bits(D, I, Q, X, k):
    rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k)
    assert np.array_equal(I, rI), f"ids differ at {np.argwhere(I != rI)[:3].tolist()}"
    assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), "score bits differ"


@pytest.mark.parametrize("n,d,nq,k,adds", [
    (4096, 768, 32, 1, 1), (4096, 768, 32, 100, 1), (5000, 768, 300, 100, 2),
    (20000, 100, 513, 100, 3),        # d not a multiple of 32, three adds, three query blocks
    (9000, 17, 64, 512, 2),           # 4k = 2048 candidates: the longest list
    (9000, 48, 64, 1000, 1),          # hi + lo query parts, 2000 candidates
    (9000, 48, 64, 1500, 1),          # 3000 candidates (lists up to 4096 entries: depths up to 2048)
    (9000, 48, 300, 2048, 2),         # the deepest search the bf16 scan serves, 4096 candidates, large batch
    (777, 768, 1, 100, 1), (255, 16, 257, 5, 1), (257, 16, 256, 300, 1), (1, 8, 3, 4, 1),
])
def test_bf16scan_is_bit_identical_to_the_fp32_scan(R, n, d, nq, k, adds):
    X, Q = gaussian(1234, (n, d)), gaussian(4321, (nq, d))
    ix = index(R, X, "bf16scan", adds)
    D, I = ix.search(Q, k)
    assert D.dtype == np.float32 and I.dtype == np.int64 and D.shape == (nq, k) == I.shape
    assert_bits(D, I, Q, X, k)
    st = ix.last_bf16_stats()
    assert st["candidates"] > 0                        # every depth up to IPX_MAX_K takes the bf16 scan
    # same index, fp32 scan forced: the two scans of one index agree bit for bit
    ix.set_param("f32_scan", 1)
    D2, I2 = ix.search(Q, k)
    assert np.array_equal(I, I2) and np.array_equal(D.view(np.uint32), D2.view(np.uint32))
    assert ix.last_bf16_stats()["candidates"] == 0


@pytest.mark.parametrize("n,d,nq,k", [(4096, 768, 32, 10), (30000, 768, 300, 100), (20000, 100, 513, 100),
                                      (9000, 17, 64, 1000), (300, 64, 20, 500)])
def test_bf16_rows_exact_on_the_stored_values(R, n, d, nq, k):
    X, Q = gaussian(11, (n, d)), gaussian(12, (nq, d))
    Xr = bf16_round(X)
    ix = index(R, X, "bf16")
    D, I = ix.search(Q, k)
    assert_bits(D, I, Q, Xr, k)
    assert O.check_topk(D, I, Q, Xr, k) == []
    # rows that are already bf16 (as float16-free torch tensors) are kept exactly
    import torch
    t = torch.from_numpy(Xr).to("cuda:0").to(torch.bfloat16)
    ix2 = R.FlatIPIndex(d, storage="bf16")
    ix2.add(t)
    D2, I2 = ix2.search(Q, k)
    assert np.array_equal(I, I2) and np.array_equal(D.view(np.uint32), D2.view(np.uint32))


@pytest.mark.parametrize("parts", [1, 2])
@pytest.mark.parametrize("n,d,nq,k", [(30000, 768, 300, 100), (9000, 100, 64, 300), (50000, 64, 20, 1000)])
def test_query_parts_do_not_change_results(R, parts, n, d, nq, k):
    X, Q = gaussian(21, (n, d)), gaussian(22, (nq, d))
    for storage, ref in (("bf16scan", X), ("bf16", bf16_round(X))):
        ix = index(R, X, storage)
        ix.set_param("query_parts", parts)
        D, I = ix.search(Q, k)
        assert_bits(D, I, Q, ref, k)


def test_short_candidate_lists_are_caught_and_repeated(R):
    # many near-ties: rows are small perturbations of one vector, so far more than kc rows sit
    # inside the bf16 error margin of the k-th score. With the shortest list every query fails
    # the check; the answer must not change.
    rng = np.random.default_rng(5)
    d, n, nq, k = 64, 30000, 40, 10
    base = rng.standard_normal(d).astype(np.float32)
    X = (base[None, :] + 1e-3 * rng.standard_normal((n, d))).astype(np.float32)
    Q = (base[None, :] + 0.1 * rng.standard_normal((nq, d))).astype(np.float32)
    ix = index(R, X, "bf16scan")
    ix.set_param("bf16_candidates", 16)
    D, I = ix.search(Q, k)
    assert_bits(D, I, Q, X, k)
    assert ix.last_bf16_stats()["fallback_queries"] == nq
    # bf16-only storage: the longest list is tried, then the exact pass over every stored row
    ix2 = index(R, X, "bf16")
    ix2.set_param("bf16_candidates", 16)
    D2, I2 = ix2.search(Q, k)
    assert_bits(D2, I2, Q, bf16_round(X), k)


def test_zero_query_and_mass_ties_do_not_fail_the_batch(R):
    """A zero embedding (every score 0: as many exact ties as rows) and thousands of duplicate rows at the top of
    a ranking, next to ordinary queries in one batch — every storage returns the one total order (score desc, row
    asc), the same bits as the fp32 scan."""
    rng = np.random.default_rng(9)
    d, n, k = 64, 9000, 50
    X = gaussian(31, (n, d))
    dup = gaussian(32, (1, d))
    X[2000:5000] = dup                                   # 3000 identical rows: more than the longest candidate list
    Q = gaussian(33, (70, d))
    Q[3] = 0.0                                           # the zero query
    Q[7] = dup[0] * 2.0                                  # the duplicates are its best hits, all tied
    Q[64] = 0.0
    for storage, ref in (("f32", X), ("bf16scan", X), ("bf16", bf16_round(X))):
        ix = index(R, X, storage)
        for nq in (70, 8):                               # large-batch and small-batch kernels
            D, I = ix.search(Q[:nq], k)
            assert_bits(D, I, Q[:nq], ref, k)
            assert (D[3] == 0).all() and (I[3] == np.arange(k)).all()
            assert (I[7] == 2000 + np.arange(k)).all()
        D, I = ix.search(Q[:8], 1500)                    # deep list on tie masses
        assert_bits(D, I, Q[:8], ref, 1500)


def test_bf16_edge_cases(R):
    # k > ntotal, empty index, ties, NaN / inf rows (they poison the norm bound: everything falls back)
    X, Q = gaussian(1, (37, 64)), gaussian(2, (25, 64))
    for storage, ref in (("bf16scan", X), ("bf16", bf16_round(X))):
        ix = index(R, X, storage)
        D, I = ix.search(Q, 100)
        assert (I[:, 37:] == -1).all() and (D[:, 37:] == -FLT_MAX).all()
        assert_bits(D, I, Q, ref, 100)
        e = R.FlatIPIndex(64, storage=storage)
        D, I = e.search(Q, 7)
        assert (I == -1).all() and (D == -FLT_MAX).all()
    base = gaussian(3, (300, 32))
    X = np.concatenate([base, base, base[:50]], axis=0)
    Q = gaussian(4, (40, 32))
    ix = index(R, X, "bf16scan")
    D, I = ix.search(Q, 9)
    assert_bits(D, I, Q, X, 9)
    assert (I[:, 0] < 300).all() and np.array_equal(D[:, 0], D[:, 1])
    X = gaussian(7, (5000, 16))
    X[5, 0] = np.nan
    X[11, 3] = np.inf
    X[12, 3] = -np.inf
    Q = gaussian(8, (30, 16))
    ix = index(R, X, "bf16scan")
    D, I = ix.search(Q, 20)
    assert_bits(D, I, Q, X, 20)
    assert not (I == 5).any()


def test_bf16scan_aspect_queries_and_float16_rows(R):
    n, d, nq, A = 5000, 768, 70, 3
    X = gaussian(51, (n, d)).astype(np.float16)
    E = gaussian(52, (nq, A, d))
    logits = gaussian(53, (nq, A))
    a = R.FlatIPIndex(d, storage="f32")
    a.add(X)
    b = R.FlatIPIndex(d, storage="bf16scan")
    b.add(X)
    for kw in ({"logits": True},):
        D0, I0 = a.search_aspect(E, logits, 100, **kw)
        D1, I1 = b.search_aspect(E, logits, 100, **kw)
        assert b.last_bf16_stats()["candidates"] == 256
        assert np.array_equal(I0, I1) and np.array_equal(D0.view(np.uint32), D1.view(np.uint32))


def test_depth_limits_of_the_bf16_storage(R):
    from mural_amd._lib import IpxError
    X, Q = gaussian(31, (20000, 32)), gaussian(32, (10, 32))
    ix = index(R, X, "bf16")
    for k in (1638, 1639, 2048, 2049, 8192):       # lists beyond 4096 entries are left unsorted by the scan (deep lists)
        D, I = ix.search(Q, k)
        assert_bits(D, I, Q, bf16_round(X), k)
    with pytest.raises(IpxError) as e:
        ix.search(Q, 8193)
    assert e.value.code == 1 and "8192" in str(e.value)
    # very long rows: the rescoring kernel's LDS holds the query and a list of at most 4096 entries
    Xl = gaussian(33, (300, 4096))
    il = index(R, Xl, "bf16")
    D, I = il.search(gaussian(34, (3, 4096)), 1638)
    assert (I[:, 300:] == -1).all()
    with pytest.raises(IpxError) as e:
        il.search(gaussian(34, (3, 4096)), 1639)
    assert e.value.code == 1 and "1638" in str(e.value)


@pytest.mark.parametrize("storage", ["bf16scan", "bf16"])
@pytest.mark.parametrize("n,d,nq,k", [(30000, 64, 70, 3000), (30000, 64, 300, 5000), (20000, 48, 9, 8192), (3000, 32, 70, 5000)])
def test_deep_candidate_lists_on_the_bf16_scan(R, storage, n, d, nq, k):
    """Depths beyond IPX_MAX_K from the bf16 scan: candidate lists of up to 16384 entries, kept unsorted per split and
    selected + sorted across the splits by deep_finish_kernel, then rescored exactly — the same bits as the oracle
    (and as the fp32 scan's deep one-pass lists), with one and two query parts, several splits and a single one."""
    X, Q = gaussian(91, (n, d)), gaussian(92, (nq, d))
    ref = bf16_round(X) if storage == "bf16" else X
    rD, rI = O.topk_total_order(O.scores_mfma_order(Q, ref), k)
    ix = index(R, X, storage, adds=2)
    for prm in ({}, {"query_parts": 2}, {"max_splits": 1}, {"max_splits": 3, "bf16_candidates": k + 7}):
        for key, val in {"query_parts": 0, "max_splits": 0, "bf16_candidates": 0, **prm}.items():
            ix.set_param(key, val)
        D, I = ix.search(Q, k)
        st = ix.last_bf16_stats()
        assert st["candidates"] >= k, (prm, st)
        assert np.array_equal(I, rI), (prm, st, np.argwhere(I != rI)[:3].tolist())
        assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), prm


def test_storage_can_only_be_chosen_on_an_empty_index(R):
    from mural_amd._lib import IpxError
    ix = R.FlatIPIndex(16)
    ix.add(gaussian(1, (10, 16)))
    with pytest.raises(IpxError):
        ix.set_param("storage", 1)
    with pytest.raises(IpxError):
        R.FlatIPIndex(16, storage="bf16").set_param("f32_scan", 1)


def test_config2_scale_bf16scan_equals_fp32_scan(R):
    """BASELINE.json configs[1] size (1.8M x 768, 8192 queries, top-100): both scans of one
    storage-1 index return the same bits; storage 2 agrees with a float64 recomputation."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    n, d, nq, k = 1_800_000, 768, 8192, 100
    ix = R.FlatIPIndex(d, device=0, storage="bf16scan")
    ix.reserve(n)
    for lo in range(0, n, 300_000):
        ix.add(torch.randn((300_000, d), generator=g, device=dev, dtype=torch.float32))
    g.manual_seed(4321)
    Q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)
    D1, I1 = ix.search_device(Q, k)
    st = ix.last_bf16_stats()
    assert st["candidates"] == 256 and st["fallback_queries"] < nq // 50
    ix.set_param("f32_scan", 1)
    D0, I0 = ix.search_device(Q, k)
    ix.check()
    assert torch.equal(I0, I1) and torch.equal(D0.view(torch.int32), D1.view(torch.int32))


def test_margin_rule_against_coherent_rounding_errors(R):
    """Rounding errors that all pull the same way — the worst case the margin rule has to cover, with bf16's
    unit roundoff 2^-8 per operand. The query rounds up on dims 0..31 and down on dims 32..63.
      U rows (100): on dims 0..31, rounding up:   exact 32.25, approximate 32.50 (+0.25)
      D rows (10):  on dims 32..63, rounding down: exact 32.30..32.34, approximate 32.05..32.09 (-0.25)
      fillers (3000): exactly representable, exact 32.227, approximate 32.10
    The exact top-10 are the D rows, but the approximate order is U > fillers > D: a candidate list of 256
    ends among the fillers, 0.40 below the 10th best approximate score. A rule that assumed less than the
    true error bound (2e >= 0.50 here) would trust that list and return U rows."""
    d, k = 64, 10
    up = np.float32(1 + 2.0 ** -8 + 2.0 ** -20)        # -> 1 + 2^-7 in bf16
    down = np.float32(1 + 2.0 ** -8 - 2.0 ** -20)      # -> 1.0
    q = np.concatenate([np.full(32, up), np.full(32, down)]).astype(np.float32)
    U = np.zeros((100, d), np.float32)
    U[:, :32] = up
    U[:, 0] += np.arange(100, dtype=np.float32) * np.float32(2.0 ** -18)       # distinct scores, still rounding up
    Dn = np.zeros((10, d), np.float32)
    Dn[:, 32:63] = np.float32(1.0117)                   # just below the midpoint 1.01171875 -> 1.0078125
    Dn[:, 63] = np.float32(0.8125) + np.arange(10, dtype=np.float32) * np.float32(2.0 ** -8)   # exact in bf16
    F = np.zeros((3000, d), np.float32)
    F[:, 32:] = 1.0
    F[:, 63] = np.float32(1.1015625)
    assert np.array_equal(bf16_round(F), F) and np.array_equal(bf16_round(Dn[:, 63]), Dn[:, 63])
    rng = np.random.default_rng(3)
    perm = rng.permutation(3110)
    X = np.concatenate([U, Dn, F])[perm]
    Q = np.stack([q, q])
    exact = X.astype(np.float64) @ q.astype(np.float64)
    approx = bf16_round(X).astype(np.float64) @ bf16_round(q).astype(np.float64)
    is_d = np.isin(perm, np.arange(100, 110))
    assert set(np.argsort(-exact)[:k]) == set(np.flatnonzero(is_d))            # the construction holds:
    a_k = np.sort(approx)[::-1][k - 1]
    assert 0.36 < a_k - approx[is_d].max() < 0.50 and np.abs(approx - exact).max() > 0.25
    assert (approx > approx[is_d].max()).sum() > 256                           # the 256-list ends above the D rows
    ix = index(R, X, "bf16scan")
    ix.set_param("query_parts", 1)
    ix.set_param("bf16_candidates", 256)
    D, I = ix.search(Q, k)
    assert ix.last_bf16_stats()["fallback_queries"] == 2                       # caught, repeated
    assert_bits(D, I, Q, X, k)
    assert set(I[0].tolist()) == set(np.flatnonzero(is_d).tolist())
    ix.set_param("query_parts", 0)
    ix.set_param("bf16_candidates", 0)
    D, I = ix.search(Q, k)
    assert_bits(D, I, Q, X, k)


@pytest.mark.parametrize("n,d,nq,k,storage", [(60000, 768, 300, 100, "bf16scan"), (60000, 768, 40, 100, "bf16scan"),
                                              (40000, 128, 200, 300, "bf16scan"), (60000, 768, 300, 100, "bf16")])
def test_after_the_fact_margin_test_accepts_shorter_lists_with_the_same_bits(R, n, d, nq, k, storage):
    """rescore_kernel's second test (list reaches e below the exact k-th score of the candidates) passes whenever the
    a-priori one (2e below the k-th approximate score) does, so with the same list it never repeats more queries;
    the shorter default lists it allows return the same bits as the old ones and as the oracle."""
    X, Q = gaussian(77, (n, d)), gaussian(78, (nq, d))
    ix = index(R, X, storage)
    res = {}
    for post in (0, 1):
        ix.set_param("margin_post", post)
        D, I = ix.search(Q, k)
        res[post] = (D, I, ix.last_bf16_stats())
        if storage == "bf16scan":
            assert_bits(D, I, Q, X, k)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    assert 0 < res[1][2]["candidates"] <= res[0][2]["candidates"]
    # one forced list length, both rules: the second chance can only take queries off the repeat pass
    kc = max(128, k + k // 4)
    ix.set_param("bf16_candidates", kc)
    rep = {}
    for post in (0, 1):
        ix.set_param("margin_post", post)
        D, I = ix.search(Q, k)
        rep[post] = ix.last_bf16_stats()["fallback_queries"]
        assert np.array_equal(I, res[0][1]) and np.array_equal(D.view(np.uint32), res[0][0].view(np.uint32))
    assert rep[1] <= rep[0]


def _with_outliers(X, rows, rng, special=True):
    X = X.copy()
    for j, r in enumerate(rows):
        X[r] *= np.float32(3.0 + (j % 7))                     # 3x .. 9x the ordinary norm
    if special and len(rows) >= 4:
        X[rows[0], 1] = np.nan
        X[rows[1], 0] = np.inf
        X[rows[2], 2] = -np.inf
        X[rows[3]] *= np.float32(1e30)                        # products overflow to +-inf
    return X


@pytest.mark.parametrize("n,d,nq,k,adds,nout", [
    (30000, 768, 300, 100, 1, 12),     # a few long rows + NaN / inf rows, large batch (16-wave kernel)
    (30000, 768, 32, 100, 3, 12),      # small batch (64-query tile, hi + lo parts), outliers arrive over three adds
    (20000, 100, 513, 10, 2, 40),
    (50000, 64, 70, 1000, 1, 9),       # deep lists
    (9000, 48, 64, 1500, 1, 9),        # 3000 candidates
    (20000, 96, 260, 50, 4, 300),      # more outliers than the side list holds: the rest count as ordinary rows
    (600, 32, 5, 700, 1, 20),          # k > n
])
def test_outlier_rows_are_scored_on_the_side(R, n, d, nq, k, adds, nout):
    """Storage 1: rows far longer than the rest (and NaN / inf rows) stay out of the bf16 scan and of the margin rule's
    norm bounds; every search scores them exactly and merges them in. Same bits as the oracle and as the fp32 scan,
    and — the point — no query is sent to the repeat pass or the fp32 scan because of them."""
    rng = np.random.default_rng(77)
    rows = np.sort(rng.choice(np.arange(2100, n) if n > 4000 else np.arange(n // 2, n), size=nout, replace=False))
    X = _with_outliers(gaussian(1234, (n, d)), rows, rng)
    Q = gaussian(4321, (nq, d))
    Q[0] = 0                                                  # a zero query: NaN scores for the inf rows
    ix = index(R, X, "bf16scan", adds)
    base = 1000
    ix.set_id_base(base)
    D, I = ix.search(Q, k)
    with np.errstate(all="ignore"):
        rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k, id_base=base)
    assert np.array_equal(I, rI), f"ids differ at {np.argwhere(I != rI)[:3].tolist()}"
    assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), "score bits differ"
    st = ix.last_bf16_stats()
    if n > 4000:                        # (the small index takes its scale from rows that include the long ones)
        assert st["candidates"] > 0 and st["outlier_rows"] == min(nout, 256)
        if nout <= 256 and k <= 100:
            assert st["fallback_queries"] == 0
    ix.set_param("f32_scan", 1)
    D2, I2 = ix.search(Q, k)
    assert np.array_equal(I, I2) and np.array_equal(D.view(np.uint32), D2.view(np.uint32))
    ix.set_param("f32_scan", 0)
    # every large-batch scan kernel skips the NaN copies the same way
    if nq > 64:
        variants = [{"query_parts": 2}, {"no_small_batch": 1, "max_splits": 1}]
        if diag_build():
            variants += [{"bf16_waves": 8}, {"bf16_waves": 12}, {"bf16_pp": 1}, {"bf16_pp": 3}, {"bf16_shape16": 0}]
        for prm in variants:
            for key, val in prm.items():
                ix.set_param(key, val)
            D3, I3 = ix.search(Q, k)
            assert ix.last_bf16_stats()["candidates"] > 0
            assert np.array_equal(I, I3) and np.array_equal(D.view(np.uint32), D3.view(np.uint32)), prm
            for key in prm:
                ix.set_param(key, {"bf16_shape16": 1}.get(key, 0))


def test_index_remembers_a_depth_the_bf16_scan_cannot_serve(R):
    """More long rows than the side list holds: the rest widen every margin, a deep search fails its checks for most
    of the batch, goes to the fp32 scan — and the next search at that depth goes there directly; shallower searches
    and a grown index try the bf16 scan again."""
    n, d, nq, k = 20000, 96, 128, 700
    rng = np.random.default_rng(3)
    X = gaussian(8, (n, d))
    # (300 x the ordinary norm: the repeat pass with the longest list, a fifth of the rows, fails as well)
    X[rng.choice(np.arange(3000, n), size=400, replace=False)] *= np.float32(300.0)
    Q = gaussian(9, (nq, d))
    ix = index(R, X, "bf16scan")
    rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k)
    D, I = ix.search(Q, k)
    st = ix.last_bf16_stats()
    assert st["outlier_rows"] == 256 and st["candidates"] > 0 and st["fallback_queries"] > nq // 2
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    D, I = ix.search(Q, k)
    assert ix.last_bf16_stats()["candidates"] == 0            # straight to the fp32 scan
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    D, I = ix.search(Q, 10)
    assert ix.last_bf16_stats()["candidates"] > 0             # shallower: the bf16 scan again
    assert np.array_equal(I, rI[:, :10]) and np.array_equal(D.view(np.uint32), rD[:, :10].view(np.uint32))
    ix.add(gaussian(10, (100, d)))
    ix.search(Q, k)
    assert ix.last_bf16_stats()["candidates"] > 0             # new rows: the memory is cleared


def test_outlier_threshold_comes_from_the_first_rows(R):
    """The scale is set by the first rows added; an index whose first rows are all zero (or NaN) lists only NaN / inf
    rows; an index of uniformly long rows lists none."""
    n, d, nq, k = 8000, 64, 100, 20
    Q = gaussian(5, (nq, d))
    X = gaussian(6, (n, d)) * np.float32(50.0)                # long, but uniformly so
    ix = index(R, X, "bf16scan", 2)
    D, I = ix.search(Q, k)
    assert ix.last_bf16_stats()["outlier_rows"] == 0
    assert_bits(D, I, Q, X, k)
    X = gaussian(6, (n, d))
    X[:3000] = 0
    X[10, 0] = np.nan
    X[5000, 3] = np.inf
    ix = index(R, X, "bf16scan", 1)
    D, I = ix.search(Q, k)
    assert ix.last_bf16_stats()["outlier_rows"] == 2
    with np.errstate(all="ignore"):
        rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))


@pytest.mark.parametrize("storage", ["bf16scan", "bf16"])
@pytest.mark.parametrize("nq", [32, 300])
def test_short_rows_many_runs_per_workgroup(R, storage, nq):
    """d <= 32: one K-stage per corpus tile, so the stage ring looks ahead further than a 4-tile run is long.
    (Regression: the look-ahead cursor once needed a run that was only handed out when the computing cursor
    got there, and the workgroup stopped after its first runs — found by scripts/fuzz_parity.py.)"""
    X, Q = gaussian(41, (58317, 7)), gaussian(42, (nq, 7))
    Xe = bf16_round(X) if storage == "bf16" else X
    ix = index(R, X, storage, adds=2)
    for max_splits in (1, 5, 0):
        ix.set_param("max_splits", max_splits)
        for parts in (1, 2):
            ix.set_param("query_parts", parts)
            D, I = ix.search(Q, 100)
            assert ix.last_bf16_stats()["candidates"] > 0
            assert_bits(D, I, Q, Xe, 100)


@pytest.mark.parametrize("n,d,nq,k", [(60000, 768, 300, 100), (40000, 192, 700, 10), (50000, 64, 257, 100),
                                      (30000, 100, 260, 100), (70000, 384, 70, 256)])
def test_large_batch_scan_variants_return_the_same_bits(R, n, d, nq, k):
    """Every large-batch bf16 scan the loaded library carries on one index: the same bits as the oracle. The default
    library has the 16-wave kernel with one and with two query parts; a diagnostic build (-DIPX_DIAG) adds both MFMA
    shapes and issue orders, 8 waves, the ping-pong kernel, the direct-corpus-operand kernel, ring depths, workgroup
    mappings. Split caps 1 / 2 make a workgroup scan many tiles: compaction episodes, pipeline restarts."""
    X, Q = gaussian(61, (n, d)), gaussian(62, (nq, d))
    for storage, ref in (("bf16scan", X), ("bf16", bf16_round(X))):
        rD, rI = O.topk_total_order(O.scores_mfma_order(Q, ref), k)      # the oracle once per index, not per variant
        ix = index(R, X, storage, adds=2)
        ix.set_param("no_small_batch", 1)
        ix.set_param("query_parts", 1)
        for ms in (2, 0, 1):
            ix.set_param("max_splits", ms)
            variants = [{}, {"query_parts": 2}]
            if diag_build():
                variants += [{"bf16_ring": 3}, {"bf16_ring": 4, "bf16_map": 2}, {"bf16_ring": 2, "bf16_map": 7}, {"bf16_ring": 2, "bf16_occ": 2},
                             {"bf16_shape16": 0}, {"bf16_shape16": 0, "bf16_variant": 2}, {"bf16_shape16": 0, "bf16_variant": 3},
                             {"bf16_shape16": 0, "bf16_variant": 5, "bf16_map": 1}, {"bf16_waves": 8}, {"bf16_waves": 12}, {"bf16_waves": 12, "bf16_ring": 3}, {"bf16_pp": 1},
                             {"bf16_pp": 2, "bf16_map": 1}, {"bf16_pp": 3, "bf16_map": 3}, {"bf16_pp": 4}]
            for prm in variants:
                for key, val in {"bf16_shape16": 1, "bf16_variant": 0, "bf16_map": 0, "bf16_waves": 0, "bf16_pp": 0, "bf16_ring": 0,
                                 "bf16_occ": 1, "query_parts": 1, **prm}.items():
                    ix.set_param(key, val)
                D, I = ix.search(Q, k)
                assert ix.last_bf16_stats()["candidates"] > 0
                assert np.array_equal(I, rI), (ms, prm, np.argwhere(I != rI)[:3].tolist())
                assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), (ms, prm)


def _ramp(n, d, base=0.0, sign=1.0):
    X = np.zeros((n, d), np.float32)
    X[:, 0] = (np.float32(base) + np.arange(n, dtype=np.float32)) * np.float32(sign)
    return X


@pytest.mark.parametrize("storage", ["bf16scan", "bf16"])
def test_inf_row_outside_the_side_list_is_not_lost(R, storage):
    """Regression for the one failure of round 2's fuzz runs (seed 7, case 114: 2122 ramp rows, 65 queries, hi + lo
    parts, a list longer than the index). Cause: a row with an inf element that the side list of outlier rows did
    not take (more than 256 rows qualified; which ones get a slot is decided by an atomic counter, hence
    'intermittent') is scanned like any other row, and with hi + lo query parts of opposite sign its approximate
    score is inf * hi + inf * lo = NaN — never admitted — while its exact score is +inf, the best hit. The list was
    not full, 'holds every admissible row' was taken as proof, and the top hit went missing. Here the order of the
    adds makes it deterministic: the side list is full before the inf row arrives. Storage 2 has no side list at
    all: any inf row does it there."""
    n, d, nq = 2122, 192, 65
    X = _ramp(n, d)
    X[1500, 70] = np.inf
    X[1600, 62] = np.nan
    X[1700] *= np.float32(1e30)
    Q = gaussian(4321, (nq, d))
    hi = bf16_round(Q[:, 70])
    lo = bf16_round(Q[:, 70] - hi)
    assert ((Q[:, 70] > 0) & (lo < 0)).sum() >= 5             # queries whose parts cancel on the inf element
    ix = R.FlatIPIndex(d, storage=storage)
    ix.add(X[:425])                                           # sets the outlier scale: |x| <= 424
    ix.add(X[425:1025])                                       # 534 rows beyond it: the 256 slots are taken
    ix.add(X[1025:])                                          # the inf / NaN / huge rows count as ordinary rows
    ref = bf16_round(X) if storage == "bf16" else X
    for k in (100, 1024, 1638) + ((2048,) if storage == "bf16scan" else ()):
        with np.errstate(all="ignore"):
            rD, rI = O.topk_total_order(O.scores_mfma_order(Q, ref), k)
        assert (rI[:, 0] == 1500).sum() >= 5                  # the inf row is the best hit of those queries
        for parts in (0, 1, 2):
            for q in (Q, Q[:9]):                              # large-batch and small-batch kernels
                ix.set_param("query_parts", parts)            # (also clears the index's memory of failed depths)
                D, I = ix.search(q, k)
                m = len(q)
                assert np.array_equal(I, rI[:m]), (k, parts, m, np.argwhere(I != rI[:m])[:3].tolist())
                assert np.array_equal(D.view(np.uint32), rD[:m].view(np.uint32)), (k, parts, m)
                st = ix.last_bf16_stats()
                assert st["candidates"] > 0 and st["exact_queries"] == m      # no bf16 list is trusted here


@pytest.mark.parametrize("sign", [1.0, -1.0])
@pytest.mark.parametrize("n", [2122, 20000])
def test_repeat_pass_with_the_longest_list_on_ramp_data(R, n, sign):
    """The repeat pass every flagged query takes — 4096 candidates, hi + lo query parts, the 16-wave kernel — driven
    deterministically on more than one query block's worth of one 64-query group (65 queries: a full group and a
    ragged one) over ramp data, whose bf16 copy ties hundreds of rows per value: the first pass is forced to lists of
    k entries, which cannot pass the margin test, so every query is repeated; with 2122 rows the long list holds the
    whole index, with 20000 it has to pass the margin test on its own. Depths up to the deepest the bf16 scan
    serves."""
    d, nq = 192, 65
    X = _ramp(n, d, base=40000.0, sign=sign)                  # norms within a factor 1.5: no outlier rows
    rng = np.random.default_rng(11)
    Q = (0.02 * rng.standard_normal((nq, d))).astype(np.float32)
    Q[:, 0] = (1.0 + rng.random(nq) * 2.0) * np.where(rng.random(nq) < 0.5, -1.0, 1.0)
    for storage, ref in (("bf16scan", X), ("bf16", bf16_round(X))):
        ix = index(R, X, storage, adds=3)
        ix.set_param("bf16_candidates", 16)                   # first pass: lists of max(16, k) = k entries
        for k in (100, 1024, 1638) + ((2048,) if storage == "bf16scan" else ()):
            rD, rI = O.topk_total_order(O.scores_mfma_order(Q, ref), k)
            for ms in (0, 5):
                ix.set_param("max_splits", ms)
                D, I = ix.search(Q, k)
                assert np.array_equal(I, rI), (storage, k, ms, np.argwhere(I != rI)[:3].tolist())
                assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), (storage, k, ms)
                st = ix.last_bf16_stats()
                assert st["candidates"] == k and st["fallback_queries"] == nq          # all repeated ...
                assert st["exact_queries"] == 0, (storage, k, ms, st)                  # ... and served by the long list
        assert ix.last_bf16_stats()["outlier_rows"] == 0


def test_norm_spread_rows_teach_the_index_a_longer_list(R):
    """Rows whose norms spread smoothly (log-normal, sigma 1: no clear outliers, thousands of rows beyond the side
    list's 256 slots) widen every query's margin: the automatic list fails for much of a large batch, the repeat pass
    serves it — and the index starts the next search with a longer list (x1.25 per lesson) instead of repeating again.
    Same bits as the fp32 scan of the same index every time (that scan is checked against the oracle elsewhere);
    unit-norm rows and mildly spread ones need no lesson."""
    n, d, nq, k = 200_000, 768, 300, 300
    rng = np.random.default_rng(17)
    Q = gaussian(18, (nq, d))
    G = gaussian(19, (n, d))
    for sigma, learns in ((1.0, True), (0.2, False), (None, False)):
        X = G / np.linalg.norm(G, axis=1, keepdims=True) if sigma is None else \
            (G * np.exp(sigma * rng.standard_normal((n, 1)))).astype(np.float32)
        X = np.ascontiguousarray(X, dtype=np.float32)
        ix = index(R, X, "bf16scan", adds=2)
        ix.set_param("f32_scan", 1)
        rD, rI = ix.search(Q, k)
        ix.set_param("f32_scan", 0)
        seen = []
        for it in range(12):
            D, I = ix.search(Q, k)
            assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32)), (sigma, it)
            st = ix.last_bf16_stats()
            seen.append((st["candidates"], st["fallback_queries"]))
        if learns:
            assert seen[0][1] * 100 > nq and seen[1][0] > seen[0][0], seen      # repeats, then a longer list
            assert seen[-1][1] * 100 <= nq and seen[-1][0] >= seen[1][0], seen   # the lessons end the repeats
        else:
            assert all(c == seen[0][0] and f == 0 for c, f in seen), seen
        ix.add(X[:10])                                                           # new rows: lessons forgotten
        ix.search(Q, k)
        assert ix.last_bf16_stats()["candidates"] == seen[0][0]


def test_forced_candidate_count_does_not_lift_depth_limits(R):
    """`bf16_candidates` changes the list length, never which depths the bf16 scan serves (it once bypassed both
    checks: a depth beyond the longest list read past the list)."""
    from mural_amd._lib import IpxError
    X, Q = gaussian(31, (20000, 32)), gaussian(32, (70, 32))
    ix = index(R, X, "bf16scan")
    ix.set_param("bf16_candidates", 4096)
    D, I = ix.search(Q, 9000)
    assert ix.last_bf16_stats()["candidates"] == 0            # beyond the bf16 scan's depths: the fp32 scan
    assert_bits(D, I, Q, X, 9000)
    D, I = ix.search(Q, 5000)
    assert ix.last_bf16_stats()["candidates"] == 5000         # a list is never shorter than the depth
    assert_bits(D, I, Q, X, 5000)
    D, I = ix.search(Q, 2048)
    assert ix.last_bf16_stats()["candidates"] == 4096
    assert_bits(D, I, Q, X, 2048)
    ix2 = index(R, X, "bf16")
    ix2.set_param("bf16_candidates", 4096)
    with pytest.raises(IpxError) as e:
        ix2.search(Q, 9000)
    assert e.value.code == 1
    D, I = ix2.search(Q, 1638)
    assert_bits(D, I, Q, bf16_round(X), 1638)


def test_randomised_differential_cases(R):
    """A bounded run of scripts/fuzz_parity.py (random shapes, depths, split caps, storage modes, tie-heavy data,
    every scan variant against the oracle, bit for bit)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_parity.py"), "120", "7"],
                         capture_output=True, text=True, timeout=250)
    assert out.returncode == 0 and "all 120 cases ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
