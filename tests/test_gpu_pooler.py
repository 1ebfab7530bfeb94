"""GPU parity for the MURAL pooler entry points (ipx_pool / ipx_add_pooled / ipx_search_pooled) through
the C ABI, against tests/golden/pooler_att.npz — outputs of the reference's own MULGRAIN_Pooler
(modeling/mul_grain.py:97-132) on seeded inputs, made by tests/golden/make_golden_pooler.py — and the oracle.

Tolerance (floating point; stated here): attention weights and pooled vectors within 1e-5 relative
(+1e-6 absolute) of the reference module's outputs — the softmax and the A-term weighted sum are a few
fp32 roundings either way; searches on the pooled vectors within the north-star rule (scores 1e-4
relative, ids equal except across near-ties)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, gaussian
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from mural_amd.faiss_retriever import retriever
    from mural_amd import _lib
    assert _lib.lib().ipx_device_count() >= 1, "no HIP device visible: the search path has no CPU fallback"
    return retriever


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN_DIR, "pooler_att.npz"))


@pytest.mark.parametrize("tag", ["small", "bert", "wide"])
def test_pool_matches_reference_module(R, G, tag):
    W, b, begin = G[f"{tag}_W"], G[f"{tag}_b"], int(G[f"{tag}_begin"])
    ix = R.FlatIPIndex(W.shape[1])
    for side in "qp":
        rep, att = ix.pool(G[f"{tag}_h{side}"], W, b, begin_id=begin)
        assert np.allclose(att, G[f"{tag}_{side}_att"], rtol=1e-5, atol=1e-7)
        assert np.allclose(rep, G[f"{tag}_{side}_rep"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("storage", ["f32", "bf16scan"])
def test_encoder_output_to_ranking(R, G, storage):
    # passages and queries go from hidden states to a ranking without leaving the device
    d, A, L = 768, 3, 10
    n, nq, k = 6000, 70, 100
    W, b = G["bert_W"], G["bert_b"]
    Hp, Hq = gaussian(61, (n, L, d)), gaussian(62, (nq, L, d))
    Xp, _ = O.mural_pool_att(Hp, W, b, 1)
    Xq, att_q = O.mural_pool_att(Hq, W, b, 1)
    ix = R.FlatIPIndex(d, storage=storage)
    ix.add_pooled(Hp[:2500], W, b)
    ix.add_pooled(Hp[2500:], W, b)
    assert ix.ntotal == n
    D, I, att = ix.search_pooled(Hq, W, b, k, return_att=True)
    assert np.allclose(att, att_q, rtol=1e-5, atol=1e-7)
    assert O.check_topk(D, I, Xq, Xp, k) == []
    # and it is the same thing as pooling, then add/search on the pooled rows: bit-identical
    rep_p, _ = ix.pool(Hp, W, b)
    rep_q, _ = ix.pool(Hq, W, b)
    ix2 = R.FlatIPIndex(d, storage=storage)
    ix2.add(rep_p)
    D2, I2 = ix2.search(rep_q, k)
    assert np.array_equal(D, D2) and np.array_equal(I, I2)


def test_pooled_device_pointers_and_stream(R, G):
    import torch
    from mural_amd import _lib
    L = _lib.lib()
    d, A, Ls, n, nq, k = 64, 3, 9, 900, 33, 10
    W, b = G["small_W"], G["small_b"]
    Hp, Hq = gaussian(71, (n, Ls, d)), gaussian(72, (nq, Ls, d))
    ix = R.FlatIPIndex(d)
    dev = torch.device("cuda", 0)
    tW, tb = torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev)
    tHp, tHq = torch.from_numpy(Hp).to(dev), torch.from_numpy(Hq).to(dev)
    st = torch.cuda.Stream(device=dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    tD = torch.empty((nq, k), dtype=torch.float32, device=dev)
    tI = torch.empty((nq, k), dtype=torch.int64, device=dev)
    tA = torch.empty((nq, A), dtype=torch.float32, device=dev)
    vp = C.c_void_p
    _lib.check(L.ipx_add_pooled(ix._h, vp(tHp.data_ptr()), Ls, vp(tW.data_ptr()), vp(tb.data_ptr()), A, 1, n, 1,
                                vp(st.cuda_stream)))
    _lib.check(L.ipx_search_pooled(ix._h, vp(tHq.data_ptr()), Ls, vp(tW.data_ptr()), vp(tb.data_ptr()), A, 1, nq, k,
                                   vp(tD.data_ptr()), vp(tI.data_ptr()), vp(tA.data_ptr()), 1, vp(st.cuda_stream)))
    _lib.check(L.ipx_check(ix._h, vp(st.cuda_stream)))
    Xp, _ = O.mural_pool_att(Hp, W, b, 1)
    Xq, att = O.mural_pool_att(Hq, W, b, 1)
    assert np.allclose(tA.cpu().numpy(), att, rtol=1e-5, atol=1e-7)
    assert O.check_topk(tD.cpu().numpy(), tI.cpu().numpy(), Xq, Xp, k) == []


def test_pool_argument_errors(R, G):
    W, b = G["small_W"], G["small_b"]
    ix = R.FlatIPIndex(64)
    H = gaussian(5, (4, 3, 64))
    with pytest.raises(RuntimeError, match="aspect tokens outside"):
        ix.pool(H, W, b, begin_id=1)          # needs positions 1..3 of a 3-token sequence
    with pytest.raises(AssertionError):
        ix.pool(gaussian(5, (4, 9, 32)), W, b)
    rep, att = ix.pool(np.zeros((0, 9, 64), np.float32), W, b)
    assert rep.shape == (0, 64) and att.shape == (0, 3)
    D, I = ix.search_pooled(np.zeros((0, 9, 64), np.float32), W, b, 5)
    assert D.shape == (0, 5)
