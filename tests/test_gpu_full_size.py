"""GPU tests at BASELINE.json's full sizes, through size-independent properties (the oracle cannot run
these in seconds): the headline corpus (configs[2]: 8.8M x 768 fp32 = 27 GB, 6980 queries) and the
per-GPU shard of configs[4] (12.5M x 768 bf16 = 19.2 GB, batches of 1..64). Properties: planted rows are
found at rank 0 with the exact score bits of a float64-checked recomputation; lists are sorted; ids are in
range and distinct; top-10 is the prefix of top-100; the small-batch kernels, the large tiles and the bf16
scan + fp32 rescoring return the same bits; a float64 recomputation of a query's full score row finds
nothing above the returned k-th score."""
import pytest

from conftest import diag_build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    from mural_amd.faiss_retriever import retriever
    from mural_amd import _lib
    assert _lib.lib().ipx_device_count() >= 1, "no HIP device visible: the search path has no CPU fallback"
    return retriever


def _blocks(torch, dev, n, d, block, seed0, dtype=None):
    for b, lo in enumerate(range(0, n, block)):
        g = torch.Generator(device=dev)
        g.manual_seed(seed0 + b)
        x = torch.randn((min(block, n - lo), d), generator=g, device=dev, dtype=torch.float32)
        yield lo, (x.to(dtype) if dtype is not None else x)


def test_headline_corpus_8p8m_fp32(R):
    import torch
    dev = torch.device("cuda:0")
    n, d, nq, block = 8_800_000, 768, 6980, 400_000
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    Q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)
    plant_q = torch.arange(0, nq, 97, device=dev)                     # 72 queries get a planted best row
    plant_row = (plant_q * 1213 + 5) % n
    ix = R.FlatIPIndex(d, storage="bf16scan")                          # fp32 rows + bf16 copy: every scan on one index
    ix.reserve(n)
    keep = {}
    for lo, x in _blocks(torch, dev, n, d, block, 1234):
        m = (plant_row >= lo) & (plant_row < lo + x.shape[0])
        if bool(m.any()):
            x[plant_row[m] - lo] = Q[plant_q[m]] * 3.0               # a row no random row can beat
        if lo == 0:
            keep["first"] = x[:4096].clone()
        ix.add(x)
    assert ix.ntotal == n
    # (1) the headline search: fp32 scan, 6980 queries, top-100
    ix.set_param("f32_scan", 1)
    D, I = ix.search_device(Q, 100)
    ix.check()
    assert torch.equal(I[plant_q, 0], plant_row)
    assert bool((D[:, :-1] >= D[:, 1:]).all()) and int(I.min()) >= 0 and int(I.max()) < n
    srt = torch.sort(I, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                    # no id twice in a list
    want = (3.0 * (Q[plant_q].double() ** 2).sum(1))
    assert bool(((D[plant_q, 0].double() - want).abs() <= 1e-4 * want).all())
    # (2) bf16 scan + exact fp32 rescoring: the same bits
    ix.set_param("f32_scan", 0)
    D2, I2 = ix.search_device(Q, 100)
    assert ix.last_bf16_stats()["candidates"] > 0
    assert torch.equal(I, I2) and torch.equal(D.view(torch.int32), D2.view(torch.int32))
    # (2b) a diagnostic build's other large-batch bf16 scan kernels at full size (ping-pong, direct corpus operand)
    for pp in ((1, 3) if diag_build() else ()):
        ix.set_param("bf16_pp", pp)
        D2b, I2b = ix.search_device(Q, 100)
        assert ix.last_bf16_stats()["candidates"] > 0
        assert torch.equal(I, I2b) and torch.equal(D.view(torch.int32), D2b.view(torch.int32)), pp
    ix.set_param("bf16_pp", 0)
    # (2c) configs[2]'s own depth, top-1000: the fp32 scan and the bf16 scan (hi + lo query parts) agree bit for
    #      bit, and top-100 is its prefix
    ix.set_param("f32_scan", 1)
    Dk, Ik = ix.search_device(Q, 1000)
    ix.check()
    assert bool((Dk[:, :-1] >= Dk[:, 1:]).all()) and int(Ik.min()) >= 0 and int(Ik.max()) < n
    assert torch.equal(Ik[:, :100], I) and torch.equal(Dk[:, :100].contiguous().view(torch.int32), D.view(torch.int32))
    ix.set_param("f32_scan", 0)
    Dk2, Ik2 = ix.search_device(Q, 1000)
    # (the planted rows have 3x the norm of the others: the index keeps them out of the bf16 scan and of the margin
    #  rule's norm bounds and scores them on the side — without that every query would go on to the fp32 scan here)
    st = ix.last_bf16_stats()
    assert st["candidates"] == 2000 and st["outlier_rows"] == plant_q.numel() and st["fallback_queries"] <= nq // 100
    assert torch.equal(Ik, Ik2) and torch.equal(Dk.view(torch.int32), Dk2.view(torch.int32))
    del Dk, Ik, Dk2, Ik2
    # (3) small batches on the fp32 rows (scan_f32_small_kernel) and on the bf16 copy: the same bits again,
    #     and top-10 is the prefix of top-100
    for bnq in (1, 16, 64, 150):
        for f32 in (1, 0):
            ix.set_param("f32_scan", f32)
            Ds, Is = ix.search_device(Q[:bnq].contiguous(), 100)
            assert torch.equal(Is, I[:bnq]) and torch.equal(Ds.view(torch.int32), D[:bnq].view(torch.int32)), (bnq, f32)
            D10, I10 = ix.search_device(Q[:bnq].contiguous(), 10)
            assert torch.equal(I10, I[:bnq, :10]) and torch.equal(D10, D[:bnq, :10])
    # (4) nothing in the first 4096 rows beats a query's k-th score beyond tolerance, unless it is in the list
    full = Q[:64].double() @ keep["first"].double().T
    kth = D[:64, -1].double()
    for qi in range(64):
        above = torch.nonzero(full[qi] > kth[qi] * (1 + 1e-4)).flatten()
        assert bool(torch.isin(above, I[qi]).all())


def test_config4_shard_12p5m_bf16_small_batches(R):
    import torch
    dev = torch.device("cuda:0")
    n, d, block = 12_500_000, 768, 500_000
    ix = R.FlatIPIndex(d, storage="bf16")
    ix.reserve(n)
    probe_rows = torch.tensor([0, 1, 499_999, 500_000, 6_250_123, 12_499_999] + [i * 195_311 + 7 for i in range(58)],
                              device=dev)
    probes = torch.empty((probe_rows.numel(), d), device=dev, dtype=torch.float32)
    for lo, x in _blocks(torch, dev, n, d, block, 777, torch.bfloat16):
        m = (probe_rows >= lo) & (probe_rows < lo + x.shape[0])
        if bool(m.any()):
            probes[m] = x[probe_rows[m] - lo].float()
        ix.add(x)
    assert ix.ntotal == n
    want = (probes.double() ** 2).sum(1)
    ref = None
    for bnq in (64, 1, 16):
        q = probes[:bnq].contiguous()
        D, I = ix.search_device(q, 100)
        ix.check()
        st = ix.last_bf16_stats()
        assert st["candidates"] > 0 and st["fallback_queries"] == 0
        assert torch.equal(I[:, 0], probe_rows[:bnq])                    # each stored row finds itself first
        assert bool(((D[:, 0].double() - want[:bnq]).abs() <= 1e-5 * want[:bnq]).all())
        assert bool((D[:, :-1] >= D[:, 1:]).all()) and int(I.min()) >= 0 and int(I.max()) < n
        if ref is None:
            ref = (D, I)
            ix.set_param("no_small_batch", 1)                            # the 256-query tile: same bits
            Db, Ib = ix.search_device(q, 100)
            ix.set_param("no_small_batch", 0)
            assert torch.equal(I, Ib) and torch.equal(D.view(torch.int32), Db.view(torch.int32))
        else:
            assert torch.equal(I, ref[1][:bnq]) and torch.equal(D.view(torch.int32), ref[0][:bnq].view(torch.int32))
        D10, I10 = ix.search_device(q, 10)
        assert torch.equal(I10, I[:, :10]) and torch.equal(D10, D[:, :10])
