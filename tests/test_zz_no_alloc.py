"""ipx_search_noalloc (include/ipx.h): the entry point that allocates nothing, called through the C ABI and compared
bit for bit with ipx_search on the same index — two calls in a row on one workspace, the second with a smaller
input (stale contents of the first must not leak into it), then shapes at and beyond the workspace's limits."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _search(L, h, fn, q, k, stream):
    import torch
    D = torch.empty((q.shape[0], k), dtype=torch.float32, device=q.device)
    I = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
    if fn == "noalloc":
        rc = L.ipx_search_noalloc(h, C.c_void_p(q.data_ptr()), q.shape[0], k, C.c_void_p(D.data_ptr()),
                                  C.c_void_p(I.data_ptr()), stream)
    else:
        rc = L.ipx_search(h, C.c_void_p(q.data_ptr()), q.shape[0], k, C.c_void_p(D.data_ptr()),
                          C.c_void_p(I.data_ptr()), 1, stream)
    return rc, D, I


def test_noalloc_search_equals_search_on_two_calls_in_a_row():
    import torch
    from mural_amd import _lib
    from oracle import oracle as O
    L = _lib.lib()
    assert L.ipx_device_count() >= 1, "no HIP device visible: the search path has no CPU fallback"
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    n, d = 300_000, 768
    X = torch.randn((n, d), generator=g, device=dev)
    h = C.c_void_p()
    _lib.check(L.ipx_create(d, 0, C.byref(h)))
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    try:
        _lib.check(L.ipx_add(h, C.c_void_p(X.data_ptr()), n, _lib.IPX_F32, 1, st))
        Q1 = torch.randn((150, d), generator=g, device=dev)      # three query blocks of 64
        Q2 = torch.randn((7, d), generator=g, device=dev)        # smaller: one block of 32, more splits, other strides
        for Q, k in ((Q1, 100), (Q2, 10)):
            free0 = torch.cuda.mem_get_info(dev)[0]
            rc, D, I = _search(L, h, "noalloc", Q, k, st)
            _lib.check(rc)
            _lib.check(L.ipx_check(h, st))
            assert torch.cuda.mem_get_info(dev)[0] == free0, "ipx_search_noalloc changed the device's free memory"
            rc, D0, I0 = _search(L, h, "search", Q, k, st)
            _lib.check(rc)
            _lib.check(L.ipx_check(h, st))
            assert torch.equal(I, I0), "ids differ from ipx_search"
            assert torch.equal(D.view(torch.int32), D0.view(torch.int32)), "score bits differ from ipx_search"
        # and against the oracle (kernel-order scores, total order) on the smaller call
        rD, rI = O.topk_total_order(O.scores_mfma_order(Q2.cpu().numpy(), X[:20000].cpu().numpy()), 10)
        h2 = C.c_void_p()
        _lib.check(L.ipx_create(d, 0, C.byref(h2)))
        try:
            _lib.check(L.ipx_add(h2, C.c_void_p(X.data_ptr()), 20000, _lib.IPX_F32, 1, st))
            rc, D, I = _search(L, h2, "noalloc", Q2, 10, st)
            _lib.check(rc)
            _lib.check(L.ipx_check(h2, st))
            assert np.array_equal(I.cpu().numpy(), rI) and np.array_equal(D.cpu().numpy().view(np.uint32), rD.view(np.uint32))
        finally:
            L.ipx_destroy(h2)
        # an index that also keeps a bf16 copy (storage 1): the entry point scans its fp32 rows, same bits
        h4 = C.c_void_p()
        _lib.check(L.ipx_create(d, 0, C.byref(h4)))
        try:
            _lib.check(L.ipx_set_param(h4, b"storage", 1))
            _lib.check(L.ipx_add(h4, C.c_void_p(X.data_ptr()), 20000, _lib.IPX_F32, 1, st))
            free0 = torch.cuda.mem_get_info(dev)[0]
            rc, D, I = _search(L, h4, "noalloc", Q2, 10, st)
            _lib.check(rc)
            _lib.check(L.ipx_check(h4, st))
            assert torch.cuda.mem_get_info(dev)[0] == free0
            assert np.array_equal(I.cpu().numpy(), rI) and np.array_equal(D.cpu().numpy().view(np.uint32), rD.view(np.uint32))
            rc, D0, I0 = _search(L, h4, "search", Q2, 10, st)          # the bf16 scan + rescoring of the same index
            _lib.check(rc)
            assert torch.equal(I, I0) and torch.equal(D.view(torch.int32), D0.view(torch.int32))
        finally:
            L.ipx_destroy(h4)
        # limits: a clear error, nothing launched
        big = torch.randn((193, d), generator=g, device=dev)
        rc, _, _ = _search(L, h, "noalloc", big, 10, st)
        assert rc == 1 and b"outside [0, 192]" in L.ipx_last_error()
        rc, _, _ = _search(L, h, "noalloc", Q1[:1], 2049, st)
        assert rc == 1
        # the deepest list with the largest batch still fits (two corpus splits), and matches
        rc, D, I = _search(L, h, "noalloc", big[:192], 2048, st)
        _lib.check(rc)
        rc, D0, I0 = _search(L, h, "search", big[:192], 2048, st)
        _lib.check(rc)
        _lib.check(L.ipx_check(h, st))
        assert torch.equal(I, I0) and torch.equal(D.view(torch.int32), D0.view(torch.int32))
        # very long rows: the packed queries of 192 of them alone exceed the workspace -> reported with the sizes;
        # a few queries of the same index fit and match
        dl = 49152
        h3 = C.c_void_p()
        _lib.check(L.ipx_create(dl, 0, C.byref(h3)))
        try:
            Xl = torch.randn((40, dl), generator=g, device=dev)
            Ql = torch.randn((192, dl), generator=g, device=dev)
            _lib.check(L.ipx_add(h3, C.c_void_p(Xl.data_ptr()), 40, _lib.IPX_F32, 1, st))
            rc, _, _ = _search(L, h3, "noalloc", Ql, 5, st)
            assert rc == 1 and b"bytes of workspace" in L.ipx_last_error()
            rc, D, I = _search(L, h3, "noalloc", Ql[:7], 5, st)
            _lib.check(rc)
            rc, D0, I0 = _search(L, h3, "search", Ql[:7], 5, st)
            _lib.check(rc)
            _lib.check(L.ipx_check(h3, st))
            assert torch.equal(I, I0) and torch.equal(D.view(torch.int32), D0.view(torch.int32))
        finally:
            L.ipx_destroy(h3)
    finally:
        L.ipx_destroy(h)
