"""Host-side mirror of tevatron.faiss_retriever.retriever on the MI355X engine.

Same names, argument meaning and error behaviour as the reference
(/root/reference/src/tevatron/faiss_retriever/retriever.py:10-42); the arithmetic that the
reference delegates to faiss.IndexFlatIP runs in libipx.so (hand-written gfx950 kernels,
C ABI in include/ipx.h). There is no CPU path: without the library or a GPU, calls raise.
"""
import ctypes as C
import logging
import os

import numpy as np
from tqdm import tqdm

from .. import _lib

logger = logging.getLogger(__name__)

_NP_DTYPES = {np.dtype("float32"): _lib.IPX_F32, np.dtype("float16"): _lib.IPX_F16}


def _is_torch_tensor(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _torch_stream(x):
    import torch
    return C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)


class FlatIPIndex:
    """What `BaseFaissIPRetriever.index` is in the reference: a flat inner-product index with
    `.d`, `.ntotal`, `.add(x)` and `.search(x, k)` (faiss.IndexFlatIP through its numpy
    wrapper, retriever.py:12,16,19). Rows live on one GPU in the engine's packed layout."""

    #: batch_search sends consecutive chunks down in one call of at most this many queries
    coalesce_queries = 8192

    #: IPX_STORAGE / storage= values (include/ipx.h, ipx_set_param "storage")
    STORAGE = {"f32": 0, "bf16scan": 1, "bf16": 2}

    def __init__(self, d, device=None, storage=None):
        if device is None:
            device = int(os.environ.get("IPX_DEVICE", "0"))
        if storage is None:
            storage = os.environ.get("IPX_STORAGE", "f32")
        self.d = int(d)
        self.device = int(device)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().ipx_create(self.d, self.device, C.byref(self._h)))
        if self.STORAGE[storage]:
            self.set_param("storage", self.STORAGE[storage])
        self.storage = storage

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().ipx_destroy(h)
            except Exception:   # interpreter shutdown
                pass

    @property
    def ntotal(self):
        return int(_lib.lib().ipx_ntotal(self._h))

    @property
    def is_trained(self):
        return True

    def reserve(self, n):
        _lib.check(_lib.lib().ipx_reserve(self._h, int(n)))

    def set_id_base(self, id_base):
        _lib.check(_lib.lib().ipx_set_id_base(self._h, int(id_base)))

    def add(self, x):
        """faiss wrapper semantics: x is [n, d]; a dimension mismatch is an AssertionError.
        float16 rows are widened on the device (the reference widens them on the host,
        __main__.py:79); torch CUDA tensors (f32 / f16 / bf16) are read in place."""
        L = _lib.lib()
        if _is_torch_tensor(x):
            import torch
            n, d = x.shape
            assert d == self.d
            dt = {torch.float32: _lib.IPX_F32, torch.float16: _lib.IPX_F16, torch.bfloat16: _lib.IPX_BF16}[x.dtype]
            assert x.is_cuda and x.device.index == self.device
            x = x.contiguous()
            _lib.check(L.ipx_add(self._h, C.c_void_p(x.data_ptr()), n, dt, 1, _torch_stream(x)))
            return
        n, d = x.shape
        assert d == self.d
        if x.dtype not in _NP_DTYPES:
            x = np.ascontiguousarray(x, dtype="float32")
        x = np.ascontiguousarray(x)
        _lib.check(L.ipx_add(self._h, C.c_void_p(x.ctypes.data), n, _NP_DTYPES[x.dtype], 0, None))

    def search(self, x, k):
        """Returns (D float32 [n, k], I int64 [n, k]) as numpy arrays; rows sorted by score
        descending, (-FLT_MAX, -1) in slots beyond ntotal — faiss.IndexFlatIP.search."""
        L = _lib.lib()
        if _is_torch_tensor(x):
            return self.search_device(x, k)
        n, d = x.shape
        assert d == self.d
        assert k > 0
        if k > _lib.IPX_MAX_DEPTH:
            raise ValueError(f"k={k} exceeds the engine limit IPX_MAX_DEPTH={_lib.IPX_MAX_DEPTH}")
        x = np.ascontiguousarray(x, dtype="float32")
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        _lib.check(L.ipx_search(self._h, C.c_void_p(x.ctypes.data), n, k, C.c_void_p(D.ctypes.data),
                                C.c_void_p(I.ctypes.data), 0, None))
        return D, I

    def search_device(self, x, k, out=None):
        """Stream-ordered search on torch CUDA tensors; returns CUDA tensors (D, I)."""
        import torch
        L = _lib.lib()
        n, d = x.shape
        assert d == self.d
        assert k > 0
        x = x.contiguous().float()
        if out is None:
            D = torch.empty((n, k), dtype=torch.float32, device=x.device)
            I = torch.empty((n, k), dtype=torch.int64, device=x.device)
        else:
            D, I = out
        _lib.check(L.ipx_search(self._h, C.c_void_p(x.data_ptr()), n, k, C.c_void_p(D.data_ptr()),
                                C.c_void_p(I.data_ptr()), 1, _torch_stream(x)))
        return D, I

    def search_aspect(self, aspect_emb, weights, k, logits=False):
        """MURAL multi-aspect query (modeling/mul_grain.py:97-110): aspect_emb [n, A, d],
        weights [n, A] (post-softmax, or logits with logits=True); fused on the device."""
        L = _lib.lib()
        aspect_emb = np.ascontiguousarray(aspect_emb, dtype="float32")
        weights = np.ascontiguousarray(weights, dtype="float32")
        n, A, d = aspect_emb.shape
        assert d == self.d
        assert weights.shape == (n, A)
        assert k > 0
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        _lib.check(L.ipx_search_aspect(self._h, C.c_void_p(aspect_emb.ctypes.data), C.c_void_p(weights.ctypes.data),
                                       A, int(bool(logits)), n, k, C.c_void_p(D.ctypes.data),
                                       C.c_void_p(I.ctypes.data), 0, None))
        return D, I

    def search_aspect_device(self, aspect_emb, weights, k, logits=False, out=None):
        """search_aspect on torch CUDA tensors (aspect_emb [n, A, d], weights [n, A], fp32), stream-ordered;
        returns CUDA tensors (D, I)."""
        import torch
        n, A, d = aspect_emb.shape
        assert d == self.d and tuple(weights.shape) == (n, A) and k > 0
        aspect_emb, weights = aspect_emb.contiguous().float(), weights.contiguous().float()
        if out is None:
            D = torch.empty((n, k), dtype=torch.float32, device=aspect_emb.device)
            I = torch.empty((n, k), dtype=torch.int64, device=aspect_emb.device)
        else:
            D, I = out
        _lib.check(_lib.lib().ipx_search_aspect(self._h, C.c_void_p(aspect_emb.data_ptr()), C.c_void_p(weights.data_ptr()),
                                                A, int(bool(logits)), n, k, C.c_void_p(D.data_ptr()),
                                                C.c_void_p(I.data_ptr()), 1, _torch_stream(aspect_emb)))
        return D, I

    # ---- MURAL pooler on the device (MULGRAIN_Pooler.forward 'att', modeling/mul_grain.py:97-132) ----
    def _pool_args(self, hidden, weight, bias, begin_id):
        hidden = np.ascontiguousarray(hidden, dtype="float32")
        weight = np.ascontiguousarray(weight, dtype="float32")
        bias = np.ascontiguousarray(bias, dtype="float32")
        assert hidden.ndim == 3 and weight.ndim == 2
        n, seq_len, d = hidden.shape
        A = weight.shape[0]
        assert d == self.d and weight.shape == (A, d) and bias.shape == (A,)
        keep = (hidden, weight, bias)
        return keep, (C.c_void_p(hidden.ctypes.data), seq_len, C.c_void_p(weight.ctypes.data),
                      C.c_void_p(bias.ctypes.data), A, int(begin_id), n), n, A

    def pool(self, hidden, weight, bias, begin_id=1):
        """hidden [n, seq_len, d] last hidden states, weight [A, d] / bias [A] = cls_linear_q (= cls_linear_d
        when tied) of pooler.pt; aspect tokens at begin_id .. begin_id+A-1 (src/pretrain_finetune1.sh:22-23:
        A=3, begin_id=1). Returns (rep [n, d], att [n, A]) like the reference's (res, _, q_att)."""
        keep, args, n, A = self._pool_args(hidden, weight, bias, begin_id)
        rep = np.empty((n, self.d), dtype=np.float32)
        att = np.empty((n, A), dtype=np.float32)
        _lib.check(_lib.lib().ipx_pool(self._h, *args, C.c_void_p(rep.ctypes.data), C.c_void_p(att.ctypes.data), 0, None))
        return rep, att

    def add_pooled(self, hidden, weight, bias, begin_id=1):
        """Pool passages' hidden states and append the representations (encode + index without the pickles)."""
        keep, args, n, A = self._pool_args(hidden, weight, bias, begin_id)
        _lib.check(_lib.lib().ipx_add_pooled(self._h, *args, 0, None))

    def search_pooled(self, hidden, weight, bias, k, begin_id=1, return_att=False):
        """Pool queries' hidden states and search, one call."""
        keep, args, n, A = self._pool_args(hidden, weight, bias, begin_id)
        assert k > 0
        D = np.empty((n, k), dtype=np.float32)
        I = np.empty((n, k), dtype=np.int64)
        att = np.empty((n, A), dtype=np.float32)
        _lib.check(_lib.lib().ipx_search_pooled(self._h, *args, k, C.c_void_p(D.ctypes.data), C.c_void_p(I.ctypes.data),
                                                C.c_void_p(att.ctypes.data), 0, None))
        return (D, I, att) if return_att else (D, I)

    # bench / profiling helpers
    def set_param(self, name, value):
        _lib.check(_lib.lib().ipx_set_param(self._h, name.encode(), int(value)))

    def set_timing(self, on=True):
        _lib.check(_lib.lib().ipx_set_timing(self._h, int(on)))

    def last_timing(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _lib.check(_lib.lib().ipx_last_timing(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"pack_ms": a.value, "scan_ms": b.value, "merge_ms": c.value}

    def last_bf16_stats(self):
        a, b, c = C.c_float(), C.c_int(), C.c_int64()
        _lib.check(_lib.lib().ipx_last_bf16_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"rescore_ms": a.value, "candidates": b.value, "fallback_queries": c.value,
                "outlier_rows": int(_lib.lib().ipx_outlier_rows(self._h)),
                "exact_queries": int(_lib.lib().ipx_last_exact_queries(self._h))}

    def last_geometry(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _lib.check(_lib.lib().ipx_last_geometry(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"workgroups": a.value, "splits": b.value, "queue_cap": c.value}

    def read_trace(self):
        n = self.last_geometry()["workgroups"]
        out = np.zeros((n, 4), dtype=np.uint64)
        _lib.check(_lib.lib().ipx_read_trace(self._h, C.c_void_p(out.ctypes.data), n))
        return out

    def check(self, stream=None):
        _lib.check(_lib.lib().ipx_check(self._h, stream))


class BaseFaissIPRetriever:
    """retriever.py:10-32 — the constructor only takes the dimension from `init_reps`
    (it does not add the rows); add / search / batch_search forward to the index."""

    def __init__(self, init_reps: np.ndarray):
        index = FlatIPIndex(init_reps.shape[1])
        self.index = index

    def add(self, p_reps: np.ndarray):
        self.index.add(p_reps)

    def search(self, q_reps: np.ndarray, k: int):
        return self.index.search(q_reps, k)

    def batch_search(self, q_reps: np.ndarray, k: int, batch_size: int, quiet: bool = False):
        """retriever.py:21-32: search `batch_size` queries at a time (tqdm bar unless quiet)
        and concatenate. A query's result does not depend on its batch (scores are bit-identical
        whatever the chunking), so on the GPU index consecutive chunks are sent down in one call
        of up to `coalesce_queries` queries — the corpus is streamed once per call, and the
        reference's default of 128 queries per chunk would stream it 55 times for 6980 queries.
        The bar still counts chunks of `batch_size`."""
        starts = range(0, q_reps.shape[0], batch_size)
        group = max(1, getattr(self.index, "coalesce_queries", 0) // max(batch_size, 1))
        parts = []
        with tqdm(total=len(starts), disable=quiet) as bar:
            for g0 in range(0, len(starts), group):
                chunk_starts = starts[g0: g0 + group]
                lo, hi = chunk_starts[0], chunk_starts[-1] + batch_size
                parts.append(self.search(q_reps[lo: hi], k))
                bar.update(len(chunk_starts))
        all_scores = np.concatenate([p[0] for p in parts], axis=0)
        all_indices = np.concatenate([p[1] for p in parts], axis=0)
        return all_scores, all_indices


class FaissRetriever(BaseFaissIPRetriever):
    """retriever.py:35-42 — index_factory retriever; never referenced by the reference's own
    code. Only the exact "Flat" factory is served here (anything else is approximate search,
    out of scope for an exact engine)."""

    def __init__(self, init_reps: np.ndarray, factory_str: str):
        if factory_str.strip() != "Flat":
            raise NotImplementedError(
                f"factory string {factory_str!r}: only the exact 'Flat' inner-product index is implemented")
        super().__init__(init_reps)
