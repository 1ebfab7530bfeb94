"""Row-sharded multi-GPU search: one process per GPU, RCCL all-gather + GPU merge.

This is the on-node replacement for the reference's sharded workflow
(/root/reference/docs/retrieval.md:31-53: one `python -m tevatron.faiss_retriever` run per
corpus shard, intermediate pickles, then reducer.py's ResultHeap merge): every rank holds a
contiguous row block of the corpus in HBM, all ranks search the same queries, the per-rank
(score, global id) candidates are exchanged with ONE all-gather (torch.distributed, backend
"nccl" = RCCL over xGMI) and merged on the GPU by the same kernel that serves reducer.py
(ipx_merge_topk_packed). The result is identical to a single-index search because the top-k of a
union is the top-k of the per-part top-k's, and ties break on the global id.

The exchange is a direct all-gather of nq*k*(4+8) bytes per rank (6980 x 100: 8.4 MB; one query:
1.2 KB, where the collective's latency is what counts): the local search writes its ids and
scores straight into one block (ids first, then scores: packed_block_bytes), the blocks are
gathered into the [world][block] layout the merge kernel reads; there is no other data-path
communication. Lists deeper than one pass (k > IPX_MAX_K) take one all-gather per array and
ipx_merge_topk.
"""
import ctypes as C


from . import _lib


def shard_bounds(n_rows, world_size, rank):
    """Contiguous, near-equal row blocks: rank r owns [lo, hi)."""
    base, rem = divmod(int(n_rows), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _HipEngine:
    """Local search + merge on this rank's GPU through the C ABI (torch CUDA tensors in/out)."""

    def __init__(self, d, device, storage=None):
        from .faiss_retriever.retriever import FlatIPIndex
        self.index = FlatIPIndex(d, device=device, storage=storage)
        self.device = device

    def set_id_base(self, base):
        self.index.set_id_base(base)

    def add(self, x):
        self.index.add(x)                # host (numpy) rows or a torch CUDA tensor

    def out_device(self):
        import torch
        return torch.device("cuda", self.device)

    def search(self, q, k, out=None):
        import torch
        if not torch.is_tensor(q):       # host queries: one copy up, results stay on the device for the exchange
            q = torch.from_numpy(q).to(self.out_device())
        return self.index.search_device(q, k, out=out)

    def packed_block_bytes(self, nq, k):
        return int(_lib.lib().ipx_packed_block_bytes(nq, k))

    def merge_packed(self, blocks, S, nq, k, k_out):
        """blocks: uint8 [S * packed_block_bytes(nq, k)] on the device, one block per shard."""
        import torch
        oD = torch.empty((nq, k_out), dtype=torch.float32, device=blocks.device)
        oI = torch.empty((nq, k_out), dtype=torch.int64, device=blocks.device)
        st = C.c_void_p(torch.cuda.current_stream(blocks.device).cuda_stream)
        _lib.check(_lib.lib().ipx_merge_topk_packed(C.c_void_p(blocks.data_ptr()), S, nq, k, k_out, C.c_void_p(oD.data_ptr()),
                                                    C.c_void_p(oI.data_ptr()), self.device, st))
        return oD, oI

    def merge(self, D, I, k_out):
        import torch
        S, nq, k_in = D.shape
        oD = torch.empty((nq, k_out), dtype=torch.float32, device=D.device)
        oI = torch.empty((nq, k_out), dtype=torch.int64, device=D.device)
        st = C.c_void_p(torch.cuda.current_stream(D.device).cuda_stream)
        _lib.check(_lib.lib().ipx_merge_topk(C.c_void_p(D.data_ptr()), C.c_void_p(I.data_ptr()), S, nq, k_in, k_out,
                                             C.c_void_p(oD.data_ptr()), C.c_void_p(oI.data_ptr()), 1, self.device, st))
        return oD, oI


class ShardedFlatIP:
    """Flat inner-product index row-sharded over the ranks of a torch.distributed group.

    add_local(rows) appends this rank's rows (its global ids start at `row_lo`);
    search(q, k) returns the global top-k on every rank.
    `engine` is injectable so the collective wiring can be exercised on CPU (gloo) in tests.
    """

    def __init__(self, d, row_lo, group=None, device=None, engine=None, storage=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if engine is None:
            import torch
            if device is None:
                device = torch.cuda.current_device()
            engine = _HipEngine(d, device, storage)
        self.engine = engine
        self.d = int(d)
        self.engine.set_id_base(int(row_lo))

    def add_local(self, rows):
        self.engine.add(rows)

    def search(self, q, k, exchange=None):
        """Global top-k on every rank. `exchange`: True forces the all-gather + merge even with one rank (what the
        8-rank job does per search, measurable on one GPU), None = only when there is more than one rank."""
        import torch
        if exchange is None:
            exchange = self.world > 1
        nq = int(q.shape[0])
        packed = exchange and k <= _lib.IPX_MAX_K and nq > 0 and hasattr(self.engine, "merge_packed")
        if packed:
            # ids [nq, k] int64, then scores [nq, k] fp32, in one block: the search writes into it, one collective
            # moves it, the merge reads the gathered blocks as they lie
            dev = q.device if torch.is_tensor(q) and q.is_cuda else self.engine.out_device()
            block = self.engine.packed_block_bytes(nq, k)
            mine = torch.empty(block, dtype=torch.uint8, device=dev)
            I = mine[: nq * k * 8].view(torch.int64).view(nq, k)
            D = mine[nq * k * 8: nq * k * 12].view(torch.float32).view(nq, k)
            self.engine.search(q, k, out=(D, I))
            if self.world == 1 and not self.dist.is_initialized():
                return self.engine.merge_packed(mine, 1, nq, k, k)
            allb = torch.empty(self.world * block, dtype=torch.uint8, device=dev)
            self.dist.all_gather_into_tensor(allb, mine, group=self.group)
            return self.engine.merge_packed(allb, self.world, nq, k, k)
        D, I = self.engine.search(q, k)          # [nq, k] local top-k with global ids
        if not exchange:
            return D, I
        if self.world == 1 and not self.dist.is_initialized():
            return self.engine.merge(D.view(1, nq, k), I.view(1, nq, k), k)
        # rank-major concatenation along dim 0 == [world, nq, k], the layout ipx_merge_topk takes
        gD = torch.empty((self.world * nq, k), dtype=D.dtype, device=D.device)
        gI = torch.empty((self.world * nq, k), dtype=I.dtype, device=I.device)
        self.dist.all_gather_into_tensor(gD, D.contiguous(), group=self.group)
        self.dist.all_gather_into_tensor(gI, I.contiguous(), group=self.group)
        return self.engine.merge(gD.view(self.world, nq, k), gI.view(self.world, nq, k), k)
