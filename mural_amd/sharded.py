"""Row-sharded multi-GPU search: one process per GPU, RCCL all-gather + GPU merge.

This is the on-node replacement for the reference's sharded workflow
(/root/reference/docs/retrieval.md:31-53: one `python -m tevatron.faiss_retriever` run per
corpus shard, intermediate pickles, then reducer.py's ResultHeap merge): every rank holds a
contiguous row block of the corpus in HBM, all ranks search the same queries, the per-rank
(score, global id) candidates are exchanged with one all-gather per array (scores, ids;
torch.distributed, backend "nccl" = RCCL over xGMI) and merged on the GPU by the same kernel that serves reducer.py
(ipx_merge_topk). The result is identical to a single-index search because the top-k of a
union is the top-k of the per-part top-k's, and ties break on the global id.

The exchange is a direct all-gather of nq*k*(4+8) bytes per rank (6980 x 100: 8.4 MB), two
collectives per search (each gathers straight into the [world, nq, k] layout the merge kernel
reads); there is no other data-path communication.
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

    def __init__(self, d, device):
        from .faiss_retriever.retriever import FlatIPIndex
        self.index = FlatIPIndex(d, device=device)
        self.device = device

    def set_id_base(self, base):
        self.index.set_id_base(base)

    def add(self, x):
        self.index.add(x)                # host (numpy) rows or a torch CUDA tensor

    def search(self, q, k):
        import torch
        if not torch.is_tensor(q):       # host queries: one copy up, results stay on the device for the exchange
            q = torch.from_numpy(q).to(torch.device("cuda", self.device))
        return self.index.search_device(q, k)

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

    def __init__(self, d, row_lo, group=None, device=None, engine=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if engine is None:
            import torch
            if device is None:
                device = torch.cuda.current_device()
            engine = _HipEngine(d, device)
        self.engine = engine
        self.d = int(d)
        self.engine.set_id_base(int(row_lo))

    def add_local(self, rows):
        self.engine.add(rows)

    def search(self, q, k):
        import torch
        D, I = self.engine.search(q, k)          # [nq, k] local top-k with global ids
        if self.world == 1:
            return D, I
        nq = D.shape[0]
        # rank-major concatenation along dim 0 == [world, nq, k], the layout ipx_merge_topk takes
        gD = torch.empty((self.world * nq, k), dtype=D.dtype, device=D.device)
        gI = torch.empty((self.world * nq, k), dtype=I.dtype, device=I.device)
        self.dist.all_gather_into_tensor(gD, D.contiguous(), group=self.group)
        self.dist.all_gather_into_tensor(gI, I.contiguous(), group=self.group)
        return self.engine.merge(gD.view(self.world, nq, k), gI.view(self.world, nq, k), k)
