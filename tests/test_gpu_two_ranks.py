"""Two and three rank PROCESSES on the one GPU of this box, each with the HIP engine (libipx.so) on its own row shard:
the N-GPU path (mural_amd/sharded.py, __main__._main_sharded: per-rank id base, the packed exchange block, all-gather,
ipx_merge_topk_packed / ipx_merge_topk on every rank, rank 0 writing the ranking) with real engines in real rank
processes. The exchange blocks are staged through host memory and gathered by gloo there (HostStagedHipEngine): RCCL
needs one GPU per rank (tests/test_zy_multi_gpu.py), and gloo's own device-tensor path stalled for seconds to for ever
when two processes kept one GPU busy — a property of that combination, not of the engine, but no
basis for a test. Replaces the reference's per-shard runs + reducer (docs/retrieval.md:31-53,
faiss_retriever/reducer.py:10-20); results are compared bit for bit with the oracle's single-index answer and with the
single-process CLI's bytes."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, DIM, NQ = 9001, 96, 70
DEPTHS = (10, 100, 3000)          # 3000 > IPX_MAX_K: one all-gather per array and the deep merge


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data():
    rng = np.random.default_rng(2024)
    return rng.standard_normal((N, DIM), dtype=np.float32), rng.standard_normal((NQ, DIM), dtype=np.float32)


class HostStagedHipEngine:
    """The product's _HipEngine (mural_amd/sharded.py) on GPU 0, presenting its results and exchange blocks in host
    memory, so that ShardedFlatIP's collectives run on CPU tensors."""

    def __init__(self, d, storage=None):
        from mural_amd.sharded import _HipEngine
        self.e = _HipEngine(d, 0, storage)

    def set_id_base(self, base):
        self.e.set_id_base(base)

    def add(self, x):
        self.e.add(x)

    def out_device(self):
        import torch
        return torch.device("cpu")

    def packed_block_bytes(self, nq, k):
        return self.e.packed_block_bytes(nq, k)

    def search(self, q, k, out=None):
        import torch
        D, I = self.e.search(torch.as_tensor(q).cuda(), k)
        if out is None:
            return D.cpu(), I.cpu()
        out[0].copy_(D)
        out[1].copy_(I)
        return out

    def merge_packed(self, blocks, S, nq, k, k_out):
        oD, oI = self.e.merge_packed(blocks.cuda(), S, nq, k, k_out)
        return oD.cpu(), oI.cpu()

    def merge(self, D, I, k_out):
        oD, oI = self.e.merge(D.cuda(), I.cuda(), k_out)
        return oD.cpu(), oI.cpu()


def _group(rank, world, port):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch, dist


def _search_worker(rank, world, port, out_dir, storage):
    torch, dist = _group(rank, world, port)
    from mural_amd.sharded import ShardedFlatIP, shard_bounds
    X, Q = _data()
    lo, hi = shard_bounds(N, world, rank)
    index = ShardedFlatIP(DIM, row_lo=lo, engine=HostStagedHipEngine(DIM, storage))
    half = (hi - lo) // 2                       # two adds: host rows, then device rows
    index.add_local(X[lo:lo + half])
    index.add_local(torch.from_numpy(X[lo + half:hi]).cuda())
    Qh = torch.from_numpy(Q)
    for k in DEPTHS:
        D, I = index.search(Qh, k)
        np.save(os.path.join(out_dir, f"D_{rank}_{k}.npy"), D.cpu().numpy())
        np.save(os.path.join(out_dir, f"I_{rank}_{k}.npy"), I.cpu().numpy())
    D, I = index.search(Q[:1], 5)               # one query: the latency-shaped exchange
    np.save(os.path.join(out_dir, f"D_{rank}_one.npy"), D.cpu().numpy())
    np.save(os.path.join(out_dir, f"I_{rank}_one.npy"), I.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,storage", [(2, "f32"), (3, "f32"), (2, "bf16scan")])
def test_rank_processes_with_hip_engines_return_the_single_index_result(tmp_path, world, storage):
    import torch.multiprocessing as mp
    from mural_amd import _lib
    assert _lib.lib().ipx_device_count() >= 1, "no HIP device visible: the search path has no CPU fallback"
    mp.spawn(_search_worker, args=(world, _free_port(), str(tmp_path), storage), nprocs=world, join=True)
    X, Q = _data()
    S = O.scores_mfma_order(Q, X)
    for k in DEPTHS:
        rD, rI = O.topk_total_order(S, k)
        for rank in range(world):               # every rank holds the global answer
            D, I = np.load(tmp_path / f"D_{rank}_{k}.npy"), np.load(tmp_path / f"I_{rank}_{k}.npy")
            assert np.array_equal(I, rI), (k, rank, np.argwhere(I != rI)[:3].tolist())
            assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), (k, rank)
    rD, rI = O.topk_total_order(S[:1], 5)
    for rank in range(world):
        assert np.array_equal(np.load(tmp_path / f"I_{rank}_one.npy"), rI)
        assert np.array_equal(np.load(tmp_path / f"D_{rank}_one.npy").view(np.uint32), rD.view(np.uint32))


def _cli_worker(rank, world, port, argv):
    torch, dist = _group(rank, world, port)
    from mural_amd.faiss_retriever import __main__ as drv
    drv._main_sharded(drv._parse_args(argv), engine_factory=HostStagedHipEngine)    # the group exists: the CLI uses it
    dist.destroy_process_group()


def test_cli_rank_processes_write_the_single_process_bytes(tmp_path):
    import torch.multiprocessing as mp
    from mural_amd.faiss_retriever import __main__ as drv
    argv = ["--query_reps", os.path.join(GOLDEN_DIR, "E_query_emb.pkl"),
            "--passage_reps", os.path.join(GOLDEN_DIR, "E_corpus_emb*.pkl"),
            "--depth", "10", "--batch_size", "4", "--save_text", "--quiet", "--save_ranking_to"]
    drv.main(argv + [str(tmp_path / "plain.tsv")])
    mp.spawn(_cli_worker, args=(2, _free_port(), argv + [str(tmp_path / "sharded.tsv")]), nprocs=2, join=True)
    assert (tmp_path / "sharded.tsv").read_bytes() == (tmp_path / "plain.tsv").read_bytes()
