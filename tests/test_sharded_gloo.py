"""The N>1 path on CPU: two processes, gloo backend, the same ShardedFlatIP wiring
(partition -> local top-k with global ids -> all-gather -> merge) that runs over RCCL on the
GPUs. The per-rank engine is injected: here it is the CPU oracle (tests may use it as the
checker/stand-in), on the GPU it is the HIP engine (_HipEngine)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    def __init__(self, d):
        from oracle import oracle as O
        self.O, self.d, self.base, self.X = O, d, 0, None

    def set_id_base(self, base):
        self.base = base

    def add(self, x):
        x = x.numpy()
        self.X = x if self.X is None else np.concatenate([self.X, x])

    def search(self, q, k, out=None):
        D, I = self.O.topk_total_order(self.O.scores_mfma_order(q.numpy(), self.X), k, id_base=self.base)
        return self._deliver(torch.from_numpy(D), torch.from_numpy(I), out)

    @staticmethod
    def _deliver(D, I, out):
        if out is None:
            return D, I
        out[0].copy_(D)
        out[1].copy_(I)
        return out

    def merge(self, D, I, k_out):
        oD, oI = self.O.merge_total_order(D.numpy(), I.numpy(), k_out)
        return torch.from_numpy(oD), torch.from_numpy(oI)

    # the one-collective exchange format (include/ipx.h, ipx_merge_topk_packed), read back on the CPU
    def out_device(self):
        return torch.device("cpu")

    def packed_block_bytes(self, nq, k):
        return (nq * k * 12 + 15) // 16 * 16

    def merge_packed(self, blocks, S, nq, k, k_out):
        b = blocks.numpy().reshape(S, -1)
        assert b.shape[1] == self.packed_block_bytes(nq, k)
        I = np.stack([b[s, : nq * k * 8].copy().view(np.int64).reshape(nq, k) for s in range(S)])
        D = np.stack([b[s, nq * k * 8: nq * k * 12].copy().view(np.float32).reshape(nq, k) for s in range(S)])
        return self.merge(torch.from_numpy(D), torch.from_numpy(I), k_out)


def _worker(rank, world, port, n, d, nq, k, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mural_amd.sharded import ShardedFlatIP, shard_bounds
        rng = np.random.default_rng(1234)
        X = rng.standard_normal((n, d), dtype=np.float32)
        Q = np.random.default_rng(4321).standard_normal((nq, d), dtype=np.float32)
        lo, hi = shard_bounds(n, world, rank)
        sh = ShardedFlatIP(d, row_lo=lo, engine=OracleEngine(d))
        sh.add_local(torch.from_numpy(X[lo:hi]))
        D, I = sh.search(torch.from_numpy(Q), k)
        ret[rank] = (D.numpy(), I.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_packed_block_size_is_the_librarys():
    from mural_amd import _lib
    e = OracleEngine(8)
    for nq, k in ((1, 1), (1, 100), (3, 7), (64, 100), (6980, 1000)):
        assert _lib.lib().ipx_packed_block_bytes(nq, k) == e.packed_block_bytes(nq, k)


# k <= 2048: one all-gather of packed (ids, scores) blocks; deeper: one all-gather per array
@pytest.mark.parametrize("n,k", [(1001, 10), (333, 200), (2500, 2100)])
def test_two_rank_gloo_equals_single_index(n, k):
    from oracle import oracle as O
    d, nq, world = 48, 17, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, d, nq, k, ret), nprocs=world, join=True)
    X = np.random.default_rng(1234).standard_normal((n, d), dtype=np.float32)
    Q = np.random.default_rng(4321).standard_normal((nq, d), dtype=np.float32)
    rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k)
    for rank in range(world):
        D, I = ret[rank]
        assert np.array_equal(D, rD) and np.array_equal(I, rI)


def test_shard_bounds_cover_rows_exactly():
    from mural_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 8_800_000, 100_000_001):
        for w in (1, 2, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


# ---- the CLI under torch.distributed.run: shard files dealt to the ranks, rank 0 writes ----------
class NumpyOracleEngine(OracleEngine):
    def add(self, x):
        x = np.asarray(x, dtype=np.float32)
        self.X = x if self.X is None else np.concatenate([self.X, x])

    def search(self, q, k, out=None):
        q = q.numpy() if torch.is_tensor(q) else np.asarray(q, dtype=np.float32)
        if self.X is None:                                   # a rank that owns no shard file
            return self._deliver(torch.full((q.shape[0], k), -np.finfo(np.float32).max),
                                 torch.full((q.shape[0], k), -1, dtype=torch.int64), out)
        D, I = self.O.topk_total_order(self.O.scores_mfma_order(q, self.X), k, id_base=self.base)
        return self._deliver(torch.from_numpy(D), torch.from_numpy(I), out)


def _cli_worker(rank, world, port, tmp, depth):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    from mural_amd.faiss_retriever import __main__ as drv
    args = drv._parse_args(["--query_reps", os.path.join(tmp, "q.pkl"), "--passage_reps", os.path.join(tmp, "corpus_emb*.pkl"),
                            "--depth", str(depth), "--batch_size", "4", "--save_ranking_to", os.path.join(tmp, "rank.tsv"),
                            "--save_text", "--quiet"])
    drv._main_sharded(args, engine_factory=NumpyOracleEngine)


@pytest.mark.parametrize("world,nfiles", [(2, 3), (2, 1), (3, 2)])
def test_cli_under_torchrun_equals_single_process_output(tmp_path, world, nfiles):
    import glob
    import pickle
    from oracle import oracle as O
    from mural_amd.faiss_retriever import __main__ as drv
    d, nq, depth = 24, 9, 7
    rng = np.random.default_rng(99)
    sizes = [57, 131, 20][:nfiles]
    for i, n in enumerate(sizes):
        reps = rng.standard_normal((n, d), dtype=np.float32)
        if i == 1:
            reps = reps.astype(np.float16)
        with open(tmp_path / f"corpus_emb{i}.pkl", "wb") as f:
            pickle.dump((reps, [f"D{i}_{j}" for j in range(n)]), f)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    with open(tmp_path / "q.pkl", "wb") as f:
        pickle.dump((Q, [str(100 + j) for j in range(nq)]), f)
    mp.spawn(_cli_worker, args=(world, _free_port(), str(tmp_path), depth), nprocs=world, join=True)
    # expected: the shards concatenated in glob order (what the single-process add loop builds)
    X, look = [], []
    for f in glob.glob(str(tmp_path / "corpus_emb*.pkl")):
        reps, lk = drv.pickle_load(f)
        X.append(np.asarray(reps, dtype=np.float32))
        look += lk
    X = np.concatenate(X)
    rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), depth)
    drv.write_ranking(drv._lookup_strings(look, rI), rD, [str(100 + j) for j in range(nq)], str(tmp_path / "want.tsv"))
    assert (tmp_path / "rank.tsv").read_bytes() == (tmp_path / "want.tsv").read_bytes()
    assert len((tmp_path / "rank.tsv").read_bytes().splitlines()) == nq * depth


def _failing_cli_worker(rank, world, port, tmp, out):
    """Runs the sharded CLI on broken input and records how this rank's call ended."""
    try:
        _cli_worker(rank, world, port, tmp, 5)
        said = "finished"
    except BaseException as e:                               # noqa: BLE001
        said = f"{type(e).__name__}: {e}"
    with open(os.path.join(out, f"rank{rank}.txt"), "w") as f:
        f.write(said)


@pytest.mark.parametrize("broken", ["shard", "queries"])
def test_cli_ranks_raise_together_when_a_load_fails(tmp_path, broken):
    """A rank whose load step fails must not leave the others in a collective: with a shard
    file rank 0 cannot unpickle, or a query file no rank can, every rank's call ends with an error promptly."""
    import pickle
    import time
    d = 16
    rng = np.random.default_rng(3)
    for i, n in enumerate([30, 41]):
        with open(tmp_path / f"corpus_emb{i}.pkl", "wb") as f:
            pickle.dump((rng.standard_normal((n, d), dtype=np.float32), [f"D{i}_{j}" for j in range(n)]), f)
    with open(tmp_path / "q.pkl", "wb") as f:
        pickle.dump((rng.standard_normal((4, d), dtype=np.float32), ["1", "2", "3", "4"]), f)
    (tmp_path / ("corpus_emb1.pkl" if broken == "shard" else "q.pkl")).write_bytes(b"not a pickle")
    out = tmp_path / "said"
    out.mkdir()
    t0 = time.monotonic()
    mp.spawn(_failing_cli_worker, args=(2, _free_port(), str(tmp_path), str(out)), nprocs=2, join=True)
    assert time.monotonic() - t0 < 60
    said = [(out / f"rank{r}.txt").read_text() for r in range(2)]
    assert all(s != "finished" for s in said), said
    if broken == "shard":                                     # rank 0 read the ids: its own error; rank 1 is told
        assert "rank 0 failed while loading" in said[1], said
    assert not (tmp_path / "rank.tsv").exists()


def test_one_command_spawns_the_ranks(tmp_path, monkeypatch):
    """IPX_GPUS=N: the plain CLI command starts N rank processes itself (mural_amd/faiss_retriever/__main__.py::
    _spawn_ranks) — here 2 ranks over gloo with the oracle engine injected through the rank command — and the
    ranking file has the bytes of the single-process output. A stray WORLD_SIZE without RANK / MASTER_PORT does
    not switch the CLI to the distributed path."""
    import glob
    import pickle
    from oracle import oracle as O
    from mural_amd.faiss_retriever import __main__ as drv
    d, nq, depth = 24, 9, 7
    rng = np.random.default_rng(5)
    for i, n in enumerate([40, 77, 15]):
        with open(tmp_path / f"corpus_emb{i}.pkl", "wb") as f:
            pickle.dump((rng.standard_normal((n, d), dtype=np.float32), [f"D{i}_{j}" for j in range(n)]), f)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    with open(tmp_path / "q.pkl", "wb") as f:
        pickle.dump((Q, [str(100 + j) for j in range(nq)]), f)
    argv = ["--query_reps", str(tmp_path / "q.pkl"), "--passage_reps", str(tmp_path / "corpus_emb*.pkl"),
            "--depth", str(depth), "--batch_size", "4", "--save_ranking_to", str(tmp_path / "rank.tsv"), "--save_text", "--quiet"]
    for var in ("RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "8")                       # a leftover from some scheduler
    assert not drv._is_rank_of_a_launch()
    status = drv._spawn_ranks(argv, 2, entry=[sys.executable, os.path.join(ROOT, "tests", "gloo_rank_main.py")])
    assert status == 0
    X, look = [], []
    for f in glob.glob(str(tmp_path / "corpus_emb*.pkl")):
        reps, lk = drv.pickle_load(f)
        X.append(np.asarray(reps, dtype=np.float32))
        look += lk
    rD, rI = O.topk_total_order(O.scores_mfma_order(Q, np.concatenate(X)), depth)
    drv.write_ranking(drv._lookup_strings(look, rI), rD, [str(100 + j) for j in range(nq)], str(tmp_path / "want.tsv"))
    assert (tmp_path / "rank.tsv").read_bytes() == (tmp_path / "want.tsv").read_bytes()
    # a rank that fails takes the launch down with its status
    bad = argv[:3] + [str(tmp_path / "nothing*.pkl")] + argv[4:]
    assert drv._spawn_ranks(bad, 2, entry=[sys.executable, os.path.join(ROOT, "tests", "gloo_rank_main.py")]) != 0


def test_assign_files_contiguous_and_balanced():
    from mural_amd.faiss_retriever.__main__ import _assign_files
    assert _assign_files([10, 10, 10, 10], 2) == [0, 0, 1, 1]
    assert _assign_files([100], 4) == [2] or _assign_files([100], 4)[0] in range(4)
    for rows, w in (([5, 1, 1, 1, 5, 7, 2], 3), ([1] * 20, 8), ([3, 3], 8), ([0, 0], 2)):
        o = _assign_files(rows, w)
        assert o == sorted(o) and all(0 <= r < w for r in o)
