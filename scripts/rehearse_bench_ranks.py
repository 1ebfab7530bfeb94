"""Rehearsal of `bench.py --gpus N` on ONE GPU: N rank processes share GPU 0 and run bench.py's own main() — its N > 1
flow (row shards, barriers, max over ranks, cfg5 on every rank, rank 0's JSON line) — with two substitutions made here,
not in the product: the process group is gloo, and ShardedFlatIP's exchange goes through host memory (gloo's
device-tensor all-gather stalls when several processes keep one GPU busy: DESIGN section 6). The numbers it prints are
NOT measurements (the ranks time-share one GPU); what it shows is that every rank gets through and the line is well formed.

    python scripts/rehearse_bench_ranks.py 2 [bench.py arguments...]
"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, argv):
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("REHEARSAL_WATCHDOG", "300")), exit=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from mural_amd import sharded

    real_init = dist.init_process_group

    def init_gloo(backend=None, **kw):
        kw.pop("device_id", None)
        return real_init("gloo", **kw)

    dist.init_process_group = init_gloo

    def staged_search(self, q, k, exchange=None):
        if exchange is None:
            exchange = self.world > 1
        D, I = self.engine.search(q, k)
        if not exchange:
            return D, I
        nq = int(q.shape[0])
        block = self.engine.packed_block_bytes(nq, k)
        mine = torch.zeros(block, dtype=torch.uint8)
        mine[: nq * k * 8].view(torch.int64).view(nq, k).copy_(I)
        mine[nq * k * 8: nq * k * 12].view(torch.float32).view(nq, k).copy_(D)
        allb = torch.empty(self.world * block, dtype=torch.uint8)
        dist.all_gather_into_tensor(allb, mine, group=self.group)
        return self.engine.merge_packed(allb.cuda(), self.world, nq, k, k)

    sharded.ShardedFlatIP.search = staged_search
    sys.argv = ["bench.py", "--gpus", str(world)] + list(argv)
    import bench
    bench.main()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port, sys.argv[2:]), nprocs=world, join=True)
    print("rehearsal: all ranks finished", file=sys.stderr, flush=True)
