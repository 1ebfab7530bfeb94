"""Small-batch regime on a plain fp32 index (storage 0): 1..256 queries against `rows` x 768 fp32 rows.
Checks that the small-tile scan returns the same bits as the 256-query tile, then prints per-search time
(device-resident inputs), the kernel split and the corpus stream rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_841_823
ks = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100]
d = 768
dev = torch.device("cuda:0")
ix = FlatIPIndex(d, storage="f32")
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 500000):
    x = torch.randn((min(500000, rows - lo), d), generator=g, device=dev)
    ix.add(x)
del x
ix.set_timing(True)
if os.environ.get('MAX_SPLITS'):
    ix.set_param('max_splits', int(os.environ['MAX_SPLITS']))
NQS = [int(a) for a in os.environ.get('NQS', '1,8,32,33,64,65,128,192,256').split(',')]
gb = rows * d * 4 / 1e9
for k in ks:
    for nq in NQS:
        Q = torch.randn((nq, d), generator=g, device=dev)
        ix.set_param("no_small_batch", 1)
        D0, I0 = ix.search_device(Q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix.search_device(Q, k)
        torch.cuda.synchronize()
        big = time.perf_counter() - t0
        ix.set_param("no_small_batch", 0)
        ix.set_param("small_f32_limit", 256)
        for _ in range(2):
            D, I = ix.search_device(Q, k)
        same = bool(torch.equal(D, D0) and torch.equal(I, I0))
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 5
        for _ in range(n):
            ix.search_device(Q, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        tm, geo = ix.last_timing(), ix.last_geometry()
        print(f"k {k} nq {nq:4d}: {dt*1e3:7.3f} ms/search (256-tile kernel {big*1e3:7.3f}) scan {tm['scan_ms']:.3f} merge {tm['merge_ms']:.3f} "
              f"pack {tm['pack_ms']:.3f} | {gb:.1f} GB -> {gb/tm['scan_ms']:.2f} TB/s in scan | wgs {geo['workgroups']} S {geo['splits']} "
              f"| bit-identical {same}", flush=True)
        assert same
