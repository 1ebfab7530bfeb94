"""k=1000 on the bf16 scan: one query part with the longest list vs hi+lo parts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows, nq = 8_800_000, 6980
dev = torch.device("cuda:0")
ix = FlatIPIndex(768, storage="bf16scan")
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 400000):
    ix.add(torch.randn((min(400000, rows - lo), 768), generator=g, device=dev))
Q = torch.randn((nq, 768), generator=g, device=dev)
ix.set_timing(True)
for k in (250, 400, 500, 640, 641, 1000):
    for qp, kc in ((0, 0), (2, 0)):
        ix.set_param("query_parts", qp); ix.set_param("bf16_candidates", kc)
        for it in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            D, I = ix.search_device(Q, k); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        st = ix.last_bf16_stats()
        print(f"k {k} parts {qp} kc {st['candidates']}: {dt*1e3:.1f} ms/search, scan {ix.last_timing()['scan_ms']:.1f}, rescore {st['rescore_ms']:.1f}, repeated {st['fallback_queries']}", flush=True)
