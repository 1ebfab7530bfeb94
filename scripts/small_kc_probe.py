"""small batches on the bf16 copy: candidate-list length vs repeated queries / latency for deeper k."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows = 8_800_000
dev = torch.device("cuda:0")
ix = FlatIPIndex(768, storage="bf16scan")
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 400000):
    ix.add(torch.randn((min(400000, rows - lo), 768), generator=g, device=dev))
ix.set_timing(True)
for nq in (1, 32, 64):
    Q = torch.randn((nq, 768), generator=g, device=dev)
    for k in (100, 102, 103, 128, 150, 256, 257):
        for kc in (0,):
            ix.set_param("bf16_candidates", kc)
            for it in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ix.search_device(Q, k); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            st = ix.last_bf16_stats()
            print(f"nq {nq} k {k} kc {st['candidates']}: {dt*1e3:.2f} ms/search, repeated {st['fallback_queries']}", flush=True)
