"""bf16 scan with 1 / 2 query parts at several depths (storage bf16scan, 8.8M x 768, 6980 queries)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows, nq, d = 8_800_000, 6980, 768
dev = torch.device("cuda:0")
ix = FlatIPIndex(d, storage="bf16scan"); ix.reserve(rows)
g = torch.Generator(device=dev)
for b in range(rows // 200000):
    g.manual_seed(1234 + b); ix.add(torch.randn((200000, d), generator=g, device=dev))
g.manual_seed(4321); Q = torch.randn((nq, d), generator=g, device=dev)
ix.set_timing(True)
for k, parts, kc in [(100, 0, 0), (100, 2, 0), (1000, 0, 0), (10, 0, 0)]:
    ix.set_param("query_parts", parts); ix.set_param("bf16_candidates", kc)
    ix.search_device(Q, k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): ix.search_device(Q, k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    tm, bf, geo = ix.last_timing(), ix.last_bf16_stats(), ix.last_geometry()
    print(f"k {k} parts {parts}: {dt*1e3:.1f} ms/search scan {tm['scan_ms']:.1f} merge {tm['merge_ms']:.2f} rescore {bf['rescore_ms']:.2f} "
          f"kc {bf['candidates']} repeated {bf['fallback_queries']} cap {geo['queue_cap']}", flush=True)
ix.set_param("f32_scan", 1)
for k in (1000,):
    ix.search_device(Q, k); torch.cuda.synchronize()
    t0 = time.perf_counter(); ix.search_device(Q, k); torch.cuda.synchronize()
    print(f"k {k} fp32 scan: {(time.perf_counter()-t0)*1e3:.1f} ms/search")
