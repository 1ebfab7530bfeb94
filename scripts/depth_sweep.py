"""Deep searches on the fp32+bf16 index (storage 1): for each depth k, the fp32 scan against the bf16 scan with
one / two query parts and the longest candidate list — time per search and how many queries the margin rule
sent to the repeat pass. Feeds the depth policy in scan_bf16_path (ipx.hip).

    python scripts/depth_sweep.py [rows] [queries]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
dev = torch.device("cuda:0")
ix = FlatIPIndex(768, storage="bf16scan")
ix.reserve(rows)
for b, lo in enumerate(range(0, rows, 200000)):
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + b)
    ix.add(torch.randn((200000, 768), generator=g, device=dev)[: rows - lo])
g = torch.Generator(device=dev)
g.manual_seed(4321)
Q = torch.randn((nq, 768), generator=g, device=dev)


def run(k, **prm):
    defaults = {"f32_scan": 0, "query_parts": 0, "bf16_candidates": 0, "bf16_waves": 0, "bf16_ring": 0}
    for key, val in {**defaults, **prm}.items():
        ix.set_param(key, val)
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        D, I = ix.search_device(Q, k)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    ix.check()
    st = ix.last_bf16_stats()
    return best, st, (D, I)


for k in [int(x) for x in os.environ.get("KS", "500,700,800,900,1000,1024").split(",")]:
    t_f, _, ref = run(k, f32_scan=1)
    line = [f"k={k:5d}  fp32 {t_f:8.1f} ms"]
    for name, prm in (("default", {}), ("1part/2048", {"query_parts": 1, "bf16_candidates": 2048}),
                      ("2part/2048", {"query_parts": 2, "bf16_candidates": 2048}),
                      ("1part/3072", {"query_parts": 1, "bf16_candidates": 3072}),
                      ("1part/4096", {"query_parts": 1, "bf16_candidates": 4096})):
        if prm.get("bf16_candidates", 0) and prm["bf16_candidates"] < k:
            continue
        t, st, out = run(k, **prm)
        same = bool(torch.equal(out[1], ref[1])) and bool(torch.equal(out[0].view(torch.int32), ref[0].view(torch.int32)))
        line.append(f"{name} {t:8.1f} ms kc={st['candidates']} repeat={st['fallback_queries']}{'' if same else ' DIFF'}")
    print("  |  ".join(line), flush=True)
