"""Diagnostic: where scan_bf16_dx_kernel's wave 0 spends its cycles outside the K-pair loop (append path,
episodes), per workgroup. python scripts/trace_dx.py [rows] [queries]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
dev = torch.device("cuda:0")
ix = FlatIPIndex(768, storage=os.environ.get("STORAGE", "bf16"))
ix.reserve(rows)
for b, lo in enumerate(range(0, rows, 200000)):
    g = torch.Generator(device=dev); g.manual_seed(1234 + b)
    ix.add(torch.randn((200000, 768), generator=g, device=dev)[: rows - lo])
g = torch.Generator(device=dev); g.manual_seed(4321)
Q = torch.randn((nq, 768), generator=g, device=dev)
ix.set_timing(True); ix.set_param("bf16_pp", int(os.environ.get("PP", "3"))); ix.set_param("trace", 1)
for _ in range(2):
    ix.search_device(Q, int(os.environ.get("K", "100"))); torch.cuda.synchronize()
print("scan ms", ix.last_timing()["scan_ms"], ix.last_geometry())
tr = ix.read_trace().astype(np.float64)
tot, hit, epi = tr[:, 0], tr[:, 1], tr[:, 2]
cnt = ix.read_trace()[:, 3]
hits, eps = (cnt >> np.uint64(32)).astype(np.int64), (cnt & np.uint64(0xffffffff)).astype(np.int64)
print(f"total cycles: mean {tot.mean():.3e} min {tot.min():.3e} max {tot.max():.3e}")
print(f"append path: {100 * hit.sum() / tot.sum():.2f} % of cycles; tiles with appends per workgroup: mean {hits.mean():.0f} (min {hits.min()}, max {hits.max()})")
print(f"episodes   : {100 * epi.sum() / tot.sum():.2f} % of cycles; per workgroup: mean {eps.mean():.1f} (min {eps.min()}, max {eps.max()}); cycles each {epi.sum() / max(eps.sum(), 1):.0f}")
