"""Small-batch (HBM-bound) regime: BASELINE.json configs[4] per-GPU shard — bf16 rows, 1..64 queries.
Prints per-search wall time (device-resident inputs), the kernel split and the corpus stream rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd import _lib
if os.environ.get("IPX_LIB"):      # an ablation / diagnostic build next to libipx.so
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IPX_LIB"])
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
storage = sys.argv[2] if len(sys.argv) > 2 else "bf16"
d, k = 768, 100
dev = torch.device("cuda:0")
ix = FlatIPIndex(d, storage=storage)
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 500000):
    x = torch.randn((min(500000, rows - lo), d), generator=g, device=dev)
    ix.add(x.to(torch.bfloat16) if storage == "bf16" else x)
del x
ix.set_timing(True)
for kv in filter(None, os.environ.get("PARAMS", "").split(",")):      # e.g. PARAMS=query_parts=1,bf16_candidates=256
    name, val = kv.split("=")
    ix.set_param(name, int(val))
for nq in [int(x) for x in os.environ.get("NQ", "1,4,16,32,64,65,128,256").split(",")]:
    Q = torch.randn((nq, d), generator=g, device=dev)
    if os.environ.get("QNAN") == "1":      # diagnostic: no score is ever admitted (the scan's floor without appends)
        Q[:, 0] = float("nan")
    for _ in range(2):
        ix.search_device(Q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 5
    for _ in range(n):
        ix.search_device(Q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    tm, bf, geo = ix.last_timing(), ix.last_bf16_stats(), ix.last_geometry()
    gb = rows * d * 2 / 1e9
    print(f"nq {nq:4d}: {dt*1e3:7.3f} ms/search ({nq/dt:9.0f} q/s) scan {tm['scan_ms']:.3f} merge {tm['merge_ms']:.3f} "
          f"rescore {bf['rescore_ms']:.3f} pack {tm['pack_ms']:.3f} | corpus {gb:.1f} GB -> {gb/tm['scan_ms']:.2f} TB/s in scan, "
          f"{gb/dt/1e3:.2f} TB/s end to end | wgs {geo['workgroups']} S {geo['splits']} kc {bf['candidates']} fb {bf['fallback_queries']}", flush=True)
