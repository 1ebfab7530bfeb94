"""Deep searches (k beyond IPX_MAX_K) on the headline corpus: one scan with unsorted per-split lists + deep_finish_kernel
(k <= 16384), one scan per 16384 ranks beyond; fp32 scan and bf16 scan + rescoring of the same index, bitwise compared.

    python scripts/deep_bench.py [rows] [queries]      # KS=1000,5000,8192,16384,20000"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
ks = [int(v) for v in os.environ.get("KS", "1000,5000,8192,16384,20000").split(",")]
d = 768
dev = torch.device("cuda:0")
ix = FlatIPIndex(d, storage=os.environ.get("STORAGE", "bf16scan"))
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 500000):
    ix.add(torch.randn((min(500000, rows - lo), d), generator=g, device=dev))
Q = torch.randn((nq, d), generator=g, device=dev)
ix.set_timing(True)
for k in ks:
    res = {}
    for scan in ("f32", "bf16"):
        ix.set_param("f32_scan", int(scan == "f32"))
        D = I = None
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            D, I = ix.search_device(Q, k)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[scan] = (D, I, dt, ix.last_timing() if k <= 16384 else None, ix.last_bf16_stats(), ix.last_geometry())
    same = bool(torch.equal(res["f32"][1], res["bf16"][1])) and bool(torch.equal(res["f32"][0].view(torch.int32), res["bf16"][0].view(torch.int32)))
    for scan in ("f32", "bf16"):
        D, I, dt, tm, st, geo = res[scan]
        print(f"k {k:6d} {scan:5s}: search {dt * 1e3:9.2f} ms  {nq / dt:9.0f} q/s  timing {tm}  stats {st}  geometry {geo}", flush=True)
    print(f"k {k:6d} bit-identical {same}; sorted {bool((res['f32'][0][:, :-1] >= res['f32'][0][:, 1:]).all())}", flush=True)
    del res
    torch.cuda.empty_cache()
