"""Interleaved A/B of small-batch scan parameters in ONE process (same index, same queries, rounds interleaved so that
thermal drift hits every arm alike): prints the median scan ms per arm and batch size.

    CONFIGS="query_parts=1;query_parts=1,bf16_candidates=200" NQ=16,64 python scripts/small_ab.py [rows] [storage]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd import _lib
if os.environ.get("IPX_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IPX_LIB"])
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
storage = sys.argv[2] if len(sys.argv) > 2 else "bf16scan"
d, k = 768, int(os.environ.get("K", "100"))
dev = torch.device("cuda:0")
ix = FlatIPIndex(d, storage=storage)
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 500000):
    x = torch.randn((min(500000, rows - lo), d), generator=g, device=dev)
    ix.add(x.to(torch.bfloat16) if storage == "bf16" else x)
del x
ix.set_timing(True)
arms = [("default", {})]
for spec in filter(None, os.environ.get("CONFIGS", "").split(";")):
    arms.append((spec, {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",")}))
keys = sorted({key for _, prm in arms for key in prm})
rounds = int(os.environ.get("ROUNDS", "5"))
for nq in [int(x) for x in os.environ.get("NQ", "1,16,64").split(",")]:
    Q = torch.randn((nq, d), generator=g, device=dev)
    t = {name: [] for name, _ in arms}
    fb = {name: 0 for name, _ in arms}
    ref = None
    for r in range(rounds + 1):
        for name, prm in arms:
            for key in keys:
                ix.set_param(key, prm.get(key, 0))
            for _ in range(4):
                D, I = ix.search_device(Q, k)
                torch.cuda.synchronize()
                if r:
                    t[name].append(ix.last_timing()["scan_ms"])
            fb[name] = ix.last_bf16_stats()["fallback_queries"]
            if ref is None:
                ref = (D.clone(), I.clone())
            elif not (torch.equal(I, ref[1]) and torch.equal(D.view(torch.int32), ref[0].view(torch.int32))):
                print(f"!! {name}: results differ from default", flush=True)
    gb = rows * d * 2 / 1e9
    for name, _ in arms:
        v = sorted(t[name]); med = v[len(v) // 2]
        print(f"nq {nq:3d} {name:40s} scan med {med:.3f} min {v[0]:.3f} ms  {gb / med / 8:.3f} of 8 TB/s  kc {ix.last_bf16_stats()['candidates'] if name == arms[-1][0] else '-'} fb {fb[name]}", flush=True)
