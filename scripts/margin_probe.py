"""Candidate-list length vs queries sent to the repeat pass (margin rule, rescore_kernel), per regime.

    python scripts/margin_probe.py [rows] [storage]     # SPECS="nq:k:parts:kc;..." MARGIN_POST=0|1
Small batches (nq <= 64) are run over 20 different batches and the repeats summed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
storage = sys.argv[2] if len(sys.argv) > 2 else "bf16scan"
d = 768
dev = torch.device("cuda:0")
ix = FlatIPIndex(d, storage=storage)
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 500000):
    x = torch.randn((min(500000, rows - lo), d), generator=g, device=dev)
    ix.add(x.to(torch.bfloat16) if storage == "bf16" else x)
del x
ix.set_timing(True)
ix.set_param("margin_post", int(os.environ.get("MARGIN_POST", "1")))
Qall = torch.randn((6980, d), generator=g, device=dev)
for spec in os.environ["SPECS"].split(";"):
    nq, k, parts, kc = (int(v) for v in spec.split(":"))
    ix.set_param("query_parts", parts)
    ix.set_param("bf16_candidates", kc)
    batches = [Qall[:nq]] if nq > 64 else [Qall[i * nq:(i + 1) * nq].contiguous() for i in range(20)]
    fb = 0; scan = []; resc = []
    for Q in batches:
        for rep in range(2):
            ix.search_device(Q, k)
        torch.cuda.synchronize()
        fb += ix.last_bf16_stats()["fallback_queries"]
        scan.append(ix.last_timing()["scan_ms"]); resc.append(ix.last_bf16_stats()["rescore_ms"])
    print(f"nq {nq:5d} k {k:5d} parts {parts} kc {ix.last_bf16_stats()['candidates']:5d}: repeated {fb:5d} of {nq * len(batches):5d}  "
          f"scan {sorted(scan)[len(scan)//2]:8.3f} ms  rescore {sorted(resc)[len(resc)//2]:7.3f} ms", flush=True)
