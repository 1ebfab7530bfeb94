"""bf16 scan + rescoring on data whose row norms spread (MURAL representations are unnormalised weighted sums of hidden
states, modeling/mul_grain.py:97-110): what the margin rule does with it.

    python scripts/norm_spread_probe.py [rows] [kinds]      # kinds: gauss,lognormal,l2norm,dups   SIGMA=0.5 NQ=6980 K=100,1000 KC=0,3000

Per data kind: outlier rows taken by the side list, candidates, queries repeated / answered by the exact path, scan /
rescoring / whole-search time, and bit-identity with the fp32 scan of the same index."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd.faiss_retriever.retriever import FlatIPIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
kinds = (sys.argv[2] if len(sys.argv) > 2 else "gauss,lognormal,l2norm").split(",")
sigma = float(os.environ.get("SIGMA", "0.5"))
nq = int(os.environ.get("NQ", "6980"))
ks = [int(v) for v in os.environ.get("K", "100,1000").split(",")]
storage = os.environ.get("STORAGE", "bf16scan")
d = 768
dev = torch.device("cuda:0")


def block(kind, g, n):
    x = torch.randn((n, d), generator=g, device=dev)
    if kind == "lognormal":
        x *= torch.exp(sigma * torch.randn((n, 1), generator=g, device=dev))
    elif kind == "l2norm":
        x /= x.norm(dim=1, keepdim=True)
    elif kind == "dups":
        x[1000:1040] = x[1000:1001]          # 40 exact duplicates per block
    return x


for kind in kinds:
    ix = FlatIPIndex(d, storage=storage)
    ix.reserve(rows)
    g = torch.Generator(device=dev); g.manual_seed(1)
    for lo in range(0, rows, 500000):
        x = block(kind, g, min(500000, rows - lo))
        ix.add(x.to(torch.bfloat16) if storage == "bf16" else x)
    del x
    ix.set_timing(True)
    Q = block("gauss" if kind != "l2norm" else kind, g, nq)
    for k, kc in [(k, kc) for k in ks for kc in [int(v) for v in os.environ.get("KC", "0").split(",")]]:
        ix.set_param("bf16_candidates", kc)
        res = {}
        for scan in ("bf16", "f32"):
            if storage == "bf16" and scan == "f32":
                continue
            ix.set_param("f32_scan", int(scan == "f32"))
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                D, I = ix.search_device(Q, k)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res[scan] = (D, I, dt, ix.last_timing(), ix.last_bf16_stats())
        D, I, dt, tm, st = res["bf16"]
        same = None
        if "f32" in res:
            same = bool(torch.equal(I, res["f32"][1])) and bool(torch.equal(D.view(torch.int32), res["f32"][0].view(torch.int32)))
        print(f"{kind:10s} rows {rows} nq {nq} k {k:5d}: outliers {st['outlier_rows']:4d} candidates {st['candidates']:5d} "
              f"repeated {st['fallback_queries']:5d} exact {st['exact_queries']:5d} | search {dt * 1e3:8.2f} ms scan {tm['scan_ms']:8.2f} "
              f"rescore {st['rescore_ms']:6.2f} | f32 search {res['f32'][2] * 1e3 if 'f32' in res else float('nan'):8.2f} ms | "
              f"bit-identical {same}", flush=True)
    ix.set_param("f32_scan", 0)
    del ix
    torch.cuda.empty_cache()
