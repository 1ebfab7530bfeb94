"""8-wave vs 16-wave large-batch bf16 scan on one index: same bits, scan time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd import _lib as _l
if os.environ.get('IPX_LIB'):
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), os.environ['IPX_LIB'])
from mural_amd.faiss_retriever.retriever import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
dev = torch.device("cuda:0")
ix = FlatIPIndex(768, storage="bf16scan")
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 400000):
    ix.add(torch.randn((min(400000, rows - lo), 768), generator=g, device=dev))
Q = torch.randn((nq, 768), generator=g, device=dev)
ix.set_timing(True)
for k in [int(a) for a in os.environ.get('KS', '100,1000').split(',')]:
    res = {}
    for waves in (8, 16, 116, 8, 16, 116):          # 116 = 16 waves with the 16x16x32 MFMA shape (one query part)
        ix.set_param("bf16_waves", waves % 100)
        ix.set_param("bf16_shape16", waves // 100)
        ts = []
        for it in range(3):
            D, I = ix.search_device(Q, k); torch.cuda.synchronize()
            ts.append(ix.last_timing()["scan_ms"])
        st = ix.last_bf16_stats()
        if waves in res:
            assert torch.equal(res[waves][0], D) and torch.equal(res[waves][1], I)
        res[waves] = (D, I)
        print(f"k {k} waves {waves}: scan {min(ts):.2f} ms ({2*rows*nq*768/min(ts)/1e9:.0f} TFLOP/s) candidates {st['candidates']} repeated {st['fallback_queries']}", flush=True)
    print("  same bits:", all(bool(torch.equal(res[8][0], res[w][0]) and torch.equal(res[8][1], res[w][1])) for w in res), flush=True)
