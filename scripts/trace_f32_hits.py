import sys, os
import numpy as np
import torch
sys.path.insert(0, '.')
from mural_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libipx_trace.so")
from mural_amd.faiss_retriever.retriever import FlatIPIndex
n, d, nq, k = 8_800_000, 768, 6980, int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda:0')
ix = FlatIPIndex(d, 0); ix.reserve(n)
for b, lo in enumerate(range(0, n, 200_000)):
    g = torch.Generator(device=dev); g.manual_seed(1234 + b)
    ix.add(torch.randn((200_000, d), generator=g, device=dev)[: n - lo])
g = torch.Generator(device=dev); g.manual_seed(4321)
Q = torch.randn((nq, d), generator=g, device=dev)
ix.set_param("trace", 1); ix.set_timing(True)
for kv in filter(None, os.environ.get("PARAMS", "").split(",")):
    ix.set_param(kv.split("=")[0], int(kv.split("=")[1]))
for it in range(2):
    ix.search_device(Q, k); torch.cuda.synchronize()
print(ix.last_timing(), ix.last_geometry())
tr = ix.read_trace().astype(np.uint64)
dur = np.full(len(tr), ix.last_timing()["scan_ms"] * 1000.0)
ph = [(tr[:, 0] & np.uint64(0xffffffff)), (tr[:, 0] >> np.uint64(32)), (tr[:, 1] & np.uint64(0xffffffff)), (tr[:, 1] >> np.uint64(32))]
print("wave 0, us per wg in the selections: load %.0f, publish %.0f, k-th %.0f, ties + move %.0f" % tuple(float(x.astype(np.float64).mean()) / 100.0 for x in ph))
epi = (tr[:, 2] >> np.uint64(32)).astype(np.float64) / 100.0
w = tr[:, 3]
tiles = (w >> np.uint64(48)).astype(np.int64); nc = ((w >> np.uint64(40)) & np.uint64(0xff)).astype(np.int64)
hits = ((w >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.int64); app = (w & np.uint64(0xfffff)).astype(np.int64)
print("wgs", len(w), "dur us p50 %.0f" % np.median(dur))
print("tiles/wg p50 %d; episodes/wg mean %.1f max %d; us in episodes/wg mean %.0f p90 %.0f max %.0f (%.2f %% of dur)" % (np.median(tiles), nc.mean(), nc.max(), epi.mean(), np.percentile(epi, 90), epi.max(), 100 * epi.sum() / dur.sum()))
print("wave 0: hit groups per tile %.3f, appends per tile %.3f, appends per hit group %.2f" % (hits.sum() / tiles.sum(), app.sum() / tiles.sum(), app.sum() / max(1, hits.sum())))
