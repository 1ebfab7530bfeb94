"""Diagnostic: per-workgroup cycle split of scan_bf16_kernel (trace records)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mural_amd import _lib as _l
if os.environ.get('IPX_LIB'):
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), os.environ['IPX_LIB'])
from mural_amd.faiss_retriever.retriever import FlatIPIndex
dev = torch.device("cuda:0")
rows, nq = int(sys.argv[1]), int(sys.argv[2])
storage = sys.argv[3] if len(sys.argv) > 3 else "bf16"
ix = FlatIPIndex(768, storage=storage)
ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 200000):
    ix.add(torch.randn((min(200000, rows - lo), 768), generator=g, device=dev))
Q = torch.randn((nq, 768), generator=g, device=dev)
ix.set_timing(True); ix.set_param("trace", 1); ix.set_param("query_parts", int(os.environ.get("QP", "0")))
for kv in filter(None, os.environ.get("PARAMS", "").split(",")):      # e.g. PARAMS=bf16_ring=2,bf16_map=7
    name, val = kv.split("=")
    ix.set_param(name, int(val))
for it in range(2):
    ix.search_device(Q, int(os.environ.get("K", "100"))); torch.cuda.synchronize()
import ctypes as C
from mural_amd import _lib
n = ix.last_geometry()["workgroups"]
raw = np.zeros((2 * n, 4), dtype=np.uint64)
_lib.check(_lib.lib().ipx_read_trace(ix._h, C.c_void_p(raw.ctypes.data), 2 * n))
tr = raw[:n].astype(np.float64); ex = raw[n:].astype(np.float64)
tot, epi, srt = tr[:, 0], tr[:, 1], tr[:, 2]
app, eps = (tr[:, 3].astype(np.uint64) >> 16).astype(np.float64), (tr[:, 3].astype(np.uint64) & 0xffff).astype(np.float64)
geo = ix.last_geometry(); tiles = (rows / 256) * np.ceil(nq / 256) / geo["workgroups"]
print(f"scan {ix.last_timing()['scan_ms']:.2f} ms wgs {geo['workgroups']} S {geo['splits']} cap {geo['queue_cap']} tiles/wg {tiles:.0f}")
print(f"per WG mean: total {tot.mean():.3e} ticks, filter {epi.mean()/tot.mean():.3f}, compaction {srt.mean()/tot.mean():.3f}, "
      f"appends {app.mean():.0f} ({app.mean()/tiles:.1f}/tile), episodes {eps.mean():.1f} (max {eps.max():.0f})")
print(f"episode split (wave 0, share of total): wait for the other waves' filters {ex[:,0].mean()/tot.mean():.3f}, own compaction {ex[:,1].mean()/tot.mean():.3f}, "
      f"wait for the other waves' compaction {ex[:,2].mean()/tot.mean():.3f}")
