"""Scan time of the headline workload (8.8M x 768 fp32, 6980 queries, top-k) on the fp32 scan, for A/B runs of
library builds in one gpurun call (IPX_LIB=<name of a build next to libipx.so>).

    python scripts/f32_headline.py [k] [rounds]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd import _lib
if os.environ.get("IPX_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IPX_LIB"])
from mural_amd.faiss_retriever.retriever import FlatIPIndex

k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows, nq = int(os.environ.get("ROWS", "8800000")), int(os.environ.get("NQ", "6980"))
dev = torch.device("cuda:0")
ix = FlatIPIndex(768, storage="f32")
ix.reserve(rows)
for b, lo in enumerate(range(0, rows, 200000)):
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + b)
    ix.add(torch.randn((200000, 768), generator=g, device=dev)[: rows - lo])
g = torch.Generator(device=dev)
g.manual_seed(4321)
Q = torch.randn((nq, 768), generator=g, device=dev)
ix.set_timing(True)
for kv in filter(None, os.environ.get("PARAMS", "").split(",")):      # e.g. PARAMS=f32_helpers=4,run_tiles=8
    name, val = kv.split("=")
    ix.set_param(name, int(val))
ts = []
for r in range(rounds + 1):
    D, I = ix.search_device(Q, k)
    torch.cuda.synchronize()
    if r:
        ts.append(ix.last_timing()["scan_ms"])
ts.sort()
fl = 2.0 * nq * rows * 768
print(f"{os.environ.get('IPX_LIB', 'libipx.so'):14s} {os.environ.get('PARAMS', ''):28s} k={k} scan min {ts[0]:.2f} med {ts[len(ts) // 2]:.2f} ms  "
      f"frac {fl / ts[0] / 1e9 / 157.3:.4f}  checksum {int(I.sum())}", flush=True)
