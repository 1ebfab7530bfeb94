"""Diagnostic: scan time of the bf16 kernel for a few shapes, optionally with an ablation build
(IPX_LIB=libipx_abl.so). Prints cycles per K-stage per workgroup at 2.4 GHz."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd import _lib
if os.environ.get("IPX_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IPX_LIB"])
from mural_amd.faiss_retriever.retriever import FlatIPIndex

dev = torch.device("cuda:0")
for rows, nq, ms in [(16000, 2048, 1), (2000000, 7168, 1), (2000000, 7168, 0)]:
    ix = FlatIPIndex(768, storage="bf16")
    ix.reserve(rows)
    g = torch.Generator(device=dev); g.manual_seed(1)
    for lo in range(0, rows, 200000):
        ix.add(torch.randn((min(200000, rows - lo), 768), generator=g, device=dev))
    Q = torch.randn((nq, 768), generator=g, device=dev)
    ix.set_timing(True)
    ix.set_param("max_splits", ms)
    ix.set_param("query_parts", int(os.environ.get("QP", "0")))
    ts = []
    for it in range(4):
        try:
            ix.search_device(Q, 100)
        except Exception as e:
            pass
        torch.cuda.synchronize()
        ts.append(ix.last_timing()["scan_ms"])
    geo = ix.last_geometry()
    t = min(ts[1:])
    tiles_per_wg = (rows / 256) * (nq / 256) / geo["workgroups"]
    print(f"rows {rows} nq {nq} wgs {geo['workgroups']} S {geo['splits']}: scan {t:.2f} ms, "
          f"{t * 1e-3 * 2.4e9 / (tiles_per_wg * 24):.0f} cycles/stage, {2 * rows * nq * 768 / t / 1e9:.0f} TFLOP/s", flush=True)
    del ix
