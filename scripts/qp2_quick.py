import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd import _lib
if os.environ.get("IPX_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IPX_LIB"])
from mural_amd.faiss_retriever.retriever import FlatIPIndex
dev = torch.device("cuda:0"); rows, nq = int(os.environ.get("ROWS", "2000000")), 6980
ix = FlatIPIndex(768, storage="bf16scan"); ix.reserve(rows)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, rows, 200000): ix.add(torch.randn((200000, 768), generator=g, device=dev))
Q = torch.randn((nq, 768), generator=g, device=dev); ix.set_timing(True)
for parts in (1, 2):
    ix.set_param("query_parts", parts)
    for _ in range(3): ix.search_device(Q, 100); torch.cuda.synchronize()
    print(os.environ.get("IPX_LIB", "libipx.so"), "parts", parts, "scan ms", round(ix.last_timing()["scan_ms"], 2), flush=True)
