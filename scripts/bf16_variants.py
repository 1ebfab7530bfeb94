"""A/B of the large-batch bf16 scan's variants in ONE process (interleaved rounds, same index, same
queries): mapping (query-block-major / split-major), issue order behind the stage barrier, non-temporal
corpus pieces, wave count, MFMA shape. Prints min / median scan ms per variant and checks that every
variant returns the same bits.

    python scripts/bf16_variants.py [rows] [queries] [rounds]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mural_amd import _lib
if os.environ.get("IPX_LIB"):      # an ablation / diagnostic build next to libipx.so
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IPX_LIB"])
from mural_amd.faiss_retriever.retriever import FlatIPIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
k = int(os.environ.get("K", "100"))
dev = torch.device("cuda:0")
ix = FlatIPIndex(768, storage=os.environ.get("STORAGE", "bf16"))
ix.reserve(rows)
for b, lo in enumerate(range(0, rows, 200000)):
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + b)
    blk = torch.randn((200000, 768), generator=g, device=dev)[: rows - lo]
    if os.environ.get("ZERO") == "1":      # diagnostic: all-zero operands (the clock the chip holds without data toggling)
        blk.zero_()
    ix.add(blk)
g = torch.Generator(device=dev)
g.manual_seed(4321)
Q = torch.randn((nq, 768), generator=g, device=dev)
if os.environ.get("ZERO") == "1":
    Q.zero_()
ix.set_timing(True)

# (name, params)
variants = [("base", {})]
for spec in os.environ.get("VARIANTS", "map1;v1;v2;v3;v4;map1,v1;map1,v2;map1,v3;map1,v4;map1,v5;w8;w8,map1;s16;s16,map1").split(";"):
    prm = {}
    for tok in spec.split(","):
        if tok.startswith("map"):
            prm["bf16_map"] = int(tok[3:])
        elif tok.startswith("v"):
            prm["bf16_variant"] = int(tok[1:])
        elif tok == "w8":
            prm["bf16_waves"] = 8
        elif tok == "w12":
            prm["bf16_waves"] = 12
        elif tok == "s16":
            prm["bf16_shape16"] = 1
        elif tok == "s32":
            prm["bf16_shape16"] = 0
        elif tok.startswith("pp"):
            prm["bf16_pp"] = int(tok[2:])
        elif tok.startswith("kc"):
            prm["bf16_candidates"] = int(tok[2:])
        elif tok.startswith("ring"):
            prm["bf16_ring"] = int(tok[4:])
        elif tok.startswith("occ"):
            prm["bf16_occ"] = int(tok[3:])
        elif tok.startswith("ms"):
            prm["max_splits"] = int(tok[2:])
    variants.append((spec, prm))
defaults = {"bf16_map": 0, "bf16_variant": 0, "bf16_waves": 0, "bf16_shape16": 1, "max_splits": 0, "bf16_pp": 0, "bf16_ring": 0, "bf16_occ": 1, "bf16_candidates": 0}

times = {name: [] for name, _ in variants}
ref = None
for r in range(rounds + 1):
    for name, prm in variants:
        for key, val in {**defaults, **prm}.items():
            ix.set_param(key, val)
        D, I = ix.search_device(Q, k)
        torch.cuda.synchronize()
        ix.check()
        if r == 0:      # warm-up round doubles as the parity check between variants
            if ref is None:
                ref = (D.clone(), I.clone())
            else:
                same = bool(torch.equal(I, ref[1])) and bool(torch.equal(D.view(torch.int32), ref[0].view(torch.int32)))
                if not same:
                    print(f"!! variant {name}: results differ from base", flush=True)
            print(f"   {name}: candidates {ix.last_bf16_stats()['candidates']}, repeated {ix.last_bf16_stats()['fallback_queries']}, "
                  f"queue cap {ix.last_geometry()['queue_cap']}", flush=True)
            continue
        times[name].append(ix.last_timing()["scan_ms"])
flops = 2.0 * nq * rows * 768
base = min(times["base"])
for name, _ in variants:
    t = sorted(times[name])
    print(f"{name:14s} min {t[0]:8.2f} ms  med {t[len(t) // 2]:8.2f} ms  {flops / t[0] / 1e9:7.0f} TFLOP/s  "
          f"frac {flops / t[0] / 1e9 / 2500:.3f}  vs base {base / t[0]:.3f}x", flush=True)
print("geometry", ix.last_geometry(), "bf16", ix.last_bf16_stats(), flush=True)
