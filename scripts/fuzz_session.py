"""Randomised session check: ONE index lives through a sequence of adds and searches of changing size, depth and
scan variant (buffers grow and are reused, thresholds / flags / counters from the previous call are still in
memory); every search must equal the oracle bit for bit. Usage: fuzz_session.py <sessions> <seed>."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as O
from mural_amd.faiss_retriever.retriever import FlatIPIndex

def bf16_round(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)

sessions, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
t0 = time.time()
for s in range(sessions):
    d = int(rng.choice([8, 24, 32, 40, 64, 100, 256]))
    storage = str(rng.choice(["f32", "bf16scan", "bf16"]))
    ix = FlatIPIndex(d, storage=storage)
    X = np.empty((0, d), np.float32)
    kmax = 1638 if storage == "bf16" else 2048
    for step in range(int(rng.integers(8, 25))):
        if X.shape[0] == 0 or rng.random() < 0.3:
            m = int(rng.choice([1, 100, 129, 1000, int(rng.integers(1, 20000))]))
            part = rng.standard_normal((m, d), dtype=np.float32)
            if rng.random() < 0.2:
                part = np.round(part)                          # ties
            X = np.concatenate([X, part])
            ix.add(part if rng.random() < 0.5 else torch.from_numpy(part).cuda())
        nq = int(rng.choice([1, 5, 32, 64, 65, 200, 257, int(rng.integers(1, 600))]))
        k = int(rng.choice([1, 10, 100, 500, 1000, kmax, int(rng.integers(1, kmax + 1))]))
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        for key in ("f32_scan", "no_small_batch", "query_parts", "max_splits"):
            ix.set_param(key, 0)
        if storage == "bf16scan" and rng.random() < 0.3:
            ix.set_param("f32_scan", 1)
        if rng.random() < 0.2:
            ix.set_param("no_small_batch", 1)
        if rng.random() < 0.3:
            ix.set_param("max_splits", int(rng.choice([1, 3, 17])))
        if storage != "f32":
            ix.set_param("query_parts", int(rng.choice([0, 1, 2])))
        Xe = bf16_round(X) if storage == "bf16" else X
        rD, rI = O.topk_total_order(O.scores_mfma_order(Q, Xe), k)
        if rng.random() < 0.5:
            D, I = ix.search(Q, k)
        else:
            D, I = ix.search_device(torch.from_numpy(Q).cuda(), k)
            ix.check()
            D, I = D.cpu().numpy(), I.cpu().numpy()
        if not (np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))):
            print("MISMATCH session", s, "step", step, dict(d=d, storage=storage, n=X.shape[0], nq=nq, k=k), flush=True)
            sys.exit(1)
    del ix
    if s % 5 == 4:
        print(f"{s + 1} sessions ok, {time.time() - t0:.0f} s", flush=True)
print("all", sessions, "sessions ok")
