"""Stress of the longest candidate list (4096, hi + lo parts, 256-query tile) on dirty device memory: every iteration
fills freed memory with random bytes, builds a fresh index and searches with forced parameters; results against the
oracle, bit for bit. Usage: stress_deep_lists.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from mural_amd.faiss_retriever.retriever import FlatIPIndex
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(11)
bad = {}
for it in range(iters):
    junk = torch.randint(-2**31, 2**31 - 1, (int(rng.integers(64, 512)) * 1024 * 1024 // 4,), dtype=torch.int32, device="cuda")
    del junk
    torch.cuda.empty_cache()
    n, d, nq = int(rng.integers(1500, 6000)), int(rng.choice([48, 192])), int(rng.choice([65, 130, 300]))
    kind = str(rng.choice(["ramp", "gauss"]))
    if kind == "ramp":
        X = np.zeros((n, d), np.float32); X[:, 0] = np.arange(n, dtype=np.float32) * (1 if rng.random() < 0.5 else -1)
    else:
        X = rng.standard_normal((n, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    ix = FlatIPIndex(d, storage="bf16scan")
    step = -(-n // 3)
    for lo in range(0, n, step):
        ix.add(X[lo:lo + step])
    ix.set_param("max_splits", int(rng.choice([0, 5])))
    for k, parts, kc in ((2048, 2, 4096), (1024, 2, 4096), (2048, 1, 4096), (1024, 2, 0), (700, 2, 4096)):
        with np.errstate(all="ignore"):
            rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), k)
        ix.set_param("query_parts", parts); ix.set_param("bf16_candidates", kc)
        D, I = ix.search(Q, k)
        ok = np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
        key = (k, parts, kc)
        bad.setdefault(key, 0)
        if not ok:
            bad[key] += 1
            rows = sorted(set(np.argwhere(I != rI)[:, 0].tolist()))
            r0 = rows[0]; j = int(np.argmax(I[r0] != rI[r0]))
            print("MISMATCH it", it, dict(n=n, d=d, nq=nq, kind=kind), key, "queries", rows[:8], "rank", j, "got", I[r0, j:j + 4].tolist(),
                  "want", rI[r0, j:j + 4].tolist(), ix.last_bf16_stats(), flush=True)
    del ix
print("mismatches per (k, parts, forced candidates):", bad, "in", iters, "iterations")
