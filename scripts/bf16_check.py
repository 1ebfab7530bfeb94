"""First GPU check of the bf16 scan + fp32 rescoring path (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mural_amd.faiss_retriever.retriever import FlatIPIndex


def gaussian(seed, shape):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def bf16_round(x):
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


for (n, d, nq, k) in [(5000, 768, 40, 10), (70000, 768, 300, 100), (300000, 128, 1000, 100), (200000, 768, 513, 500)]:
    X, Q = gaussian(1, (n, d)), gaussian(2, (nq, d))
    ref = FlatIPIndex(d, storage="f32"); ref.add(X)
    D0, I0 = ref.search(Q, k)
    a = FlatIPIndex(d, storage="bf16scan"); a.add(X)
    D1, I1 = a.search(Q, k)
    st = a.last_bf16_stats()
    print(n, d, nq, k, "bf16scan == f32:", np.array_equal(I0, I1), np.array_equal(D0.view(np.uint32), D1.view(np.uint32)), st, flush=True)
    Xr = bf16_round(X)
    ref2 = FlatIPIndex(d, storage="f32"); ref2.add(Xr)
    D2, I2 = ref2.search(Q, k)
    b = FlatIPIndex(d, storage="bf16"); b.add(X)
    try:
        D3, I3 = b.search(Q, k)
        print("   bf16 native == f32 on rounded rows:", np.array_equal(I2, I3), np.array_equal(D2.view(np.uint32), D3.view(np.uint32)), b.last_bf16_stats(), flush=True)
    except Exception as e:
        print("   bf16 native:", e, flush=True)
