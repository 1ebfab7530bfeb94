"""Randomised differential check of ipx_merge_topk (the reducer / shard-merge kernel) and of the search
entry-point variants (id base, aspect-fused queries with exactly representable weights, pooled queries)
against the oracle. Usage: fuzz_merge.py <cases> <seed>."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from mural_amd import _lib
from mural_amd.faiss_retriever.retriever import FlatIPIndex
FLT_MAX = np.finfo(np.float32).max

def gpu_merge(Ds, Is, k_out):
    S, nq, k_in = Ds.shape
    Ds, Is = np.ascontiguousarray(Ds, np.float32), np.ascontiguousarray(Is, np.int64)
    oD, oI = np.empty((nq, k_out), np.float32), np.empty((nq, k_out), np.int64)
    _lib.check(_lib.lib().ipx_merge_topk(C.c_void_p(Ds.ctypes.data), C.c_void_p(Is.ctypes.data), S, nq, k_in, k_out,
                                         C.c_void_p(oD.ctypes.data), C.c_void_p(oI.ctypes.data), 0, 0, None))
    return oD, oI

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
t0 = time.time()
for c in range(cases):
    # ---- merge ----
    S = int(rng.choice([1, 2, 3, 8, 16, 17, 40, int(rng.integers(1, 64))]))
    nq = int(rng.integers(1, 40))
    k_in = int(rng.choice([1, 2, 10, 100, 1000, 2048, int(rng.integers(1, 2049))]))
    k_out = int(rng.choice([1, 10, 100, 1000, 2048, k_in, int(rng.integers(1, 2049))]))
    if S * nq * k_in > 4_000_000:
        nq = max(1, 4_000_000 // (S * k_in))
    kind = rng.choice(["gauss", "quant", "sorted"])
    Ds = rng.standard_normal((S, nq, k_in)).astype(np.float32)
    if kind == "quant":
        Ds = np.round(Ds * 2) / 2
    if kind == "sorted":
        Ds = -np.sort(-Ds, axis=2)
    Is = rng.permutation(S * nq * k_in).reshape(S, nq, k_in).astype(np.int64) * int(rng.choice([1, 7919])) + int(rng.choice([0, -3, 10 ** 12]))
    holes = rng.random(Ds.shape) < float(rng.choice([0.0, 0.01, 0.5, 0.98]))
    if kind == "sorted":                      # empty slots of a sorted list sit at its end
        holes = np.sort(holes, axis=2)
    Ds[holes] = -FLT_MAX
    Is[holes] = -1
    if rng.random() < 0.3:
        Ds[rng.random(Ds.shape) < 0.01] = np.nan
    oD, oI = gpu_merge(Ds, Is, k_out)
    rD, rI = O.merge_total_order(np.where(np.isnan(Ds), -FLT_MAX, Ds), Is, k_out)
    nanfree = np.where(np.isnan(Ds), -FLT_MAX, Ds)
    ok = np.array_equal(oD, rD) and np.array_equal(np.where(oD > -FLT_MAX, oI, -1), np.where(rD > -FLT_MAX, rI, -1))
    if not ok:
        print("MERGE MISMATCH case", c, dict(S=S, nq=nq, k_in=k_in, k_out=k_out, kind=str(kind)), flush=True)
        sys.exit(1)
    # ---- search variants ----
    n = int(rng.integers(1, 5000)); d = int(rng.choice([8, 24, 64, 100])); q = int(rng.integers(1, 80)); k = int(rng.integers(1, 300))
    A = int(rng.integers(1, 5))
    X = rng.standard_normal((n, d), dtype=np.float32)
    E = np.round(rng.standard_normal((q, A, d)) * 16).astype(np.float32) / 16
    w = (rng.integers(1, 5, (q, A)) / 8.0).astype(np.float32)               # exactly representable: the fused vector is exact
    fused = (w[:, :, None] * E).sum(1).astype(np.float32)
    base = int(rng.choice([0, 17, 10 ** 12]))
    storage = str(rng.choice(["f32", "bf16scan"]))
    ix = FlatIPIndex(d, storage=storage)
    ix.add(X); ix.set_id_base(base)
    rD, rI = O.topk_total_order(O.scores_mfma_order(fused, X), k, id_base=base)
    for name, (D, I) in {"aspect": ix.search_aspect(E, w, k), "plain": ix.search(fused, k)}.items():
        if not (np.array_equal(I, rI) and np.array_equal(D, rD)):
            print("SEARCH MISMATCH case", c, name, dict(n=n, d=d, q=q, k=k, A=A, base=base, storage=storage), flush=True)
            sys.exit(1)
    # ---- MURAL pooler entry points: pool == oracle (tolerance), pooled add / search == pool then add / search (bits) ----
    Ls = int(rng.integers(A + 1, A + 12)); begin = int(rng.integers(1, Ls - A + 1))
    W = (rng.standard_normal((A, d)) * 0.5).astype(np.float32); bvec = rng.standard_normal(A).astype(np.float32)   # logits of a few units
    Hp = rng.standard_normal((min(n, 700), Ls, d), dtype=np.float32); Hq = rng.standard_normal((q, Ls, d), dtype=np.float32)
    px = FlatIPIndex(d, storage=storage)
    rep_p, att_p = px.pool(Hp, W, bvec, begin_id=begin)
    orep, oatt = O.mural_pool_att(Hp, W, bvec, begin)
    if not (np.allclose(rep_p, orep, rtol=1e-5, atol=1e-5) and np.allclose(att_p, oatt, rtol=1e-5, atol=1e-6)):
        print("POOL MISMATCH case", c, dict(d=d, A=A, Ls=Ls, begin=begin), "max |rep diff|", float(np.abs(rep_p - orep).max()),
              "max |att diff|", float(np.abs(att_p - oatt).max()), "max |logit|", float(np.abs(Hp[:, 0] @ W.T + bvec).max()), flush=True); sys.exit(1)
    px.add_pooled(Hp, W, bvec, begin_id=begin)
    Dp, Ip = px.search_pooled(Hq, W, bvec, min(k, 200), begin_id=begin)
    py = FlatIPIndex(d, storage=storage); py.add(rep_p)
    rep_q, _ = py.pool(Hq, W, bvec, begin_id=begin)
    Dy, Iy = py.search(rep_q, min(k, 200))
    if not (np.array_equal(Dp, Dy) and np.array_equal(Ip, Iy)):
        print("POOLED SEARCH MISMATCH case", c, dict(d=d, A=A, Ls=Ls, begin=begin, storage=storage), flush=True); sys.exit(1)
    del ix, px, py
    if c % 20 == 19:
        print(f"{c + 1} cases ok, {time.time() - t0:.0f} s", flush=True)
print("all", cases, "cases ok")
