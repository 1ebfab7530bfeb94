"""Replay of round 2's one unexplained fuzz failure (fuzz_parity.py seed 7, case 114) and of its cause.

  IPX_LIB=libipx_r2.so python scripts/repro_case114.py     # a build of the round-2 sources: shows the mismatch
  python scripts/repro_case114.py                          # the current library: all OK

Part 1 rebuilds the inputs of case 114 (the generator of fuzz_parity.py replayed on the CPU up to that case) and
searches them on storage 1 and 2. Part 2 is the deterministic form of the cause (tests/test_gpu_bf16.py::
test_inf_row_outside_the_side_list_is_not_lost): the side list of outlier rows is full before the row with the inf
element arrives, so that row is scanned like any other; with hi + lo query parts of opposite sign its approximate score
is inf * hi + inf * lo = NaN, which no threshold test admits, while its exact score is +inf."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as O
from mural_amd import _lib
if os.environ.get("IPX_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["IPX_LIB"])
from mural_amd.faiss_retriever.retriever import FlatIPIndex


def bf16_round(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def case114():
    """fuzz_parity.py's generator, replayed without a GPU up to case 114 of seed 7 (same draws in the same order)."""
    rng = np.random.default_rng(7)
    for c in range(115):
        n = int(rng.choice([1, 2, 127, 128, 129, 255, 256, 257, 511, 513, int(rng.integers(1, 3000)), int(rng.integers(3000, 60000))]))
        d = int(rng.choice([1, 7, 8, 16, 17, 31, 32, 33, 48, 64, 100, 128, 192, int(rng.integers(1, 300))]))
        nq = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 192, 193, 256, 257, int(rng.integers(1, 400))]))
        k = int(rng.choice([1, 2, 10, 100, 255, 256, 257, 1000, 1024, 1638, 2048, int(rng.integers(1, 2049))]))
        kind = rng.choice(["gauss", "ties", "quant", "ramp"])
        if kind == "gauss":
            X = rng.standard_normal((n, d), dtype=np.float32)
        elif kind == "ties":
            base = rng.standard_normal((max(1, n // 7), d), dtype=np.float32)
            X = base[rng.integers(0, base.shape[0], n)]
        elif kind == "quant":
            X = rng.integers(-2, 3, (n, d)).astype(np.float32)
        else:
            X = np.zeros((n, d), np.float32); X[:, 0] = np.arange(n, dtype=np.float32) * (1 if rng.random() < 0.5 else -1)
        storage = str(rng.choice(["f32", "bf16scan", "bf16"]))
        if rng.random() < 0.3:
            m = int(rng.choice([1, 3, 40, 300]))
            rows = rng.integers(0, n, m)
            with np.errstate(all="ignore"):
                X[rows] *= rng.choice([3.0, 10.0, 1e4, 1e30], size=(m, 1)).astype(np.float32)
            if storage != "bf16":
                for r, v in zip(rng.integers(0, n, 3), (np.nan, np.inf, -np.inf)):
                    if rng.random() < 0.5:
                        X[r, int(rng.integers(0, d))] = v
        Q = rng.integers(-2, 3, (nq, d)).astype(np.float32) if kind == "quant" else rng.standard_normal((nq, d), dtype=np.float32)
        ms = int(rng.choice([0, 0, 1, 2, 5, 40]))
        adds = int(rng.choice([1, 1, 2, 5]))
        step = -(-n // adds)
        if rng.random() < 0.3:
            rng.integers(0, 2 * n + 2)
        X = X.copy()
        hows = []
        for lo in range(0, n, step):
            how = str(rng.choice(["f32", "f16", "torch_f32", "torch_bf16"]))
            hows.append(how)
            part = X[lo:lo + step]
            if how == "f16":
                with np.errstate(all="ignore"):
                    X[lo:lo + step] = np.clip(part, -60000, 60000).astype(np.float16).astype(np.float32)
            elif how == "torch_bf16":
                X[lo:lo + step] = torch.from_numpy(part).to(torch.bfloat16).float().numpy()
        if storage != "f32":
            for draw in ([1, 2], [1, 2], [0, 1], None, [0, 1, 2, 3, 7], [2, 3, 4], [1, 2], [1, 2], [0, 1], [3, 4], [0, 2, 5]):
                rng.integers(0, 8) if draw is None else rng.choice(draw)
    return dict(n=n, d=d, nq=nq, k=k, kind=str(kind), storage=storage, ms=ms, adds=adds, hows=hows, step=step), X, Q


def report(tag, ix, Q, ref, k):
    with np.errstate(all="ignore"):
        rD, rI = O.topk_total_order(O.scores_mfma_order(Q, ref), k)
    D, I = ix.search(Q, k)
    bad = sorted(set(np.argwhere(I != rI)[:, 0].tolist()) | set(np.argwhere(D.view(np.uint32) != rD.view(np.uint32))[:, 0].tolist()))
    print(tag, "k", k, "OK" if not bad else f"MISMATCH in {len(bad)} queries {bad[:12]}", ix.last_bf16_stats(), flush=True)
    return not bad


ok = True
meta, X, Q = case114()
print("case 114:", meta, "| non-finite elements at", np.argwhere(~np.isfinite(X)).tolist(), flush=True)
assert (meta["n"], meta["d"], meta["nq"], meta["k"], meta["kind"], meta["storage"]) == (2122, 192, 65, 2048, "ramp", "bf16scan")
hi = bf16_round(Q[:, 70]); lo = bf16_round(Q[:, 70] - hi)
print("queries whose hi / lo parts of element 70 (the inf column of row 839) have opposite signs, q > 0:",
      np.flatnonzero((Q[:, 70] > 0) & (lo < 0))[:12].tolist(), flush=True)
for storage, ref in (("bf16scan", X), ("bf16", bf16_round(X))):
    ix = FlatIPIndex(meta["d"], storage=storage)
    for lo_ in range(0, meta["n"], meta["step"]):
        ix.add(X[lo_:lo_ + meta["step"]])
    ix.set_param("max_splits", meta["ms"])
    for parts in (2, 1):
        ix.set_param("query_parts", parts)
        ix.set_param("bf16_candidates", 4096)
        ok &= report(f"case 114 data, {storage}, parts {parts}, 4096 candidates:", ix, Q, ref, 1024)
    del ix

n, d, nq = 2122, 192, 65
X = np.zeros((n, d), np.float32); X[:, 0] = np.arange(n, dtype=np.float32)
X[1500, 70] = np.inf; X[1600, 62] = np.nan; X[1700] *= np.float32(1e30)
Q = np.random.default_rng(4321).standard_normal((nq, d), dtype=np.float32)
for storage, ref in (("bf16scan", X), ("bf16", bf16_round(X))):
    ix = FlatIPIndex(d, storage=storage)
    ix.add(X[:425]); ix.add(X[425:1025]); ix.add(X[1025:])
    for parts in (2, 1):
        for k in (100, 1024):
            ix.set_param("query_parts", parts)
            ix.set_param("bf16_candidates", 4096)
            ok &= report(f"side list full before the inf row, {storage}, parts {parts}:", ix, Q, ref, k)
    del ix
print("ALL OK" if ok else "MISMATCHES (see above)")
