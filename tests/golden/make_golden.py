"""Generates the golden fixtures in this directory. Run in the BUILD container only:

    cd /root/repo && python tests/golden/make_golden.py

It imports the reference's own Python plumbing (tevatron.faiss_retriever.__main__, .retriever,
.reducer from /root/reference/src) with a stand-in module named `faiss` whose IndexFlatIP /
ResultHeap are the CPU oracle (oracle/oracle.py) — faiss itself is not installed here and
the reference has no fixtures of its own. So these files pin the reference's HOST-side
behaviour (id mapping, dtype of the id array, chunking, pickle formats, negate-merge-negate,
ranking text bytes); the arithmetic inside them is the oracle's restatement (parity unpinned,
see oracle/oracle.py). Only data is written here: inputs and expected outputs.
"""
import io
import os
import pickle
import sys
import types
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

# ---- stand-in for the absent third-party module -------------------------------------------
faiss = types.ModuleType("faiss")
faiss.IndexFlatIP = O.FlatIPOracle
faiss.ResultHeap = O.ResultHeapOracle
sys.modules["faiss"] = faiss
sys.path.insert(0, "/root/reference/src")
for name in [m for m in sys.modules if m == "tevatron" or m.startswith("tevatron.")]:
    del sys.modules[name]
import tevatron.faiss_retriever.__main__ as ref_main  # noqa: E402
import tevatron.faiss_retriever.reducer as ref_reducer  # noqa: E402
from tevatron.faiss_retriever.retriever import BaseFaissIPRetriever as RefRetriever  # noqa: E402

assert ref_main.__file__.startswith("/root/reference/"), ref_main.__file__

out = {}
F32 = np.float32
FLT_MAX = np.finfo(np.float32).max


def ranking_bytes(ids, scores, qlookup):
    path = os.path.join(HERE, "_tmp_rank.txt")
    ref_main.write_ranking(ids, scores, qlookup, path)
    with open(path, "rb") as f:
        b = f.read()
    os.remove(path)
    return np.frombuffer(b, dtype=np.uint8).copy()


# ---- A. write_ranking: float formatting, stable descending sort, id kinds ------------------
special = np.array([
    0.1, 100.0, 1e16, 9.9e15, 1e-4, 9.99e-5, 1e-5, 6.581513404846191, -3.25, 0.0, -0.0,
    1.401298464324817e-45, 1.1754944e-38, 3.4028235e38, -3.4028235e38, np.inf, -np.inf,
    123456.789, 1e15, 16777216.0, 0.5, 1 / 3, 2 / 3, 1e10, 7e-10, 93.61826, 70.215164,
    1e22, 5e-324, 65.28384, 33.333332, 1e-3,
], dtype=np.float64).astype(F32)
rng = np.random.default_rng(2024)
scoresA = np.stack([special, rng.standard_normal(special.size).astype(F32) * 50,
                    np.repeat(np.array([3.0, 2.0, 3.0, 1.0], dtype=F32), 8)])       # row 2: ties
idsA_str = np.array([[f"B{r:02d}{c:03d}" for c in range(special.size)] for r in range(3)])
idsA_int = np.arange(3 * special.size, dtype=np.int64).reshape(3, -1) * 1000003 - 7
qlookA = ["q1", 17, "query-é"]
out["A_scores"] = scoresA
out["A_ids_str"] = idsA_str
out["A_ids_int"] = idsA_int
out["A_qlookup"] = np.array([str(q) for q in qlookA])
out["A_expected_str"] = ranking_bytes(idsA_str, scoresA, qlookA)
out["A_expected_int"] = ranking_bytes(idsA_int, scoresA, qlookA)
# more queries than rows / fewer queries than rows (zip truncation)
out["A_expected_short_q"] = ranking_bytes(idsA_str, scoresA, qlookA[:2])
out["A_expected_long_q"] = ranking_bytes(idsA_str[:2], scoresA[:2], qlookA)
# a larger random table
scoresA2 = (rng.standard_normal((50, 40)) * rng.choice([1e-6, 1e-3, 1, 1e3, 1e8], size=(50, 40))).astype(F32)
idsA2 = rng.integers(0, 10 ** 9, size=(50, 40)).astype(np.int64)
out["A2_scores"] = scoresA2
out["A2_ids"] = idsA2
out["A2_expected"] = ranking_bytes(idsA2.astype(str), scoresA2, list(range(50)))


# ---- B. search_queries: id mapping, '<U' width, -1 quirk, chunk boundaries -----------------
class StubRetriever:
    """Returns fixed results and records how it was called."""

    def __init__(self, scores, ids):
        self.scores, self.ids, self.calls = scores, ids, []

    def search(self, q, k):
        self.calls.append(("search", int(q.shape[0]), int(k)))
        return self.scores[: q.shape[0], :k], self.ids[: q.shape[0], :k]

    def batch_search(self, q, k, bs, quiet=False):
        self.calls.append(("batch_search", int(q.shape[0]), int(k), int(bs), bool(quiet)))
        return self.scores[: q.shape[0], :k], self.ids[: q.shape[0], :k]


idsB = np.array([[0, 5, 11, -1], [3, 3, 2, 1], [11, 10, 9, -1]], dtype=np.int64)
scoresB = np.array([[9, 8, 7, -FLT_MAX], [5, 4, 3, 2], [1, 0.5, 0.25, -FLT_MAX]], dtype=F32)
lookB_str = ["B0048", "A1", "ZZ99999", "x", "docid-with-dash", "7", "a b", "é", "B01", "B02", "B03", "LASTDOC-LONGEST-ID"]
lookB_int = [100, 2, 3000000, 4, 5, 6, 7, 8, 9, 10, 11, 12]
qB = np.zeros((3, 8), dtype=F32)
for tag, look in (("str", lookB_str), ("int", lookB_int)):
    for bs in (128, -100):
        st = StubRetriever(scoresB, idsB)
        args = types.SimpleNamespace(batch_size=bs, depth=4, quiet=True)
        s, p = ref_main.search_queries(st, qB, look, args)
        out[f"B_{tag}_bs{bs}_psg"] = p
        out[f"B_{tag}_bs{bs}_scores"] = s
        out[f"B_{tag}_bs{bs}_calls"] = np.array([repr(c) for c in st.calls])
# without the long last doc being selected: dtype width follows the selected strings only
st = StubRetriever(scoresB[:, :3], idsB[:, :3])
s, p = ref_main.search_queries(st, qB, lookB_str, types.SimpleNamespace(batch_size=0, depth=3, quiet=True))
out["B_str_nolast_psg"] = p
out["B_ids"] = idsB
out["B_scores"] = scoresB
out["B_look_str"] = np.array(lookB_str)
out["B_look_int"] = np.array(lookB_int, dtype=np.int64)


# batch_search chunk boundaries of the reference retriever (nq not a multiple of bs; bs > nq)
class RecordingIndex:
    def __init__(self):
        self.calls = []

    def search(self, q, k):
        self.calls.append(int(q.shape[0]))
        n = q.shape[0]
        base = q[:, 0].astype(np.int64)[:, None]
        return (np.tile(np.arange(k, 0, -1, dtype=F32), (n, 1)) + q[:, :1]), base * 100 + np.arange(k)[None, :]


for nq, bs in ((10, 4), (10, 5), (3, 128), (1, 1)):
    r = RefRetriever(np.zeros((1, 6), dtype=F32))
    r.index = RecordingIndex()
    q = np.arange(nq, dtype=F32)[:, None] * np.ones((1, 6), dtype=F32)
    D, I = r.batch_search(q, 3, bs, quiet=True)
    out[f"B2_nq{nq}_bs{bs}_calls"] = np.array(r.index.calls)
    out[f"B2_nq{nq}_bs{bs}_D"] = D
    out[f"B2_nq{nq}_bs{bs}_I"] = I

# ---- C. pickle_load: 2-tuple, 4-tuple, float16, list of rows -------------------------------
repsC = rng.standard_normal((5, 6)).astype(F32)
lookC = ["d0", "d1", "d2", "d3", "d4"]
cases = {
    "C_tuple2_f32": (repsC, lookC),
    "C_tuple4_f32": (repsC, lookC, "ignored", 123),
    "C_tuple2_f16": (repsC.astype(np.float16), lookC),
    "C_rows_list": ([row for row in repsC], [10, 11, 12, 13, 14]),
}
for name, obj in cases.items():
    path = os.path.join(HERE, name + ".pkl")
    with open(path, "wb") as f:
        pickle.dump(obj, f, protocol=4)
    reps, look = ref_main.pickle_load(path)
    out[name + "_reps"] = reps
    out[name + "_lookup"] = np.array([str(x) for x in look])
    out[name + "_lookup_is_int"] = np.array(isinstance(look[0], int))

# ---- D. reducer: negate-merge-negate over shard results with '<U' int-like ids -------------
nqD, kD, SD = 6, 8, 4
shardsD = []
for s in range(SD):
    n = 8 if s != 2 else 5            # shard 2 holds fewer than k docs -> padded rows
    sc = np.sort(rng.standard_normal((nqD, n)).astype(F32) * 10, axis=1)[:, ::-1]
    ids = (rng.permutation(1000)[: nqD * n].reshape(nqD, n) + 1000 * s)
    D = np.full((nqD, kD), -FLT_MAX, dtype=F32)
    I = np.full((nqD, kD), -1, dtype=np.int64)
    D[:, :n], I[:, :n] = sc, ids
    shardsD.append((D, I.astype(str)))
# exact ties across shards: the same score in shards 0 and 1
shardsD[1][0][0, 0] = shardsD[0][0][0, 0]
buf = io.StringIO()
with redirect_stdout(buf):
    cD, cI = ref_reducer.combine_faiss_results(iter(shardsD))
out["D_shard_scores"] = np.stack([d for d, _ in shardsD])
out["D_shard_ids"] = np.stack([i for _, i in shardsD])
out["D_expected_scores"] = cD
out["D_expected_ids"] = cI
out["D_stdout"] = np.array(buf.getvalue())

# ---- E. end-to-end CLI on a small corpus (reference main() over the oracle index) ----------
nE, dE, nqE, kE = 300, 32, 9, 10
XE = rng.standard_normal((nE, dE)).astype(F32)
QE = rng.standard_normal((nqE, dE)).astype(F32)
lookE = [f"doc{i:04d}" for i in range(nE)]
qlookE = [f"q{i}" for i in range(nqE)]
p0, p1, pq = (os.path.join(HERE, n) for n in ("E_corpus_emb0.pkl", "E_corpus_emb1.pkl", "E_query_emb.pkl"))
with open(p0, "wb") as f:
    pickle.dump((XE[:170], lookE[:170]), f, protocol=4)
with open(p1, "wb") as f:
    pickle.dump((XE[170:].astype(np.float16), lookE[170:]), f, protocol=4)
with open(pq, "wb") as f:
    pickle.dump((QE, qlookE), f, protocol=4)
rank_path = os.path.join(HERE, "_tmp_E_rank.txt")
argv = sys.argv
# two shards; pin the shard order the reference would get from glob by loading them one
# pattern at a time is not possible, so the test reads this order back from the fixture
import glob as _glob  # noqa: E402
orderE = _glob.glob(os.path.join(HERE, "E_corpus_emb*.pkl"))
sys.argv = ["x", "--query_reps", pq, "--passage_reps", os.path.join(HERE, "E_corpus_emb*.pkl"),
            "--depth", str(kE), "--batch_size", "4", "--save_ranking_to", rank_path, "--save_text", "--quiet"]
ref_main.main()
sys.argv = argv
with open(rank_path, "rb") as f:
    out["E_expected_ranking"] = np.frombuffer(f.read(), dtype=np.uint8).copy()
os.remove(rank_path)
out["E_shard_order"] = np.array([os.path.basename(p) for p in orderE])

np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
print("wrote", os.path.join(HERE, "golden.npz"), "with", len(out), "arrays")
