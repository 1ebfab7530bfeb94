"""The CLIs end to end on the GPU, byte-compared with what the reference's own loops would write for the oracle's
result: `python -m tevatron.faiss_retriever` (reference __main__.py:56-92) at BASELINE.json configs[0] size, the
per-shard runs + `python -m tevatron.faiss_retriever.reducer` workflow (reference reducer.py:23-35,
docs/retrieval.md:31-53), and shard pickles large enough to take the mapped-payload load path
(mural_amd/faiss_retriever/__main__.py::_pickle_load_mapped -> ipx_add's pinned staging)."""
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import gaussian
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def drv():
    from mural_amd import _lib
    assert _lib.lib().ipx_device_count() >= 1, "no HIP device visible: the search path has no CPU fallback"
    from mural_amd.faiss_retriever import __main__ as m
    return m


def _dump(path, obj):
    with open(path, "wb") as f:
        pickle.dump(obj, f, protocol=4)


def _expected_ranking(Q, X, k, q_lookup, p_lookup):
    """The bytes the reference's write_ranking loop (__main__.py:32-38) emits for the oracle's top-k: kernel-order
    fp32 scores, order (score desc, row asc), ids through str(p_lookup[row])."""
    D, I = O.topk_total_order(O.scores_mfma_order(Q, X), k)
    lines = []
    for qid, row_scores, row_ids in zip(q_lookup, D, I):
        hits = sorted(zip(row_scores, [str(p_lookup[i]) for i in row_ids]), key=lambda h: h[0], reverse=True)
        lines += [f"{qid}\t{doc}\t{s}\n" for s, doc in hits]
    return "".join(lines).encode()


def test_config1_size_through_the_cli(drv, tmp_path):
    """BASELINE.json configs[0]: 100k x 768 fp32 corpus, 1 000 queries, depth 100, text output — through main(),
    byte for byte what the reference's formatting gives for the oracle's ranking."""
    n, d, nq, k = 100_000, 768, 1000, 100
    X, Q = gaussian(1234, (n, d)), gaussian(4321, (nq, d))
    p_lookup = [f"D{i:06d}" for i in range(n)]
    q_lookup = [f"q{i}" for i in range(nq)]
    _dump(tmp_path / "corpus_emb.pkl", (X, p_lookup))
    _dump(tmp_path / "query_emb.pkl", (Q, q_lookup))
    out = tmp_path / "rank.tsv"
    drv.main(["--query_reps", str(tmp_path / "query_emb.pkl"), "--passage_reps", str(tmp_path / "corpus_emb*.pkl"),
              "--depth", str(k), "--batch_size", "128", "--save_ranking_to", str(out), "--save_text", "--quiet"])
    assert out.read_bytes() == _expected_ranking(Q, X, k, q_lookup, p_lookup)


def test_mapped_shard_pickles_reach_the_index(drv, tmp_path):
    """Shards of >= 1 MiB come back from pickle_load as np.memmap views of the pickle file (no host copy) and are
    streamed to the GPU by ipx_add from the mapping: fp32 and fp16 (widened on the device), 2- and 4-tuples."""
    d, k = 768, 20
    X0 = gaussian(1, (700, d))                                  # 2.1 MiB fp32
    X1 = gaussian(2, (900, d)).astype(np.float16)               # 1.3 MiB fp16
    Q = gaussian(3, (33, d))
    l0, l1 = list(range(1000, 1700)), [f"p{i}" for i in range(900)]
    _dump(tmp_path / "corpus_emb0.pkl", (X0, l0))
    _dump(tmp_path / "corpus_emb1.pkl", (X1, l1, "unused", None))
    _dump(tmp_path / "query_emb.pkl", (Q, list(range(33))))
    for name, X in (("corpus_emb0.pkl", X0), ("corpus_emb1.pkl", X1)):
        reps, lookup = drv.pickle_load(str(tmp_path / name))
        assert isinstance(reps, np.memmap) and reps.dtype == X.dtype and np.array_equal(np.asarray(reps), X)
        # the mapped rows through the retriever: the oracle's bits
        r = drv.BaseFaissIPRetriever(reps)
        r.add(reps)
        D, I = r.search(Q, k)
        rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X.astype(np.float32)), k)
        assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    # ... and through the CLI, both shards in one index (whatever order glob returns them in)
    out = tmp_path / "rank.tsv"
    drv.main(["--query_reps", str(tmp_path / "query_emb.pkl"), "--passage_reps", str(tmp_path / "corpus_emb*.pkl"),
              "--depth", str(k), "--batch_size", "-1", "--save_ranking_to", str(out), "--save_text", "--quiet"])
    import glob
    order = [os.path.basename(f) for f in glob.glob(str(tmp_path / "corpus_emb*.pkl"))]
    parts = {"corpus_emb0.pkl": (X0, l0), "corpus_emb1.pkl": (X1.astype(np.float32), l1)}
    Xall = np.concatenate([parts[f][0] for f in order])
    lall = sum([parts[f][1] for f in order], [])
    assert out.read_bytes() == _expected_ranking(Q, Xall, k, list(range(33)), lall)


def test_per_shard_runs_and_reducer_main_equal_the_single_index_run(drv, tmp_path, monkeypatch, capsys):
    """docs/retrieval.md:31-53: one CLI run per corpus shard into a score directory (pickle output), then
    `reducer --score_dir --query --save_ranking_to`: the bytes of the single-index run over all shards."""
    from mural_amd.faiss_retriever import reducer
    d, k, nq = 96, 50, 40
    rows = [4000, 2500, 3300]
    Xs = [gaussian(10 + s, (n, d)) for s, n in enumerate(rows)]
    base = np.cumsum([0] + rows)
    Q = gaussian(20, (nq, d))
    _dump(tmp_path / "query_emb.pkl", (Q, [f"{7000 + i}" for i in range(nq)]))
    score_dir = tmp_path / "scores"
    score_dir.mkdir()
    for s, X in enumerate(Xs):
        _dump(tmp_path / f"corpus_emb{s}.pkl", (X, list(range(int(base[s]), int(base[s + 1])))))    # global integer ids
        drv.main(["--query_reps", str(tmp_path / "query_emb.pkl"), "--passage_reps", str(tmp_path / f"corpus_emb{s}.pkl"),
                  "--depth", str(k), "--batch_size", "16", "--save_ranking_to", str(score_dir / f"{s:02d}"), "--quiet"])
    merged = tmp_path / "rank_reduced.tsv"
    monkeypatch.setattr(sys, "argv", ["reducer", "--score_dir", str(score_dir), "--query", str(tmp_path / "query_emb.pkl"),
                                      "--save_ranking_to", str(merged)])
    reducer.main()
    assert "Initializing Heap. Assuming 40 queries." in capsys.readouterr().out       # the reference's own message
    single = tmp_path / "rank_single.tsv"
    drv.main(["--query_reps", str(tmp_path / "query_emb.pkl"), "--passage_reps", str(tmp_path / "corpus_emb*.pkl"),
              "--depth", str(k), "--batch_size", "-1", "--save_ranking_to", str(single), "--save_text", "--quiet"])
    assert merged.read_bytes() == single.read_bytes()
    assert merged.read_bytes() == _expected_ranking(Q, np.concatenate(Xs), k, [f"{7000 + i}" for i in range(nq)],
                                                    list(range(int(base[-1]))))

