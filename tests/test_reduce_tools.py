"""Shard reducers beyond the integer-id pickle path: the text-file reducer (reference
scripts/reduce_results.py, golden outputs produced by that script itself, see
tests/golden/make_golden_reduce.py) and string doc ids in combine_faiss_results."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR


def run_reduce(case, depth, tmp_path):
    from mural_amd.tools.reduce_results import main
    out = tmp_path / f"{case}.tsv"
    main(["--results_dir", os.path.join(GOLDEN_DIR, f"{case}_in"), "--output", str(out), "--depth", str(depth)])
    with open(out, "rb") as f, open(os.path.join(GOLDEN_DIR, f"{case}_expected.tsv"), "rb") as g:
        return f.read(), g.read()


def test_reduce_results_non_float32_scores_take_the_host_path(tmp_path, capsys):
    got, want = run_reduce("R3", 5, tmp_path)
    assert got == want
    assert "Merging results from 2 result files." in capsys.readouterr().out


def test_reduce_results_cli_flags():
    from mural_amd.tools import reduce_results as rr
    with pytest.raises(SystemExit):
        rr.main(["--output", "x"])                      # --results_dir is required


@pytest.mark.gpu
@pytest.mark.parametrize("case,depth", [("R1", 10), ("R2", 100)])
def test_reduce_results_golden(case, depth, tmp_path):
    got, want = run_reduce(case, depth, tmp_path)
    assert got == want


@pytest.mark.gpu
def test_reduce_results_cross_file_ties_follow_read_order(tmp_path):
    # equal scores in different files: Python's stable sort keeps read order (os.listdir order)
    from mural_amd.tools import reduce_results as rr
    d = tmp_path / "in"
    d.mkdir()
    rng = np.random.default_rng(3)
    for s in range(5):
        with open(d / f"part{s}", "w") as f:
            for q in ("b", "a", "10"):
                for j in range(7):
                    f.write(f"{q}\tD{s}_{j}\t{float(np.float32(rng.integers(0, 4)))}\n")
    out = tmp_path / "o.tsv"
    rr.reduce_results(str(d), str(out), depth=20)
    want = "".join(f"{a}\t{b}\t{c}\n" for a, b, c in rr._reduce_python([str(d / n) for n in os.listdir(d)], 20))
    assert out.read_text() == want


@pytest.mark.gpu
def test_combine_faiss_results_with_string_ids():
    from mural_amd.faiss_retriever.reducer import combine_faiss_results
    rng = np.random.default_rng(9)
    nq, k, S = 13, 20, 5
    shards = []
    for s in range(S):
        D = -np.sort(-rng.standard_normal((nq, k)).astype(np.float32), axis=1)
        I = np.array([[f"B00{s}{q:02d}X{j}" for j in range(k)] for q in range(nq)])
        shards.append((D, I))
    with pytest.raises(ValueError):                 # the reference's behaviour stays the default
        combine_faiss_results(iter(shards))
    oD, oI = combine_faiss_results(iter(shards), string_ids=True)
    assert oI.dtype.kind == "U" and oD.dtype == np.float32
    allD = np.concatenate([d for d, _ in shards], axis=1)
    allI = np.concatenate([i for _, i in shards], axis=1)
    order = np.argsort(-allD, axis=1, kind="stable")[:, :k]
    assert np.array_equal(oD, np.take_along_axis(allD, order, axis=1))
    assert np.array_equal(oI, np.take_along_axis(allI, order, axis=1))
    # integer-parsable strings keep the reference behaviour: int64 ids
    ints = [(d, np.char.add("1", np.arange(k).astype(str))[None, :].repeat(nq, 0)) for d, _ in shards[:2]]
    _, iI = combine_faiss_results(iter(ints), string_ids=True)
    assert iI.dtype == np.int64
