"""IPX_GPUS=N on a box with several GPUs: one rank process per GPU over RCCL (ShardedFlatIP: packed all-gather + GPU
merge), ranking bytes equal to the single-GPU run. The build pool's boxes have one GPU, so this is skipped there; it is
collected late (file name) so that a failure on a multi-GPU box cannot hide the rest of the suite, and the ranks run
under a time limit (IPX_SPAWN_TIMEOUT) so that a collective that never completes fails instead of hanging."""
import pickle

import numpy as np
import pytest

from conftest import gaussian

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

def test_one_command_on_every_gpu_of_the_box(drv, tmp_path, monkeypatch):
    """IPX_GPUS=N: the reference's command line, unchanged, on N GPUs — one rank process per GPU over RCCL, bytes equal
    to the single-GPU run. Needs at least two visible GPUs (the build pool's boxes have one: skipped there)."""
    from mural_amd import _lib
    n_gpus = min(int(_lib.lib().ipx_device_count()), 8)
    if n_gpus < 2:
        pytest.skip("one GPU visible: the N-rank RCCL path needs at least two")
    d, k, nq = 96, 40, 70
    rows = [3000, 1700, 2600, 900, 1100]
    base = np.cumsum([0] + rows)
    for s, n in enumerate(rows):
        _dump(tmp_path / f"corpus_emb{s}.pkl", (gaussian(30 + s, (n, d)), [f"d{i}" for i in range(int(base[s]), int(base[s + 1]))]))
    _dump(tmp_path / "query_emb.pkl", (gaussian(40, (nq, d)), [f"q{i}" for i in range(nq)]))
    argv = ["--query_reps", str(tmp_path / "query_emb.pkl"), "--passage_reps", str(tmp_path / "corpus_emb*.pkl"),
            "--depth", str(k), "--batch_size", "-1", "--save_text", "--quiet"]
    drv.main(argv + ["--save_ranking_to", str(tmp_path / "rank_one.tsv")])
    monkeypatch.setenv("IPX_GPUS", str(n_gpus))
    monkeypatch.setenv("IPX_SPAWN_TIMEOUT", "300")
    drv.main(argv + ["--save_ranking_to", str(tmp_path / "rank_all.tsv")])
    assert (tmp_path / "rank_all.tsv").read_bytes() == (tmp_path / "rank_one.tsv").read_bytes()
