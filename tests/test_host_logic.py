"""Host-side mirror of the reference interface, checked on CPU against the golden fixtures
that tests/golden/make_golden.py captured from the reference's own Python plumbing
(__main__.py / retriever.py / reducer.py of /root/reference/src/tevatron/faiss_retriever).
No GPU compute here: the retriever is replaced by stubs, the C ranking writer is host code."""
import ctypes as C
import os
import pickle
import struct
import types

import numpy as np
import pytest

from mural_amd import _lib
from mural_amd.faiss_retriever import __main__ as drv
from mural_amd.faiss_retriever.retriever import BaseFaissIPRetriever, FaissRetriever

from conftest import GOLDEN_DIR


# ---- C ABI surface ---------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(ipx):
    header = open(os.path.join(os.path.dirname(GOLDEN_DIR), "..", "include", "ipx.h")).read()
    import re
    declared = set(re.findall(r"\b(ipx_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(ipx, name), name
    assert ipx.ipx_version().decode().startswith("ipx")
    assert ipx.ipx_device_count() >= 0


def test_argument_errors_need_no_gpu(ipx):
    h = C.c_void_p()
    assert ipx.ipx_create(0, 0, C.byref(h)) == 1           # IPX_ERR_ARG
    assert b"dimension" in ipx.ipx_last_error()
    assert ipx.ipx_search(None, None, 1, 10, None, None, 0, None) == 1
    assert ipx.ipx_merge_topk(None, None, 2, 4, 0, 5, None, None, 0, 0, None) == 1
    assert ipx.ipx_ntotal(None) == 0 and ipx.ipx_dim(None) == 0


# ---- ranking text ----------------------------------------------------------------------------
def _fmt(ipx, x):
    buf = C.create_string_buffer(64)
    n = ipx.ipx_format_score(C.c_float(x), buf, 64)
    return buf.raw[:n].decode()


def test_format_score_equals_python_repr(ipx):
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32)
    vals = bits.view(np.float32)
    extra = np.array([0.1, 100.0, 1e16, 9.99e15, 1e-4, 9.9e-5, 1e-5, 0.0, -0.0, np.inf, -np.inf, np.nan,
                      1.4e-45, 3.4028235e38, 16777216.0, 1e15, 123456.789], dtype=np.float32)
    for v in np.concatenate([vals, extra]):
        assert _fmt(ipx, float(v)) == f"{v}", (v, _fmt(ipx, float(v)), f"{v}")


def _ranking(ids, scores, qlookup, tmp_path):
    p = tmp_path / "rank.txt"
    drv.write_ranking(ids, scores, qlookup, str(p))
    return np.frombuffer(p.read_bytes(), dtype=np.uint8)


def test_write_ranking_golden(golden, tmp_path):
    ql = ["q1", 17, "query-é"]
    assert [str(q) for q in ql] == list(golden["A_qlookup"])
    s, ids_s, ids_i = golden["A_scores"], golden["A_ids_str"], golden["A_ids_int"]
    assert np.array_equal(_ranking(ids_s, s, ql, tmp_path), golden["A_expected_str"])
    assert np.array_equal(_ranking(ids_i, s, ql, tmp_path), golden["A_expected_int"])
    assert np.array_equal(_ranking(ids_s, s, ql[:2], tmp_path), golden["A_expected_short_q"])
    assert np.array_equal(_ranking(ids_s[:2], s[:2], ql, tmp_path), golden["A_expected_long_q"])
    assert np.array_equal(_ranking(golden["A2_ids"].astype(str), golden["A2_scores"], list(range(50)), tmp_path),
                          golden["A2_expected"])
    # integer ids straight from the reducer take the int64 path of the C writer
    assert np.array_equal(_ranking(golden["A2_ids"], golden["A2_scores"], list(range(50)), tmp_path),
                          golden["A2_expected"])


def _ranking_from_rows(lookup, rows, scores, qlookup, tmp_path):
    p = tmp_path / "rank_rows.txt"
    assert drv._write_ranking_from_rows(lookup, rows, scores, qlookup, str(p))
    return np.frombuffer(p.read_bytes(), dtype=np.uint8)


def test_write_ranking_from_rows_golden(golden, tmp_path):
    """The CLI's text output goes from row ids to the file in one C pass (ipx_write_ranking_lookup): the golden
    bytes of the reference's lookup + write_ranking loops (__main__.py:27-38) for the same hits."""
    ql = ["q1", 17, "query-é"]
    s = golden["A_scores"]
    for ids, exp in ((golden["A_ids_str"], "A_expected_str"), (golden["A_ids_int"], "A_expected_int")):
        uniq, rows = np.unique(ids, return_inverse=True)
        lookup = uniq.tolist()                              # str entries / int entries (str() applied by the writer)
        rows = rows.reshape(ids.shape).astype(np.int64)
        assert np.array_equal(_ranking_from_rows(lookup, rows, s, ql, tmp_path), golden[exp])
        # negative rows index from the end, as list indexing does in the reference
        assert np.array_equal(_ranking_from_rows(lookup, rows - len(lookup), s, ql, tmp_path), golden[exp])
    uniq, rows = np.unique(golden["A_ids_str"], return_inverse=True)
    rows = rows.reshape(s.shape).astype(np.int64)
    assert np.array_equal(_ranking_from_rows(uniq.tolist(), rows, s, ql[:2], tmp_path), golden["A_expected_short_q"])
    assert np.array_equal(_ranking_from_rows(uniq.tolist(), rows[:2], s[:2], ql, tmp_path), golden["A_expected_long_q"])
    uniq, rows = np.unique(golden["A2_ids"], return_inverse=True)
    rows = rows.reshape(golden["A2_ids"].shape).astype(np.int64)
    assert np.array_equal(_ranking_from_rows(uniq.tolist(), rows, golden["A2_scores"], list(range(50)), tmp_path),
                          golden["A2_expected"])


def test_write_ranking_from_rows_edges(tmp_path):
    rows = np.array([[2, 0, 1]], dtype=np.int64)
    s = np.array([[0.5, 2.0, 2.0]], dtype=np.float32)
    p = tmp_path / "r.txt"
    # multi-byte ids, an empty id, a 1-entry lookup
    assert drv._write_ranking_from_rows(["é€", "", "日本"], rows, s, ["q"], str(p))
    assert p.read_text(encoding="utf-8") == "q\té€\t2.0\nq\t\t2.0\nq\t日本\t0.5\n"
    assert drv._write_ranking_from_rows(["only"], np.zeros((1, 1), np.int64), s[:, :1], ["q"], str(p))
    assert p.read_text() == "q\tonly\t0.5\n"
    # out-of-range rows raise what list indexing raises in the reference; ids with a newline go the long way
    with pytest.raises(IndexError):
        drv._write_ranking_from_rows(["a", "b"], rows, s, ["q"], str(p))
    assert drv._write_ranking_from_rows(["a\nb", "c", "d"], rows, s, ["q"], str(p)) is False
    assert drv._write_ranking_from_rows(["a", "b", "c"], rows.astype(np.int32), s, ["q"], str(p)) is False
    # the C entry point checks the rows itself
    ipx = _lib.lib()
    blob, off = drv._lookup_blob(["a", "b"])
    qids = np.array(["q"])
    rc = ipx.ipx_write_ranking_lookup(str(p).encode(), C.c_void_p(qids.ctypes.data), 1, blob, C.c_void_p(off.ctypes.data), 1, 2,
                                      C.c_void_p(rows.ctypes.data), C.c_void_p(s.ctypes.data), 1, 3)
    assert rc == 1 and b"out of range" in ipx.ipx_last_error()


def test_write_ranking_generic_types_fall_back(tmp_path):
    # float64 scores / list inputs: the Python path, same format as the reference loop
    p = tmp_path / "r.txt"
    drv.write_ranking([["a", "b"]], [[0.5, 2.0]], ["q"], str(p))
    assert p.read_text() == "q\tb\t2.0\nq\ta\t0.5\n"
    drv.write_ranking(np.array([["a", "b"]]), np.array([[0.5, 2.0]]), ["q"], str(p))
    assert p.read_text() == "q\tb\t2.0\nq\ta\t0.5\n"


def test_write_ranking_unwritable_path(tmp_path):
    with pytest.raises(_lib.IpxError):
        drv.write_ranking(np.array([["a"]]), np.ones((1, 1), np.float32), ["q"], str(tmp_path / "no" / "dir.txt"))


# ---- search_queries / batch_search -----------------------------------------------------------
class StubRetriever:
    def __init__(self, scores, ids):
        self.scores, self.ids, self.calls = scores, ids, []

    def search(self, q, k):
        self.calls.append(("search", int(q.shape[0]), int(k)))
        return self.scores[: q.shape[0], :k], self.ids[: q.shape[0], :k]

    def batch_search(self, q, k, bs, quiet=False):
        self.calls.append(("batch_search", int(q.shape[0]), int(k), int(bs), bool(quiet)))
        return self.scores[: q.shape[0], :k], self.ids[: q.shape[0], :k]


@pytest.mark.parametrize("tag", ["str", "int"])
@pytest.mark.parametrize("bs", [128, -100])
def test_search_queries_golden(golden, tag, bs):
    look = list(golden["B_look_str"]) if tag == "str" else [int(x) for x in golden["B_look_int"]]
    st = StubRetriever(golden["B_scores"], golden["B_ids"])
    args = types.SimpleNamespace(batch_size=bs, depth=4, quiet=True)
    s, p = drv.search_queries(st, np.zeros((3, 8), np.float32), look, args)
    exp = golden[f"B_{tag}_bs{bs}_psg"]
    assert p.dtype == exp.dtype and np.array_equal(p, exp)      # incl. -1 -> last document
    assert np.array_equal(s, golden[f"B_{tag}_bs{bs}_scores"])
    assert [repr(c) for c in st.calls] == list(golden[f"B_{tag}_bs{bs}_calls"])


def test_search_queries_dtype_follows_selected_ids(golden):
    st = StubRetriever(golden["B_scores"][:, :3], golden["B_ids"][:, :3])
    s, p = drv.search_queries(st, np.zeros((3, 8), np.float32), list(golden["B_look_str"]),
                              types.SimpleNamespace(batch_size=0, depth=3, quiet=True))
    assert p.dtype == golden["B_str_nolast_psg"].dtype and np.array_equal(p, golden["B_str_nolast_psg"])


def test_search_queries_both_lookup_strategies_agree():
    rng = np.random.default_rng(0)
    look = [f"d{i}" for i in range(50)]
    for shape in [(3, 4), (20, 10)]:          # fewer / more hits than documents
        ids = rng.integers(-1, 50, size=shape)
        st = StubRetriever(np.zeros(shape, np.float32), ids)
        _, p = drv.search_queries(st, np.zeros((shape[0], 2), np.float32), look,
                                  types.SimpleNamespace(batch_size=0, depth=shape[1], quiet=True))
        exp = np.array([[str(look[x]) for x in row] for row in ids])
        assert p.dtype == exp.dtype and np.array_equal(p, exp)
    with pytest.raises(IndexError):
        drv.search_queries(StubRetriever(np.zeros((1, 1), np.float32), np.array([[50]])),
                           np.zeros((1, 2), np.float32), look, types.SimpleNamespace(batch_size=0, depth=1, quiet=True))


class RecordingIndex:
    def __init__(self):
        self.calls = []

    def search(self, q, k):
        self.calls.append(int(q.shape[0]))
        n = q.shape[0]
        base = q[:, 0].astype(np.int64)[:, None]
        return (np.tile(np.arange(k, 0, -1, dtype=np.float32), (n, 1)) + q[:, :1]), base * 100 + np.arange(k)[None, :]


@pytest.mark.parametrize("nq,bs", [(10, 4), (10, 5), (3, 128), (1, 1)])
def test_batch_search_chunking_golden(golden, nq, bs):
    r = BaseFaissIPRetriever.__new__(BaseFaissIPRetriever)      # no GPU: skip the index constructor
    r.index = RecordingIndex()
    q = np.arange(nq, dtype=np.float32)[:, None] * np.ones((1, 6), dtype=np.float32)
    D, I = r.batch_search(q, 3, bs, quiet=True)
    assert r.index.calls == list(golden[f"B2_nq{nq}_bs{bs}_calls"])
    assert np.array_equal(D, golden[f"B2_nq{nq}_bs{bs}_D"]) and np.array_equal(I, golden[f"B2_nq{nq}_bs{bs}_I"])


def test_faiss_retriever_name_is_importable():
    with pytest.raises(NotImplementedError):
        FaissRetriever(np.zeros((1, 8), np.float32), "IVF100,Flat")


# ---- pickles ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["C_tuple2_f32", "C_tuple4_f32", "C_tuple2_f16", "C_rows_list"])
def test_pickle_load_golden(golden, name):
    reps, look = drv.pickle_load(os.path.join(GOLDEN_DIR, name + ".pkl"))
    exp = golden[name + "_reps"]
    assert isinstance(reps, np.ndarray) and reps.dtype == exp.dtype and np.array_equal(reps, exp)
    assert [str(x) for x in look] == list(golden[name + "_lookup"])
    assert isinstance(look[0], int) == bool(golden[name + "_lookup_is_int"])


def test_pickle_load_bad_tuple_and_save_roundtrip(tmp_path):
    p = tmp_path / "bad.pkl"
    with open(p, "wb") as f:
        pickle.dump((1, 2, 3), f)
    with pytest.raises(UnboundLocalError):
        drv.pickle_load(str(p))
    obj = (np.ones((2, 3), np.float32), np.array([["a", "b", "c"], ["d", "e", "f"]]))
    drv.pickle_save(obj, str(tmp_path / "o.pkl"))
    with open(tmp_path / "o.pkl", "rb") as f:
        back = pickle.load(f)
    assert np.array_equal(back[0], obj[0]) and np.array_equal(back[1], obj[1])


def test_cli_flags_and_defaults():
    a = drv._parse_args(["--query_reps", "q", "--passage_reps", "p*", "--save_ranking_to", "o"])
    assert (a.batch_size, a.depth, a.save_text, a.quiet) == (128, 1000, False, False)
    a = drv._parse_args(["--query_reps", "q", "--passage_reps", "p*", "--save_ranking_to", "o",
                         "--batch_size", "-100", "--depth", "10", "--save_text", "--quiet"])
    assert (a.batch_size, a.depth, a.save_text, a.quiet) == (-100, 10, True, True)
    with pytest.raises(SystemExit):
        drv._parse_args(["--query_reps", "q"])


def test_empty_glob_is_index_error():
    with pytest.raises(IndexError):
        drv._load_index("/nonexistent/dir/*.pkl")


def test_tevatron_module_path_is_the_drop_in():
    import tevatron.faiss_retriever as t
    import tevatron.faiss_retriever.__main__ as tm
    import tevatron.faiss_retriever.reducer as tr
    assert t.BaseFaissIPRetriever is BaseFaissIPRetriever
    assert tm.search_queries is drv.search_queries and tm.write_ranking is drv.write_ranking
    assert tm.pickle_load is drv.pickle_load and callable(tr.combine_faiss_results) and callable(tr.main)


def test_product_code_never_touches_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for base in ("mural_amd", "tevatron", "include"):
        for dp, _, files in os.walk(os.path.join(root, base)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, fn), errors="replace").read()
                    assert "oracle" not in txt.replace("oracle-engine", ""), os.path.join(dp, fn)


# ---- index load path: mapped pickle payload ---------------------------------
def test_pickle_load_maps_large_payloads_and_equals_plain_unpickling(tmp_path, monkeypatch):
    import pickle
    from mural_amd.faiss_retriever import __main__ as drv
    rng = np.random.default_rng(5)
    big32 = rng.standard_normal((3000, 128), dtype=np.float32)           # 1.5 MB: above the 1 MiB threshold
    cases = {
        "f32": ((big32, [str(i) for i in range(3000)]), True),
        "f16_4tuple": ((big32.astype(np.float16).repeat(2, axis=0), list(range(6000)), None, {"x": 1}), True),
        "small": ((big32[:10], list(range(10))), False),
        "list_of_rows": (([r for r in big32[:7]], list(range(7))), False),
        "fortran": ((np.asfortranarray(big32), list(range(3000))), False),
        "two_payloads": ((big32, big32 + 1), False),
    }
    for name, (obj, mapped) in cases.items():
        for proto in (2, 4, 5):
            path = tmp_path / f"{name}_{proto}.pkl"
            with open(path, "wb") as f:
                pickle.dump(obj, f, protocol=proto)
            monkeypatch.setenv("IPX_MMAP_LOAD", "0")
            a = drv.pickle_load(str(path))
            monkeypatch.setenv("IPX_MMAP_LOAD", "1")
            b = drv.pickle_load(str(path))
            assert isinstance(b[0], np.ndarray) and b[0].dtype == a[0].dtype and b[0].shape == a[0].shape
            assert np.array_equal(a[0], b[0]), (name, proto)
            if isinstance(a[1], np.ndarray):
                assert np.array_equal(a[1], b[1])
            else:
                assert a[1] == b[1]
            assert isinstance(b[0], np.memmap) == (mapped and proto >= 3), (name, proto)
    with open(tmp_path / "bad.pkl", "wb") as f:
        pickle.dump((big32, [], 3), f)
    with pytest.raises(UnboundLocalError):
        drv.pickle_load(str(tmp_path / "bad.pkl"))
