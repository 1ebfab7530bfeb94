"""Randomised end-to-end check of the drop-in CLIs on the GPU box: random shard pickles (f32 / f16 rows, 2- and
4-tuples, str / int ids), random depth and batch size, text and pickle output, then per-shard runs + the reducer
CLI and the text-file reducer; the bytes written must equal what the oracle's scores give through the reference's
formatting (Python f-string of float(np.float32)). Usage: fuzz_cli.py <cases> <seed>."""
import os, sys, time, pickle, shutil, tempfile, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from mural_amd.faiss_retriever import __main__ as drv
from mural_amd.faiss_retriever import reducer as red
from mural_amd.tools import reduce_results as txt

def expected_text(D, I, look, qids):
    out = []
    for qid, ds, ids in zip(qids, D, I):
        hits = sorted(zip(ds, ids), key=lambda h: h[0], reverse=True)      # stable, as __main__.py:34-37
        out += [f"{qid}\t{look[i]}\t{float(s)}\n" for s, i in hits]
    return "".join(out).encode()

cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
t0 = time.time()
for c in range(cases):
    tmp = tempfile.mkdtemp(prefix="ipx_fuzz_cli_")
    try:
        d = int(rng.choice([8, 33, 64, 128]))
        nshards = int(rng.integers(1, 5))
        int_ids = bool(rng.random() < 0.6)
        Xs, looks = [], []
        for s in range(nshards):
            n = int(rng.integers(1, 3000))
            reps = rng.standard_normal((n, d), dtype=np.float32)
            if rng.random() < 0.3:
                reps = reps.astype(np.float16)
            look = [int(10 ** 6 * s + j) for j in range(n)] if int_ids else [f"B{s:02d}{j:06d}" for j in range(n)]
            obj = (reps, look) if rng.random() < 0.7 else (reps, look, None, None)
            with open(os.path.join(tmp, f"corpus_emb{s}.pkl"), "wb") as f:
                pickle.dump(obj, f)
        nq = int(rng.integers(1, 200))
        depth = int(rng.choice([1, 10, 100, 1000]))
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        qids = [str(7000 + j) for j in range(nq)] if rng.random() < 0.5 else [int(7000 + j) for j in range(nq)]
        with open(os.path.join(tmp, "q.pkl"), "wb") as f:
            pickle.dump((Q, qids), f)
        files = glob.glob(os.path.join(tmp, "corpus_emb*.pkl"))            # the order the CLI loads them in
        look_all = []
        for fpath in files:
            r, l = drv.pickle_load(fpath)[:2]
            Xs.append(np.asarray(r, dtype=np.float32)); look_all += l
        X = np.concatenate(Xs)
        rD, rI = O.topk_total_order(O.scores_mfma_order(Q, X), depth)
        # the reference maps a padding id -1 to the last document (python negative index)
        want = expected_text(rD, rI, look_all, qids)
        bs = int(rng.choice([128, 1, 7, -1, 5000]))
        argv = ["--query_reps", os.path.join(tmp, "q.pkl"), "--passage_reps", os.path.join(tmp, "corpus_emb*.pkl"),
                "--depth", str(depth), "--batch_size", str(bs), "--quiet", "--save_ranking_to"]
        drv.main(argv + [os.path.join(tmp, "rank.tsv"), "--save_text"])
        got = open(os.path.join(tmp, "rank.tsv"), "rb").read()
        if got != want:
            print("CLI TEXT MISMATCH case", c, dict(d=d, nshards=nshards, nq=nq, depth=depth, bs=bs, int_ids=int_ids), flush=True); sys.exit(1)
        drv.main(argv + [os.path.join(tmp, "rank.pkl")])
        sc, ids = pickle.load(open(os.path.join(tmp, "rank.pkl"), "rb"))
        if not (np.array_equal(sc, rD) and np.array_equal(ids, drv._lookup_strings(look_all, rI))):
            print("CLI PICKLE MISMATCH case", c, flush=True); sys.exit(1)
        # per-shard runs (the reference's sharded workflow, docs/retrieval.md:31-53) + reducers
        if X.shape[0] >= depth * 0 + 1:
            os.makedirs(os.path.join(tmp, "scores")); os.makedirs(os.path.join(tmp, "texts"))
            for s, fpath in enumerate(files):
                a = ["--query_reps", os.path.join(tmp, "q.pkl"), "--passage_reps", fpath, "--depth", str(depth),
                     "--batch_size", "-1", "--quiet", "--save_ranking_to"]
                drv.main(a + [os.path.join(tmp, "scores", f"s{s}.pkl")])
                drv.main(a + [os.path.join(tmp, "texts", f"s{s}.tsv"), "--save_text"])
            # a shard with fewer rows than depth pads with (-FLT_MAX, last doc): the reducers see those rows too,
            # exactly as the reference's would; compare only when no shard is shorter than depth
            if min(x.shape[0] for x in Xs) >= depth:
                sys.argv = ["reducer", "--score_dir", os.path.join(tmp, "scores"), "--query", os.path.join(tmp, "q.pkl"),
                            "--save_ranking_to", os.path.join(tmp, "reduced.tsv")] + ([] if int_ids else ["--string_ids"])
                red.main()
                got = open(os.path.join(tmp, "reduced.tsv"), "rb").read()
                gl = sorted(got.splitlines()); wl = sorted(want.splitlines())
                if gl != wl:
                    print("REDUCER MISMATCH case", c, dict(d=d, nshards=nshards, nq=nq, depth=depth, int_ids=int_ids), flush=True); sys.exit(1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if c % 10 == 9:
        print(f"{c + 1} cases ok, {time.time() - t0:.0f} s", flush=True)
print("all", cases, "cases ok")
