"""Text-file shard reducer on the MI355X merge kernel.

Mirrors /root/reference/scripts/reduce_results.py:1-28 — same flags (--results_dir, --output,
--depth), same output bytes: ranking files `qid docid score` (any whitespace) from several shards
are merged per query, hits sorted by score descending with Python's stable sort (equal scores stay
in the order they were read: files in os.listdir order, lines top to bottom), cut at --depth,
queries in first-seen order, lines f'{qid}\\t{docid}\\t{score}\\n' with score = repr(float).

The merge runs in ipx_merge_topk (include/ipx.h): every hit gets its per-query sequence number as
id, so the kernel's order (score desc, id asc) IS the stable sort. Scores written by the retrieval
CLI are float32 values printed exactly, so they go through the kernel without loss; a file with
a score that is not a float32 takes the pure-Python path below (same output, no GPU).
"""
import argparse
import ctypes as C
import os

import numpy as np

from .. import _lib

_FLT_MAX = np.finfo(np.float32).max


def _read(path):
    """One ranking file -> (qids, docids, scores float64), split on any whitespace like str.split()."""
    import pandas as pd
    if os.path.getsize(path) == 0:
        return np.array([], dtype=object), np.array([], dtype=object), np.array([], dtype=np.float64)
    df = pd.read_csv(path, sep=r"\s+", header=None, names=["q", "d", "s"], dtype={"q": str, "d": str, "s": np.float64},
                     engine="c", float_precision="round_trip", keep_default_na=False, na_filter=False, quoting=3)
    return df["q"].to_numpy(), df["d"].to_numpy(), df["s"].to_numpy()


def _reduce_python(files, depth):
    """The reference algorithm as is (used when a score is not exactly a float32)."""
    all_results = {}
    for path in files:
        with open(path) as f:
            for line in f:
                qid, docid, score = line.split()
                all_results.setdefault(qid, []).append((docid, float(score)))
    for qid, hits in all_results.items():
        for docid, score in sorted(hits, key=lambda x: x[1], reverse=True)[:depth]:
            yield qid, docid, score


def reduce_results(results_dir, output, depth=100, device=None):
    names = os.listdir(results_dir)
    print(f'Merging results from {len(names)} result files.')
    files = [os.path.join(results_dir, n) for n in names]
    parts = [_read(p) for p in files]
    q = np.concatenate([p[0] for p in parts]) if parts else np.array([], dtype=object)
    dids = np.concatenate([p[1] for p in parts]) if parts else np.array([], dtype=object)
    sc = np.concatenate([p[2] for p in parts]) if parts else np.array([], dtype=np.float64)
    print(f'Writing output to {output} with depth {depth}')
    s32 = sc.astype(np.float32)
    exact = bool(np.all((s32.astype(np.float64) == sc) | np.isnan(sc))) and not np.isnan(sc).any() \
        and bool(np.all(np.abs(s32) < _FLT_MAX)) and 0 < depth <= _lib.IPX_MAX_DEPTH \
        and not bool(np.any((sc == 0) & np.signbit(sc)))      # -0.0 == 0.0 for sorted(), not for the kernel's keys
    if q.size == 0 or not exact:
        with open(output, 'w') as f:
            f.writelines(f'{a}\t{b}\t{c}\n' for a, b, c in _reduce_python(files, depth))
        return
    # queries in first-seen order; per-query sequence number of every hit (insertion order)
    uq, first, inv = np.unique(q, return_index=True, return_inverse=True)
    order_of_q = np.argsort(first, kind="stable")            # rank -> unique index, by first appearance
    rank_of_u = np.empty_like(order_of_q)
    rank_of_u[order_of_q] = np.arange(len(uq))
    qrank = rank_of_u[inv]                                   # per line: its query's output position
    by_q = np.argsort(qrank, kind="stable")                  # lines grouped by query, insertion order kept
    counts = np.bincount(qrank, minlength=len(uq))
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    seq = np.arange(q.size) - np.repeat(starts, counts)      # within-query sequence number (grouped order)
    nq, width = len(uq), int(counts.max())
    # lists of at most IPX_MAX_K entries: [S][nq][kin]
    kin = min(width, _lib.IPX_MAX_K)
    S = -(-width // kin)
    D = np.full((S, nq, kin), -_FLT_MAX, dtype=np.float32)
    I = np.full((S, nq, kin), -1, dtype=np.int64)
    gq = qrank[by_q]
    D[seq // kin, gq, seq % kin] = s32[by_q]
    I[seq // kin, gq, seq % kin] = seq
    kout = min(depth, width)
    oD = np.empty((nq, kout), dtype=np.float32)
    oI = np.empty((nq, kout), dtype=np.int64)
    if device is None:
        device = int(os.environ.get("IPX_DEVICE", "0"))
    _lib.check(_lib.lib().ipx_merge_topk(C.c_void_p(D.ctypes.data), C.c_void_p(I.ctypes.data), S, nq, kin, kout,
                                         C.c_void_p(oD.ctypes.data), C.c_void_p(oI.ctypes.data), 0, device, None))
    # sequence numbers back to doc ids; ragged rows (fewer hits than depth) end at the first -1
    grouped_d = dids[by_q]
    with open(output, 'w') as f:
        for r in range(nq):
            n = int(min(counts[r], kout))
            qid = uq[order_of_q[r]]
            ids = grouped_d[starts[r] + oI[r, :n]]
            f.writelines(f'{qid}\t{d_}\t{float(s_)}\n' for d_, s_ in zip(ids, oD[r, :n]))


def main(argv=None):
    parser = argparse.ArgumentParser(description='Reduce retrieval results from multiple shards.')
    parser.add_argument('--results_dir', type=str, help='Directory that contains results from all shards', required=True)
    parser.add_argument('--output', help='Path to final results file', required=True)
    parser.add_argument('--depth', type=int, help='Number of retrieved doc for each query', required=False, default=100)
    args = parser.parse_args(argv)
    reduce_results(args.results_dir, args.output, args.depth)


if __name__ == '__main__':
    main()
