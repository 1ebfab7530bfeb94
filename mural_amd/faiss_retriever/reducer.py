"""Shard-result reducer: `python -m tevatron.faiss_retriever.reducer` on the MI355X engine.

Mirrors /root/reference/src/tevatron/faiss_retriever/reducer.py:10-39. The reference
streams every shard's (scores, ids) through faiss.ResultHeap on negated scores; here the
same k-way merge runs on the GPU (ipx_merge_topk, include/ipx.h), a few shards per call so
the shard pickles are still consumed lazily.
"""
import ctypes as C
import glob
import os
from argparse import ArgumentParser
from typing import Iterable, Tuple

import numpy as np
from numpy import ndarray
from tqdm import tqdm

from .. import _lib
from .__main__ import pickle_load, write_ranking

_FLT_MAX = np.finfo(np.float32).max
_SHARDS_PER_CALL = 8


def _merge(blocks_D, blocks_I, k_out, device):
    D = np.ascontiguousarray(np.stack(blocks_D))
    I = np.ascontiguousarray(np.stack(blocks_I))
    S, nq, k_in = D.shape
    out_D = np.empty((nq, k_out), dtype=np.float32)
    out_I = np.empty((nq, k_out), dtype=np.int64)
    _lib.check(_lib.lib().ipx_merge_topk(C.c_void_p(D.ctypes.data), C.c_void_p(I.ctypes.data), S, nq, k_in, k_out,
                                         C.c_void_p(out_D.ctypes.data), C.c_void_p(out_I.ctypes.data), 0, device, None))
    return out_D, out_I


def _pad_to(D, I, k):
    if D.shape[1] == k:
        return D, I
    pd = np.full((D.shape[0], k), -_FLT_MAX, dtype=np.float32)
    pi = np.full((D.shape[0], k), -1, dtype=np.int64)
    pd[:, :D.shape[1]] = D
    pi[:, :I.shape[1]] = I
    return pd, pi


def combine_faiss_results(results: Iterable[Tuple[ndarray, ndarray]], string_ids: bool = False):
    """reducer.py:10-20: global top-k over the shards' (scores, ids), k and the query count
    taken from the first shard. Returns (scores f32 [nq,k] descending, ids [nq,k]); unfilled
    slots are (-FLT_MAX, -1).

    Ids are coerced to int64 as in the reference (docs/retrieval.md:54: "reducer requires doc/query
    id being integer"; anything else is a ValueError). Beyond the reference, opt-in with
    string_ids=True (CLI: --string_ids, or IPX_REDUCER_STRING_IDS=1): id arrays that are not
    integer-parsable (MA-Amazon product ids are ASIN strings, data.py:1362) are accepted — every
    distinct string gets an integer code for the merge and the result carries the strings back
    as a '<U' array (ties between equal scores break in code order: first shard first)."""
    string_ids = string_ids or os.environ.get("IPX_REDUCER_STRING_IDS", "0") not in ("", "0")
    device = int(os.environ.get("IPX_DEVICE", "0"))
    nq = k = None
    run_D = run_I = None
    pend_D, pend_I = [], []

    def flush():
        nonlocal run_D, run_I, pend_D, pend_I
        if not pend_D:
            return
        kin = max([k] + [d.shape[1] for d in pend_D])
        bd, bi = [], []
        if run_D is not None:
            pd, pi = _pad_to(run_D, run_I, kin)
            bd.append(pd), bi.append(pi)
        for d, i in zip(pend_D, pend_I):
            pd, pi = _pad_to(d, i, kin)
            bd.append(pd), bi.append(pi)
        run_D, run_I = _merge(bd, bi, k, device)
        pend_D, pend_I = [], []

    codes = None            # str -> code, once a shard with non-integer ids has been seen
    names = []

    def intern(ids):
        nonlocal codes
        if codes is None:
            codes = {}
        uniq, inv = np.unique(np.asarray(ids).astype(str), return_inverse=True)
        table = np.empty(len(uniq), dtype=np.int64)
        for j, name in enumerate(uniq.tolist()):
            c = codes.get(name)
            if c is None:
                c = codes[name] = len(names)
                names.append(name)
            table[j] = c
        return table[inv].reshape(np.shape(ids))

    for scores, indices in results:
        if nq is None:
            print(f'Initializing Heap. Assuming {scores.shape[0]} queries.')
            nq, k = scores.shape[0], scores.shape[1]
        n, kd = scores.shape
        scores = np.ascontiguousarray(scores, dtype='float32')
        if codes is None:
            try:
                indices = np.ascontiguousarray(indices, dtype='int64')
            except ValueError:
                if not string_ids or run_D is not None or pend_D:
                    raise           # integer shards were merged already: mixing the two id kinds is an error
                indices = intern(indices)
        else:
            indices = intern(indices)
        assert indices.shape == (n, kd)
        assert n == nq
        pend_D.append(scores), pend_I.append(indices)
        if len(pend_D) >= _SHARDS_PER_CALL:
            flush()
    flush()
    # the reference fails on an empty iterable (rh is None -> AttributeError on finalize)
    if run_D is None:
        raise AttributeError("'NoneType' object has no attribute 'finalize'")
    if codes is not None:
        table = np.array(names + ['-1'])        # code -1 (an unfilled slot) -> '-1'
        run_I = table[run_I]
    return run_D, run_I


def main():
    parser = ArgumentParser()
    for flag in ('--score_dir', '--query', '--save_ranking_to'):
        parser.add_argument(flag, required=True)
    parser.add_argument('--string_ids', action='store_true',
                        help='not in the reference: accept doc ids that are not integers')
    args = parser.parse_args()

    partitions = glob.glob(f'{args.score_dir}/*')
    corpus_scores, corpus_indices = combine_faiss_results(map(pickle_load, tqdm(partitions)), string_ids=args.string_ids)

    _, q_lookup = pickle_load(args.query)
    write_ranking(corpus_indices, corpus_scores, q_lookup, args.save_ranking_to)


if __name__ == '__main__':
    main()
