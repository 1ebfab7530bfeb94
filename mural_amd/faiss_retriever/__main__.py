"""Retrieval CLI / driver: `python -m tevatron.faiss_retriever` on the MI355X engine.

Mirrors /root/reference/src/tevatron/faiss_retriever/__main__.py:21-96 — same functions,
positional arguments, flags, defaults, log lines, input pickles and output bytes. What
changes is where the work happens: the search runs in libipx.so on the GPU, and the two
per-hit Python loops of the reference (row id -> doc id strings, and the ranking writer)
are a vectorised numpy gather and one C call.
"""
import ctypes as C
import glob
import logging
import mmap
import operator
import os
import pickle
from argparse import ArgumentParser
from itertools import chain

import numpy as np
from tqdm import tqdm

from .. import _lib
from .retriever import BaseFaissIPRetriever

logger = logging.getLogger(__name__)
logging.basicConfig(
    format="%(asctime)s - %(levelname)s - %(name)s -   %(message)s",
    datefmt="%m/%d/%Y %H:%M:%S",
    level=logging.INFO,
)


def _lookup_strings(p_lookup, all_indices):
    """[[str(p_lookup[x]) for x in row] for row in all_indices] as one '<U' array
    (__main__.py:27-28), without a Python iteration per hit. Negative row ids index from
    the end, as they do in the reference (a -1 padding id picks the last document)."""
    all_indices = np.asarray(all_indices)
    n = len(p_lookup)
    flat = all_indices.reshape(-1)
    if flat.size == 0:
        return np.array([[str(p_lookup[x]) for x in row] for row in all_indices])
    if (flat >= n).any() or (flat < -n).any():
        raise IndexError("list index out of range")
    flat = np.where(flat < 0, flat + n, flat)
    # the distinct rows hit, by a mark-and-scan over the lookup's length (no sort of nq*k ids)
    hit = np.zeros(n, dtype=bool)
    hit[flat] = True
    uniq = np.flatnonzero(hit)
    got = operator.itemgetter(*uniq.tolist())(p_lookup) if uniq.size > 1 else (p_lookup[int(uniq[0])],)
    table = np.array(got) if set(map(type, got)) == {str} else np.array([str(x) for x in got])
    slot = np.cumsum(hit, dtype=np.int64) - 1
    out = table[slot[flat]].reshape(all_indices.shape)
    # np.array(list of str lists) sizes the dtype to the longest string present: every table entry is
    # present in `out`, so the table's own width is that width (at least 1, as numpy gives '' -> '<U1')
    return out if out.dtype.itemsize >= 4 else out.astype("<U1")


def search_queries(retriever, q_reps, p_lookup, args):
    """__main__.py:21-29: chunked search when args.batch_size > 0, one call otherwise; row ids
    become str(p_lookup[row]) in a '<U' array shaped like the id matrix."""
    if args.batch_size > 0:
        all_scores, all_indices = retriever.batch_search(q_reps, args.depth, args.batch_size, args.quiet)
    else:
        all_scores, all_indices = retriever.search(q_reps, args.depth)
    return all_scores, _lookup_strings(p_lookup, all_indices)


def _write_ranking_py(corpus_indices, corpus_scores, q_lookup, ranking_save_file):
    """Generic path for score / id types the C writer does not take: the same ordering
    (stable sort on the score alone, descending) and the same f-string formatting."""
    with open(ranking_save_file, 'w') as out:
        for qid, row_scores, row_ids in zip(q_lookup, corpus_scores, corpus_indices):
            hits = sorted(zip(row_scores, row_ids), key=lambda hit: hit[0], reverse=True)
            out.writelines(f'{qid}\t{doc}\t{score}\n' for score, doc in hits)


def write_ranking(corpus_indices, corpus_scores, q_lookup, ranking_save_file):
    """__main__.py:32-38: per query, hits sorted by score descending (stable), one line
    f'{qid}\\t{idx}\\t{s}\\n' per hit. float32 scores with '<U' or integer ids go through
    ipx_write_ranking (byte-identical output); anything else takes the Python loop."""
    scores = np.asarray(corpus_scores)
    ids = np.asarray(corpus_indices)
    fast = (scores.dtype == np.float32 and scores.ndim == 2 and ids.shape == scores.shape
            and ids.dtype.kind in "Uiu" and scores.size > 0)
    if not fast:
        return _write_ranking_py(corpus_indices, corpus_scores, q_lookup, ranking_save_file)
    nq = min(len(q_lookup), scores.shape[0])      # zip() stops at the shortest
    k = scores.shape[1]
    qids = np.array([str(q) for q in q_lookup[:nq]]) if nq else np.array([], dtype="<U1")
    if qids.dtype.kind != "U":
        return _write_ranking_py(corpus_indices, corpus_scores, q_lookup, ranking_save_file)
    scores = np.ascontiguousarray(scores[:nq])
    ids = np.ascontiguousarray(ids[:nq])
    qw = qids.dtype.itemsize // 4
    if ids.dtype.kind == "U":
        dw, dptr, iptr = ids.dtype.itemsize // 4, C.c_void_p(ids.ctypes.data), None
    else:
        ids = ids.astype(np.int64, copy=False)
        dw, dptr, iptr = 0, None, C.c_void_p(ids.ctypes.data)
    _lib.check(_lib.lib().ipx_write_ranking(
        str(ranking_save_file).encode(), C.c_void_p(qids.ctypes.data), qw, dptr, dw, iptr,
        C.c_void_p(scores.ctypes.data), nq, k))


def _lookup_blob(p_lookup):
    """The lookup list as one UTF-8 blob + offsets (entry r = blob[off[r]:off[r+1] - 1], a newline behind every entry
    but the last) for ipx_write_ranking_lookup, or None when an entry's text holds a newline itself."""
    try:
        text = "\n".join(p_lookup)                       # all str: one C-level join
    except TypeError:
        text = "\n".join(map(str, p_lookup))
    joined = text.encode("utf-8")
    n = len(p_lookup)
    seps = np.flatnonzero(np.frombuffer(joined, dtype=np.uint8) == 10)
    if seps.size != max(n - 1, 0):
        return None
    off = np.empty(n + 1, dtype=np.int64)
    off[0] = 0
    off[1:n] = seps + 1
    off[n] = len(joined) + 1                               # as if a separator followed the last entry too
    return joined, off


def _lookup_blob_async(p_lookup):
    """_lookup_blob on a worker thread: the id table only depends on the shards' lookup lists, so it is put together
    while the GPU searches (the search waits inside the library with the GIL released). Returns a future."""
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ipx-lookup")
    fut = pool.submit(_lookup_blob, p_lookup)
    pool.shutdown(wait=False)
    return fut


def _write_ranking_from_rows(p_lookup, all_indices, all_scores, q_lookup, ranking_save_file, table=None):
    """search_queries' id mapping + write_ranking in one pass (text output of the CLI): the ranking file straight
    from the row ids the search returned, byte for byte what write_ranking(_lookup_strings(...), ...) writes.
    `table`: a future from _lookup_blob_async(p_lookup), if the caller started one.
    Returns False when the inputs are not of the kind the C writer takes (the caller then goes the long way)."""
    scores, rows = np.asarray(all_scores), np.asarray(all_indices)
    n = len(p_lookup)
    if not (scores.dtype == np.float32 and scores.ndim == 2 and rows.shape == scores.shape and rows.dtype == np.int64
            and scores.size > 0 and n > 0):
        return False
    if (rows >= n).any() or (rows < -n).any():
        raise IndexError("list index out of range")
    nq = min(len(q_lookup), scores.shape[0])
    qids = np.array([str(q) for q in q_lookup[:nq]]) if nq else np.array([], dtype="<U1")
    table = table.result() if table is not None else _lookup_blob(p_lookup)
    if qids.dtype.kind != "U" or table is None:
        return False
    blob, off = table
    scores = np.ascontiguousarray(scores[:nq])
    rows = np.ascontiguousarray(rows[:nq])
    _lib.check(_lib.lib().ipx_write_ranking_lookup(
        str(ranking_save_file).encode(), C.c_void_p(qids.ctypes.data), qids.dtype.itemsize // 4, blob,
        C.c_void_p(off.ctypes.data), 1, n, C.c_void_p(rows.ctypes.data), C.c_void_p(scores.ctypes.data), nq, scores.shape[1]))
    return True


class _PayloadSkippingReader:
    """File wrapper for pickle.Unpickler that does not read large byte payloads: a request of
    >= `threshold` bytes (the raw data of an ndarray: pickle writes it unframed and the C unpickler
    asks for it with one readinto() straight into the new bytes object) is recorded as
    (offset, size) and skipped with a seek — the destination stays untouched memory that is never
    looked at. Everything else (opcodes, the id list) is read normally. No peek(): the unpickler
    then never buffers ahead, so offsets are exact."""

    def __init__(self, f, threshold):
        self._f, self._threshold, self.skipped = f, threshold, []

    def read(self, n=-1):
        if n is not None and n >= self._threshold:
            self.skipped.append(None)           # a large read that cannot be skipped: give up the fast path
        return self._f.read(n)

    def readline(self):
        return self._f.readline()

    def readinto(self, b):
        n = len(b)
        if n >= self._threshold:
            self.skipped.append((self._f.tell(), n))
            self._f.seek(n, os.SEEK_CUR)
            return n
        return self._f.readinto(b)


def _pickle_load_mapped(path, threshold=1 << 20):
    """(reps, lookup, ...) with reps as a read-only np.memmap of the pickle file itself, or None when
    the file does not have the expected shape (then the caller unpickles the ordinary way). The page
    cache is the only host copy of the rows: ipx_add's staging threads read them from the mapping."""
    with open(path, 'rb') as f:
        rd = _PayloadSkippingReader(f, threshold)
        try:
            payload = pickle.Unpickler(rd).load()
        except Exception:
            return None
        size = f.seek(0, os.SEEK_END)
    if not isinstance(payload, tuple) or not payload or len(rd.skipped) != 1 or rd.skipped[0] is None:
        return None
    reps = payload[0]
    offset, nbytes = rd.skipped[0]
    if not (isinstance(reps, np.ndarray) and reps.dtype.kind == 'f' and reps.dtype.isnative and reps.ndim == 2
            and reps.flags.c_contiguous and reps.nbytes == nbytes and offset + nbytes <= size):
        return None
    shape, dtype = reps.shape, reps.dtype
    del reps                                    # its pages were never touched
    mapped = np.memmap(path, dtype=dtype, mode='r', offset=offset, shape=shape)
    try:
        mapped._mmap.madvise(mmap.MADV_SEQUENTIAL)
    except (AttributeError, OSError, ValueError):
        pass
    return (mapped,) + tuple(payload[1:])


def pickle_load(path):
    """__main__.py:41-48: a pickle holding (reps, lookup) or (reps, lookup, _, _); reps may be
    an ndarray (f16 / f32) or a list of rows (jax_encode.py:112-114) and comes back as an
    ndarray. Any other tuple length fails the way the reference does (UnboundLocalError).

    A float ndarray of >= 1 MiB is not read at all here: it comes back as a read-only mapping of
    the file (IPX_MMAP_LOAD=0 turns this off), so a 27 GB shard costs no host copy before ipx_add
    streams it to the GPU. The reference makes three (unpickle, np.array, faiss add)."""
    payload = _pickle_load_mapped(path) if os.environ.get('IPX_MMAP_LOAD', '1') != '0' else None
    if payload is None:
        with open(path, 'rb') as f:
            payload = pickle.load(f)
    if len(payload) not in (2, 4):
        raise UnboundLocalError("local variable 'reps' referenced before assignment")
    # the reference copies here (np.array); an unpickled ndarray is already private to this call,
    # so it is handed on as is — a 27 GB corpus shard is not duplicated in host memory
    reps = payload[0]
    return (reps if isinstance(reps, np.ndarray) else np.array(reps)), payload[1]


def pickle_save(obj, path):
    with open(path, 'wb') as f:
        pickle.dump(obj, f)


def _parse_args(argv=None):
    # the seven flags of __main__.py:57-64, same names, types and defaults
    parser = ArgumentParser()
    for flag in ('--query_reps', '--passage_reps', '--save_ranking_to'):
        parser.add_argument(flag, required=True)
    parser.add_argument('--batch_size', type=int, default=128)
    parser.add_argument('--depth', type=int, default=1000)
    for flag in ('--save_text', '--quiet'):
        parser.add_argument(flag, action='store_true')
    return parser.parse_args(argv)


def _load_index(pattern):
    """__main__.py:68-80: glob the shard pickles (file-system order, IndexError when nothing
    matches), size the index from the first one, add every shard, concatenate the lookups."""
    index_files = glob.glob(pattern)
    logger.info(f'Pattern match found {len(index_files)} files; loading them into index.')
    first = pickle_load(index_files[0])
    retriever = BaseFaissIPRetriever(first[0])
    shards = chain([first], map(pickle_load, index_files[1:]))
    if len(index_files) > 1:
        shards = tqdm(shards, desc='Loading shards into index', total=len(index_files))
    look_up = []
    for p_reps, p_lookup in shards:
        # float16 shards are widened to float32 on the device (same values as .astype);
        # float32 shards go down as they are (no host copy)
        retriever.add(p_reps if p_reps.dtype == np.float16 else p_reps.astype('float32', copy=False))
        look_up += p_lookup
    return retriever, look_up


def _assign_files(rows_per_file, world):
    """Contiguous groups of shard files, balanced by rows: owner[i] = rank that loads file i.
    Contiguous, so every rank's rows are one block [row_lo, row_hi) of the concatenated corpus."""
    total = sum(rows_per_file)
    owner, start = [], 0
    for n in rows_per_file:
        owner.append(min(world - 1, int((start + n / 2.0) * world / total)) if total else 0)
        start += n
    return owner


class _RankFailure:
    """What a rank tells the others when a load step failed on it (picklable: rank + the error's text)."""

    def __init__(self, rank, error):
        self.rank, self.text = rank, f"{type(error).__name__}: {error}"

    def __str__(self):
        return f"rank {self.rank} failed while loading: {self.text}"


def _raise_together(dist, rank, world, failure):
    """All ranks exchange whether their load step failed; if any did, every rank raises (the failing ones their
    own exception, the others a RuntimeError naming the first failing rank), so no rank is left in a collective."""
    said = [None] * world
    dist.all_gather_object(said, _RankFailure(rank, failure) if failure is not None else None)
    if failure is not None:
        raise failure
    for s in said:
        if s is not None:
            raise RuntimeError(str(s))


def _main_sharded(args, engine_factory=None):
    """`python -m torch.distributed.run --nproc-per-node G -m tevatron.faiss_retriever ...`: one process
    per GPU, the shard files dealt to the ranks in contiguous groups, every rank searches all queries on its
    rows, candidates all-gathered (RCCL) and merged on the GPU (mural_amd.sharded), rank 0 writes the
    ranking. The output is the single-process output: global row ids are positions in the concatenation of
    the shards in glob order, exactly what the reference's add loop produces (__main__.py:74-80), and the
    merge breaks ties on them. This replaces the reference's per-shard runs + reducer (docs/retrieval.md:31-53).
    `engine_factory(d)` injects the per-rank engine (CPU tests); the default is the HIP engine."""
    import torch
    import torch.distributed as dist
    from ..sharded import ShardedFlatIP
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    index_files = glob.glob(args.passage_reps)
    logger.info(f'Pattern match found {len(index_files)} files; loading them into index.')
    index_files[0]                               # IndexError on every rank when nothing matches (__main__.py:71)
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if engine_factory is None:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 reads every shard's ids; the rows themselves are mapped, not read (pickle_load)
        # a rank that fails while the others wait in a collective would leave them there until the launcher's
        # timeout: every load step below ends in an exchange of its outcome, so all ranks raise together and the
        # message names the rank and the error (the reference, one process, just raises: __main__.py:74-80)
        meta, look_up, failure = [None], [], None
        if rank == 0:
            try:
                rows, dim = [], None
                for f in index_files:
                    reps, lookup = pickle_load(f)
                    rows.append(int(reps.shape[0]))
                    dim = int(reps.shape[1]) if dim is None else dim
                    look_up += lookup
                meta = [(rows, dim)]
            except Exception as e:                       # noqa: BLE001 (re-raised below, on every rank)
                failure, meta = e, [_RankFailure(0, e)]
        dist.broadcast_object_list(meta, src=0)
        if isinstance(meta[0], _RankFailure):
            raise failure if failure is not None else RuntimeError(str(meta[0]))
        rows, dim = meta[0]
        owner = _assign_files(rows, world)
        starts = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
        mine = [i for i, r in enumerate(owner) if r == rank]
        row_lo = int(starts[mine[0]]) if mine else int(starts[-1])
        try:
            index = ShardedFlatIP(dim, row_lo=row_lo, device=None if engine_factory else local,
                                  engine=engine_factory(dim) if engine_factory else None)
            for i in mine:
                p_reps, _ = pickle_load(index_files[i])
                index.add_local(p_reps if p_reps.dtype == np.float16 else p_reps.astype('float32', copy=False))

            q_reps, q_lookup = pickle_load(args.query_reps)
            q_reps = q_reps.astype('float32')
        except Exception as e:                           # noqa: BLE001
            failure = e
        _raise_together(dist, rank, world, failure)
        table = _lookup_blob_async(look_up) if rank == 0 and args.save_text and look_up else None
        logger.info('Index Search Start')
        # chunking never changes a result here (a score does not depend on its batch): large chunks
        chunk = q_reps.shape[0] if args.batch_size <= 0 else max(args.batch_size, 8192)
        all_scores, all_indices = [], []
        for lo in range(0, q_reps.shape[0], max(chunk, 1)):
            D, I = index.search(q_reps[lo:lo + chunk], args.depth)
            if rank == 0:
                all_scores.append(D.cpu().numpy())
                all_indices.append(I.cpu().numpy())
        logger.info('Index Search Finished')
        if rank == 0:
            k = args.depth
            all_scores = np.concatenate(all_scores) if all_scores else np.empty((0, k), np.float32)
            all_indices = np.concatenate(all_indices) if all_indices else np.empty((0, k), np.int64)
            if args.save_text:
                if not _write_ranking_from_rows(look_up, all_indices, all_scores, q_lookup, args.save_ranking_to, table):
                    write_ranking(_lookup_strings(look_up, all_indices), all_scores, q_lookup, args.save_ranking_to)
            else:
                pickle_save((all_scores, _lookup_strings(look_up, all_indices)), args.save_ranking_to)
        dist.barrier()
    finally:
        if own_group:
            dist.destroy_process_group()


def _is_rank_of_a_launch():
    """True inside a process that torch.distributed.run (or _spawn_ranks below) started as one rank of several: the
    rendezvous variables must all be there. A stray WORLD_SIZE alone (a batch scheduler's, say) does not switch the
    CLI to the distributed path — the reference ignores that variable."""
    env = os.environ
    return int(env.get("WORLD_SIZE", "1") or 1) > 1 and "RANK" in env and "MASTER_PORT" in env


def _spawn_ranks(argv, n_ranks, entry=None):
    """One command for N GPUs (IPX_GPUS=N): start N fresh processes, one per GPU, each running this CLI as a rank
    of a torch.distributed group (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set in their environment), wait for
    them, return the first non-zero exit status. The parent never touches the GPU: it only imports, parses the
    arguments and waits. `entry` is the command prefix of a rank (tests inject a CPU engine through it)."""
    import socket
    import subprocess
    import sys
    import time
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    entry = entry or [sys.executable, "-m", "mural_amd.faiss_retriever"]
    procs = []
    for rank in range(n_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), IPX_GPUS="1")
        procs.append(subprocess.Popen(entry + list(argv), env=env))
    status = 0
    # IPX_SPAWN_TIMEOUT (seconds, optional): a launch whose ranks have not all finished by then is ended with
    # status 124 instead of waiting for ever on a collective that cannot complete
    limit = float(os.environ.get("IPX_SPAWN_TIMEOUT", "0") or 0)
    deadline = time.monotonic() + limit if limit > 0 else None
    try:
        for proc in procs:
            try:
                rc = proc.wait(timeout=max(deadline - time.monotonic(), 0.0) if deadline else None)
            except subprocess.TimeoutExpired:
                rc = 124
            if rc != 0 and status == 0:
                status = rc
                for other in procs:           # a rank failed: the others would wait in a collective for ever
                    if other.poll() is None:
                        other.terminate()
    finally:
        for proc in procs:
            if proc.poll() is None:
                proc.kill()
    return status


def main(argv=None):
    import sys
    args = _parse_args(argv)
    if _is_rank_of_a_launch():
        return _main_sharded(args)
    n_gpus = int(os.environ.get("IPX_GPUS", "1") or 1)
    if n_gpus > 1:
        # the reference's command line, unchanged, on N GPUs of this node (src/pretrain_finetune1.sh:218-228)
        status = _spawn_ranks(sys.argv[1:] if argv is None else argv, n_gpus)
        if status != 0:
            raise SystemExit(status)
        return
    retriever, look_up = _load_index(args.passage_reps)

    q_reps, q_lookup = pickle_load(args.query_reps)
    q_reps = q_reps.astype('float32')

    logger.info('Index Search Start')
    if args.save_text:
        # text output: the row ids go straight to the writer together with the lookup table (put together while the
        # GPU searches); the [nq, depth] array of id strings that search_queries builds is only needed for the
        # pickle output
        table = _lookup_blob_async(look_up) if look_up else None
        if args.batch_size > 0:
            all_scores, all_indices = retriever.batch_search(q_reps, args.depth, args.batch_size, args.quiet)
        else:
            all_scores, all_indices = retriever.search(q_reps, args.depth)
        logger.info('Index Search Finished')
        if not _write_ranking_from_rows(look_up, all_indices, all_scores, q_lookup, args.save_ranking_to, table):
            write_ranking(_lookup_strings(look_up, all_indices), all_scores, q_lookup, args.save_ranking_to)
        return
    all_scores, psg_indices = search_queries(retriever, q_reps, look_up, args)
    logger.info('Index Search Finished')
    pickle_save((all_scores, psg_indices), args.save_ranking_to)


if __name__ == '__main__':
    main()
