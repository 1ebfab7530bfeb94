"""End-to-end timing of the drop-in CLI at BASELINE configs[2] scale (8.8M x 768 fp32 rows in 4 shard pickles,
6980 queries, --depth 1000, --save_text): the phases of `python -m tevatron.faiss_retriever`, timed one by one
with the same functions main() calls. Synthetic shards are written to /tmp first (not timed)."""
import os, sys, time, pickle, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mural_amd.faiss_retriever import __main__ as drv

rows_per_shard = int(sys.argv[1]) if len(sys.argv) > 1 else 2_200_000
nshards = int(sys.argv[2]) if len(sys.argv) > 2 else 4
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
d, nq = 768, 6980
tmp = "/tmp/ipx_e2e"
os.makedirs(tmp, exist_ok=True)
rng = np.random.default_rng(0)
base = rng.standard_normal((rows_per_shard, d), dtype=np.float32)
t0 = time.perf_counter()
for s in range(nshards):
    X = base * np.float32(1.0 + 0.01 * s) if s else base
    with open(f"{tmp}/corpus_emb{s}.pkl", "wb") as f:
        pickle.dump((X, [f"B{s}{i:08d}" for i in range(rows_per_shard)]), f)
del base, X
with open(f"{tmp}/q.pkl", "wb") as f:
    pickle.dump((rng.standard_normal((nq, d), dtype=np.float32), [str(1000000 + i) for i in range(nq)]), f)
print(f"wrote {nshards} shards of {rows_per_shard} rows in {time.perf_counter()-t0:.1f} s", flush=True)

args = argparse.Namespace(batch_size=128, depth=depth, quiet=True)
t = [time.perf_counter()]
retriever, look_up = drv._load_index(f"{tmp}/corpus_emb*.pkl"); t.append(time.perf_counter())
q_reps, q_lookup = drv.pickle_load(f"{tmp}/q.pkl"); q_reps = q_reps.astype('float32'); t.append(time.perf_counter())
table = drv._lookup_blob_async(look_up)          # as main() does: the id table is put together while the GPU searches
all_scores, all_indices = retriever.batch_search(q_reps, args.depth, args.batch_size, args.quiet); t.append(time.perf_counter())
# --save_text: row ids + lookup table straight to the C writer (ipx_write_ranking_lookup)
assert drv._write_ranking_from_rows(look_up, all_indices, all_scores, q_lookup, f"{tmp}/rank.tsv", table); t.append(time.perf_counter())
# the pickle-output route, for comparison: the [nq, depth] array of id strings, then ipx_write_ranking
psg = drv._lookup_strings(look_up, all_indices); t.append(time.perf_counter())
drv.write_ranking(psg, all_scores, q_lookup, f"{tmp}/rank2.tsv"); t.append(time.perf_counter())
assert open(f"{tmp}/rank.tsv", "rb").read() == open(f"{tmp}/rank2.tsv", "rb").read()
names = ["load index (unpickle + ipx_add)", "load queries", "batch_search", "text output: rows -> ranking file",
         "(old route) row ids -> doc id strings", "(old route) write_ranking"]
gb = nshards * rows_per_shard * d * 4 / 1e9
for n, a, b in zip(names, t, t[1:]):
    print(f"{n:34s} {b-a:7.2f} s")
print(f"total {t[-1]-t[0]:.2f} s for {gb:.1f} GB of rows, {nq} queries, depth {depth}; storage {os.environ.get('IPX_STORAGE','f32')}; "
      f"ranking file {os.path.getsize(tmp + '/rank.tsv')/1e6:.0f} MB", flush=True)
with open(f"{tmp}/rank.tsv") as f:
    print(f.readline().rstrip(), "|", f.readline().rstrip())
for fn in os.listdir(tmp):
    os.remove(os.path.join(tmp, fn))
