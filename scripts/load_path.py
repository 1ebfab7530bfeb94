"""Index load path timing: pickle -> host array -> ipx_add (pinned double-buffered
staging) on the GPU box, with the plain unpickle and with the mapped payload. Writes a synthetic shard
pickle to /tmp first."""
import os, sys, time, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mural_amd.faiss_retriever.__main__ import pickle_load
from mural_amd.faiss_retriever.retriever import BaseFaissIPRetriever
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X = np.random.default_rng(0).standard_normal((rows, 768), dtype=np.float32)
path = "/tmp/ipx_shard.pkl"
with open(path, "wb") as f:
    pickle.dump((X, [str(i) for i in range(rows)]), f)
gb = X.nbytes / 1e9
r = BaseFaissIPRetriever(X)
r.add(X[:1000])            # warm up: context, kernels, staging buffers
del X
for mode in ("0", "1", "1"):
    os.environ["IPX_MMAP_LOAD"] = mode
    t0 = time.perf_counter(); reps, look = pickle_load(path); t1 = time.perf_counter()
    r = BaseFaissIPRetriever(reps)
    r.index.reserve(rows)
    t2 = time.perf_counter(); r.add(reps.astype('float32', copy=False)); t3 = time.perf_counter()
    print(f"{gb:.2f} GB shard, IPX_MMAP_LOAD={mode} ({type(reps).__name__}): pickle_load {t1-t0:.2f} s ({gb/(t1-t0):.2f} GB/s), "
          f"add {t3-t2:.2f} s ({gb/(t3-t2):.2f} GB/s host->HBM incl. pack), together {t3-t2+t1-t0:.2f} s = {gb/(t3-t2+t1-t0):.2f} GB/s", flush=True)
    if mode == "0":
        h = reps.astype(np.float16)
        t4 = time.perf_counter(); r.add(h); t5 = time.perf_counter()
        print(f"float16 rows: add {t5-t4:.2f} s ({h.nbytes/1e9/(t5-t4):.2f} GB/s over PCIe, widened on the device)")
        del h
    del r, reps, look
os.remove(path)
