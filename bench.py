"""bench.py — QPS of exact top-100 inner-product search over an 8.8M x 768 fp32 corpus.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one search of all 6980 queries (BASELINE.json: MS MARCO-scale corpus, top-100),
with the corpus and the queries already resident in HBM. With N > 1 the corpus is row-sharded
over the ranks (strong scaling: the total corpus is fixed), every rank searches all queries on
its shard, the (score, id) candidates are all-gathered over RCCL and merged on the GPU.

`--workload cfg5` times BASELINE.json configs[4] instead (the default run carries the same measurement as
`extra.cfg5`): every rank holds a 12.5M x 768 bf16 shard generated on its GPU (weak scaling: 8 ranks = the 100M-row
corpus), batches of 1..64 queries, each search timed end to end INCLUDING the candidate exchange (one RCCL
all-gather) and the merge. With one rank the collective runs on a 1-rank NCCL group.

Rank 0 prints ONE JSON line on stdout; see DESIGN.md §Measurement for the roofline arithmetic. Every rank writes a
line per stage to stderr (index built / headline timed / cfg5); `IPX_BENCH_WATCHDOG=<seconds>` makes a run that is
still going after that long print every thread's Python stack and exit.
"""
import hashlib
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# cpu_baseline alternates OpenBLAS sgemm blocks (its own thread pool) with OpenMP heap stages: OpenMP workers that
# spin between parallel regions take the cores from the sgemm threads (measured 3x slower). Read by libgomp when it
# loads (with torch), so it is set before any import that brings it in.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: BF16 dense ~2.5 PF
PEAK_HBM_GBS = 8000.0             # same guide: HBM3E 8 TB/s peak (about 6.3 TB/s achievable)
BLOCK_ROWS = 200_000              # synthetic corpus is generated in blocks, seed = 1234 + block


def synth_block(torch, dev, block, d):
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + block)
    return torch.randn((BLOCK_ROWS, d), generator=g, device=dev, dtype=torch.float32)


def cpu_baseline(X_sample, Q, k, n_total):
    """The oracle (faiss-algorithm restatement: blocked sgemm + heap) on this box's host cores,
    on a bounded sample of the same workload: all queries against the first rows of the same
    corpus; search time scales linearly in the row count."""
    import numpy as np
    from oracle import oracle as O
    ix = O.FlatIPOracle(Q.shape[1])
    ix.add(X_sample)
    t = time.perf_counter()
    ix.search(Q, k)
    dt = time.perf_counter() - t
    scale = n_total / X_sample.shape[0]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    threads = min(O.num_threads(), cores)                # OpenMP threads of the heap stages (and OpenBLAS's default)
    return {"value": Q.shape[0] / (dt * scale), "unit": "queries/s", "cores": threads, "kind": "port",
            "sample": f"all {Q.shape[0]} queries x first {X_sample.shape[0]} of {n_total} rows, top-{k}, "
                      f"{dt:.1f} s measured on {threads} threads (OpenBLAS sgemm blocks + OpenMP heap stages over "
                      f"queries, as faiss runs them), scaled x{scale:.1f} in rows; faiss IndexFlatIP restatement "
                      f"(real faiss is not installed on this image)"}


def kernel_source_hash():
    """sha256 of the header the fp32 scan kernel lives in: profiles/latest_scan_f32_pmc.json records the hash of the
    source it profiled, and its traffic figure is only quoted while the two agree."""
    with open(os.path.join(ROOT, "mural_amd", "csrc", "ipx_kernels.h"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


XGMI_LINK_GBS = 153.0             # same guide: 7 xGMI links per GPU, ~153 GB/s each (point to point)


def _fill(torch, dev, ix, rows, d, seed, kind="gauss", sigma=0.5, block=500_000):
    """Synthetic rows generated on the device and added block by block. kind: gauss | lognormal (row norms spread by
    exp(sigma * N(0,1)): unnormalised representations) | l2norm (unit rows) | dups (gauss + 40 exact duplicates per block)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ix.reserve(rows)
    for lo in range(0, rows, block):
        x = torch.randn((min(block, rows - lo), d), generator=g, device=dev, dtype=torch.float32)
        if kind == "lognormal":
            x *= torch.exp(sigma * torch.randn((x.shape[0], 1), generator=g, device=dev))
        elif kind == "l2norm":
            x /= x.norm(dim=1, keepdim=True)
        elif kind == "dups":
            x[1000:1040] = x[1000:1001]
        ix.add(x)
    torch.cuda.synchronize()
    return g


def _time_search(torch, fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def run_cfg4(torch, dev, d, steps=3, n=1_800_000, nq=8192):
    """BASELINE.json configs[3] on configs[1]'s corpus: MURAL multi-aspect queries (aspect embeddings + logits, fused
    on the device, modeling/mul_grain.py:97-110) over 1.8M x 768 rows, bf16 MFMA scan + exact fp32 rescoring."""
    from mural_amd.faiss_retriever.retriever import FlatIPIndex
    A, k = 3, 100
    ix = FlatIPIndex(d, device=dev.index, storage="bf16scan")
    g = _fill(torch, dev, ix, n, d, 2234)
    E = torch.randn((nq, A, d), generator=g, device=dev, dtype=torch.float32)
    logits = torch.randn((nq, A), generator=g, device=dev, dtype=torch.float32)
    ix.set_timing(True)
    t_bf, (D, I) = _time_search(torch, lambda: ix.search_aspect_device(E, logits, k, logits=True), steps, 1)
    tm, bf = ix.last_timing(), ix.last_bf16_stats()
    ix.set_param("f32_scan", 1)
    t_f32, (D0, I0) = _time_search(torch, lambda: ix.search_aspect_device(E, logits, k, logits=True), 1, 1)
    tm0 = ix.last_timing()
    # the reference fuses at encode time (the pickles hold the fused vectors): the same search on pre-fused queries
    fused = (torch.softmax(logits, dim=1)[:, :, None] * E).sum(1)
    D1, I1 = ix.search_device(fused, k)
    torch.cuda.synchronize()
    ix.set_param("f32_scan", 0)
    flops = 2.0 * nq * n * d
    res = {
        "what": f"BASELINE.json configs[3]: {nq} MURAL multi-aspect queries ([{nq}, {A}, {d}] aspect embeddings + logits, soft-max "
                f"and weighted sum fused into the query pack on the device) over a {n} x {d} fp32 corpus with a bf16 copy: "
                f"bf16 MFMA scan + exact fp32 rescoring, top-{k}; compared bitwise with the fp32 scan of the same aspect "
                f"queries, and with the search on queries pre-fused by torch (fp32 softmax + sum: same ranking up to near-ties)",
        "queries_per_s": nq / t_bf, "ms_per_step": t_bf * 1e3, "pack_fuse_ms": round(tm["pack_ms"], 3), "scan_ms": tm["scan_ms"],
        "merge_ms": round(tm["merge_ms"], 3), "rescore_ms": round(bf["rescore_ms"], 3), "candidates": bf["candidates"],
        "repeated_queries": bf["fallback_queries"],
        "bit_identical_to_f32_scan": bool(torch.equal(I, I0)) and bool(torch.equal(D.view(torch.int32), D0.view(torch.int32))),
        "f32_scan_ms_per_step": t_f32 * 1e3, "f32_scan_kernel_ms": tm0["scan_ms"],
        "prefused_ids_equal_frac": float((I0 == I1).float().mean()),
        "prefused_max_rel_score_diff": float(((D0 - D1).abs() / D1.abs().clamp_min(1e-30)).max()),
        "roofline": {"bound": "mfma", "achieved": flops / (tm["scan_ms"] * 1e-3) / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": flops / (tm["scan_ms"] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                     "kernel": "scan_bf16_w16_kernel<1, true>", "flops_per_launch": flops}}
    del ix
    torch.cuda.empty_cache()
    return res


def run_projected_8gpu(torch, dev, d, n_total, nq, steps=3):
    """What one of 8 ranks does per search of the headline workload, measured on this GPU, plus a model of the one
    collective: NOT a measured 8-GPU number."""
    from mural_amd import _lib
    from mural_amd.faiss_retriever.retriever import FlatIPIndex
    import ctypes as C
    world = 8
    rows = n_total // world
    ix = FlatIPIndex(d, device=dev.index, storage="bf16scan")
    _fill(torch, dev, ix, rows, d, 1234)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    Q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)
    ix.set_timing(True)
    L = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for k in (100, 1000):
        block = int(L.ipx_packed_block_bytes(nq, k))
        mine = torch.empty(block, dtype=torch.uint8, device=dev)
        I = mine[: nq * k * 8].view(torch.int64).view(nq, k)
        D = mine[nq * k * 8: nq * k * 12].view(torch.float32).view(nq, k)
        row = {"block_bytes_per_rank": block}
        for scan in ("f32", "bf16scan"):
            ix.set_param("f32_scan", int(scan == "f32"))
            t, _ = _time_search(torch, lambda: ix.search_device(Q, k, out=(D, I)), steps, 1)
            row[scan + "_shard_search_ms"] = t * 1e3
            row[scan + "_scan_kernel_ms"] = ix.last_timing()["scan_ms"]
        # the merge every rank runs on the gathered blocks: 8 blocks of this shape (this shard's lists, ids shifted per
        # block as the other shards' id bases would)
        allb = torch.empty(world * block, dtype=torch.uint8, device=dev)
        for r in range(world):
            allb[r * block: (r + 1) * block].copy_(mine)
            allb[r * block: r * block + nq * k * 8].view(torch.int64).add_(r * rows)
        oD = torch.empty((nq, k), dtype=torch.float32, device=dev)
        oI = torch.empty((nq, k), dtype=torch.int64, device=dev)

        def merge():
            _lib.check(L.ipx_merge_topk_packed(C.c_void_p(allb.data_ptr()), world, nq, k, k, C.c_void_p(oD.data_ptr()),
                                               C.c_void_p(oI.data_ptr()), dev.index, st))
        t_merge, _ = _time_search(torch, merge, 10, 2)
        # xGMI is point to point: each of the 7 peers' blocks arrives over its own link
        t_gather = block / (XGMI_LINK_GBS * 1e9) + 20e-6
        row.update({"merge_ms": t_merge * 1e3, "modelled_allgather_ms": t_gather * 1e3})
        for scan in ("f32", "bf16scan"):
            total = row[scan + "_shard_search_ms"] * 1e-3 + t_gather + t_merge
            row[scan + "_projected_queries_per_s"] = nq / total
            row[scan + "_projected_ms_per_step"] = total * 1e3
        out[f"top{k}"] = row
    ix.set_param("f32_scan", 0)
    del ix
    torch.cuda.empty_cache()
    return {"projection": True,
            "what": f"NOT measured on 8 GPUs. One rank's share of the headline workload measured on this GPU ({rows} of {n_total} rows, "
                    f"all {nq} queries: search into the packed exchange block) + ipx_merge_topk_packed over 8 such blocks, "
                    f"measured + the all-gather MODELLED as block bytes / {XGMI_LINK_GBS:.0f} GB/s (each peer's block over its own "
                    f"xGMI link) + 20 us; sum = projected time per step of the 8-rank job (ranks run the same work in parallel)",
            "ranks": world, "rows_per_rank": rows, **out}


def run_norm_spread(torch, dev, d, rows=1_000_000, nq=6980, k=100, steps=3):
    """The bf16 scan's margin rule on rows whose norms spread (MURAL representations are unnormalised, SURVEY 8d names an
    L2-normalised and a duplicates variant too): same rows x queries shape per variant, Gaussian as the yardstick."""
    from mural_amd.faiss_retriever.retriever import FlatIPIndex
    res = {"what": f"{rows} x {d} rows, {nq} queries, top-{k}, storage bf16scan: candidates kept, queries sent to the repeat pass / "
                   f"answered by the fp32 scan, times (after 8 untimed searches: the index learns a longer list from its first "
                   f"repeat passes, `first_search` shows where it started), bitwise comparison with the fp32 scan of the same "
                   f"index; `vs_gaussian` = queries/s relative to the Gaussian variant"}
    base = None
    for kind in ("gauss", "lognormal", "l2norm", "dups"):
        ix = FlatIPIndex(d, device=dev.index, storage="bf16scan")
        g = _fill(torch, dev, ix, rows, d, 3234, kind=kind)
        Q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)
        if kind == "l2norm":
            Q /= Q.norm(dim=1, keepdim=True)
        ix.set_timing(True)
        first = None
        for _ in range(8):       # the index learns its list length from the first searches (kc_boost): timed in steady state
            ix.search_device(Q, k)
            st0 = ix.last_bf16_stats()
            first = first or (st0["candidates"], st0["fallback_queries"])
        t, (D, I) = _time_search(torch, lambda: ix.search_device(Q, k), steps, 0)
        tm, bf = ix.last_timing(), ix.last_bf16_stats()
        ix.set_param("f32_scan", 1)
        D0, I0 = ix.search_device(Q, k)
        torch.cuda.synchronize()
        if base is None:
            base = nq / t
        res[kind] = {"queries_per_s": nq / t, "vs_gaussian": (nq / t) / base, "ms_per_step": t * 1e3, "scan_ms": tm["scan_ms"],
                     "rescore_ms": round(bf["rescore_ms"], 3), "candidates": bf["candidates"], "outlier_rows": bf["outlier_rows"],
                     "repeated_queries": bf["fallback_queries"], "exact_path_queries": bf["exact_queries"],
                     "first_search": {"candidates": first[0], "repeated_queries": first[1]},
                     "bit_identical": bool(torch.equal(I, I0)) and bool(torch.equal(D.view(torch.int32), D0.view(torch.int32)))}
        del ix
        torch.cuda.empty_cache()
    res["lognormal"]["sigma"] = 0.5
    return res


CFG5_ROWS_PER_RANK = 12_500_000    # BASELINE.json configs[4]: 100M x 768 bf16 over 8 GPUs
CFG5_BLOCK = 500_000


def run_cfg5(torch, dist, dev, local_rank, rank, world, d, k, rows_per_rank, steps=20, warmup=3):
    """BASELINE.json configs[4] on this rank's shard: nq in {1..64}, end to end with exchange + merge."""
    from mural_amd.sharded import ShardedFlatIP
    lo = rank * rows_per_rank
    index = ShardedFlatIP(d, row_lo=lo, device=local_rank, storage="bf16")
    ix = index.engine.index
    ix.reserve(rows_per_rank)
    t0 = time.perf_counter()
    for b in range(0, rows_per_rank, CFG5_BLOCK):
        g = torch.Generator(device=dev)
        g.manual_seed(777 + (lo + b) // CFG5_BLOCK)           # counter-based: block id of the global corpus
        n_b = min(CFG5_BLOCK, rows_per_rank - b)
        index.add_local(torch.randn((n_b, d), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    Q = torch.randn((64, d), generator=g, device=dev, dtype=torch.float32)
    ix.set_timing(True)
    shard_bytes = rows_per_rank * d * 2

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def run(q, exchange):
        # wall time as a caller sees it (the engine records no HIP events: ipx_set_timing is off by default) ...
        ix.set_timing(False)
        for _ in range(warmup):
            index.search(q, k, exchange=exchange)
        barrier()
        t = time.perf_counter()
        for _ in range(steps):
            index.search(q, k, exchange=exchange)
        barrier()
        el = time.perf_counter() - t
        ix.check()
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        # ... and the scan kernel's duration from HIP events on its stream, over the same number of searches
        ix.set_timing(True)
        scans = []
        for _ in range(steps):
            index.search(q, k, exchange=exchange)
            scans.append(ix.last_timing()["scan_ms"])
        barrier()
        return el / steps, sum(scans) / len(scans)

    rows = []
    for bnq in (1, 2, 4, 8, 16, 32, 64):
        q = Q[:bnq].contiguous()
        t_local, _ = run(q, False)
        t_e2e, scan_ms = run(q, True)
        rows.append({"queries": bnq, "ms_per_search": t_e2e * 1e3, "queries_per_s": bnq / t_e2e,
                     "local_search_ms": t_local * 1e3, "exchange_merge_us": (t_e2e - t_local) * 1e6,
                     "scan_ms": scan_ms, "shard_bytes_streamed": shard_bytes,
                     "roofline": {"bound": "hbm", "achieved": shard_bytes / (scan_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                                  "unit": "GB/s", "frac": shard_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                  "kernel": "scan_bf16_kernel<1,2>"},
                     "end_to_end_hbm_frac": shard_bytes / t_e2e / 1e9 / PEAK_HBM_GBS})
    del index, ix
    torch.cuda.empty_cache()
    return {"what": f"BASELINE.json configs[4]: {world} rank(s) x {rows_per_rank} x {d} bf16 rows generated on the device "
                    f"(storage bf16), top-{k}, {steps} searches per batch size; ms_per_search is end to end on every rank "
                    f"(local scan + rescoring, one all-gather of packed (ids, scores) blocks over "
                    f"{'RCCL' if world > 1 else ('a 1-rank NCCL group' if dist.is_initialized() else 'no process group: merge only')}, GPU merge), max over ranks; "
                    f"exchange_merge_us = that minus the same search without exchange",
            "ranks": world, "rows_per_rank": rows_per_rank, "total_rows": world * rows_per_rank,
            "shard_build_s": round(t_build, 2), "batches": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=8_800_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-splits", type=int, default=0, help="diagnostic: cap the corpus splits per query block")
    ap.add_argument("--storage", choices=["f32", "bf16scan", "bf16"], default="bf16scan",
                    help="what the index keeps: f32 rows; f32 rows + a bf16 copy (default: serves the fp32 headline "
                         "scan and the bf16 extras from one index); bf16 rows only")
    ap.add_argument("--headline", choices=["f32", "bf16scan"], default="f32",
                    help="which scan the timed K steps use on a bf16scan index (results are bit-identical)")
    ap.add_argument("--no-extras", action="store_true", help="skip the bf16-scan and small-batch measurements")
    ap.add_argument("--workload", choices=["headline", "cfg5"], default="headline",
                    help="cfg5: only BASELINE.json configs[4] (12.5M x 768 bf16 shard per rank, batches of 1..64 queries)")
    ap.add_argument("--cfg5-rows", type=int, default=CFG5_ROWS_PER_RANK, help="rows per rank of the cfg5 shard")
    ap.add_argument("--extras-scale", type=float, default=0.0,
                    help="run extra.cfg4 / projected_8gpu / norm_spread with their row counts scaled by this factor "
                         "(the full-size default run carries them at scale 1; a reduced --rows run only when asked)")
    args = ap.parse_args()

    # stdout carries ONE line, the JSON: libraries that print there (RCCL's version banner at communicator set-up) are
    # sent to stderr for the whole run, and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import numpy as np
    import torch
    import torch.distributed as dist
    from mural_amd.sharded import ShardedFlatIP, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("IPX_BENCH_WATCHDOG"):
        # a run that is still going after this many seconds prints every thread's Python stack to stderr and exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["IPX_BENCH_WATCHDOG"]), exit=True)

    def stage(msg):
        """Progress on stderr (the JSON line owns stdout): where a run that stops has got to."""
        print(f"bench[{rank}/{world}] {time.strftime('%H:%M:%S')} {msg}", file=sys.stderr, flush=True)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        # one rank: a 1-rank NCCL group on a free port, so the cfg5 measurement runs its all-gather through RCCL; if
        # that cannot be set up the run goes on without a group (the exchange then is the merge alone)
        import datetime
        import socket
        try:
            if "TORCHELASTIC_RUN_ID" in os.environ or "MASTER_PORT" in os.environ:
                # launched by torch.distributed.run with one rank: its agent holds the store, a tcp:// rendezvous of our
                # own would wait for a server that nobody starts
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
            else:
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    port = sock.getsockname()[1]
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev,
                                        timeout=datetime.timedelta(seconds=120))
        except Exception as exc:      # noqa: BLE001 - any failure here only removes the 1-rank collective
            print(f"bench: no 1-rank process group ({exc}); continuing without", file=sys.stderr, flush=True)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload == "cfg5":
        res = run_cfg5(torch, dist, dev, local_rank, rank, world, args.dim, args.k, args.cfg5_rows,
                       steps=max(args.steps, 1), warmup=max(args.warmup, 1))
        if rank == 0:
            b64 = res["batches"][-1]
            emit({
                "metric": "QPS exact top-100 IP search, 100M×768 bf16 corpus row-sharded @8 GPUs, batches of 1..64 queries; % HBM roofline",
                "value": b64["queries_per_s"], "unit": "queries/s", "n_gpus": world, "steps": max(args.steps, 1),
                "warmup": max(args.warmup, 1), "ms_per_step": b64["ms_per_search"], "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16 scan + f32 rescoring on bf16 rows",
                "data": "synthetic: torch.randn -> bf16 on device, corpus seed 777+global block, query seed 4321",
                "config": {"workload": f"BASELINE.json configs[4]: {world} x {args.cfg5_rows}x{args.dim} bf16 shard(s), "
                                       f"64-query batches (value), top-{args.k}; all batch sizes in extra.cfg5",
                           "rows_per_rank": args.cfg5_rows, "dim": args.dim, "k": args.k,
                           "parallelism": f"row-shard x{world}"},
                "roofline": b64["roofline"], "cpu_baseline": None, "extra": {"cfg5": res}})
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    os.environ["IPX_STORAGE"] = args.storage
    n, d, nq, k = args.rows, args.dim, args.queries, args.k
    lo, hi = shard_bounds(n, world, rank)
    index = ShardedFlatIP(d, row_lo=lo, device=local_rank)
    ix = index.engine.index
    ix.reserve(hi - lo)
    t_build = time.perf_counter()
    sample = None
    for b in range(lo // BLOCK_ROWS, (hi + BLOCK_ROWS - 1) // BLOCK_ROWS):
        blk = synth_block(torch, dev, b, d)
        s, e = max(lo, b * BLOCK_ROWS), min(hi, (b + 1) * BLOCK_ROWS)
        part = blk[s - b * BLOCK_ROWS: e - b * BLOCK_ROWS]
        index.add_local(part)
        if rank == 0 and sample is None and not args.no_cpu_baseline:
            sample = []
        if rank == 0 and sample is not None and sum(x.shape[0] for x in sample) < args.cpu_sample_rows:
            sample.append(part[: args.cpu_sample_rows - sum(x.shape[0] for x in sample)].cpu().numpy())
        torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    stage(f"index built: rows [{lo}, {hi}) in {t_build:.1f} s")
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    Q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)

    ix.set_timing(True)
    if args.max_splits:
        ix.set_param("max_splits", args.max_splits)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(queries, steps, warmup):
        """W untimed + K timed searches; returns (max-over-ranks wall seconds, scan ms list, D, I)."""
        for _ in range(warmup):
            index.search(queries, k)
        barrier()
        scans = []
        t0 = time.perf_counter()
        for _ in range(steps):
            D, I = index.search(queries, k)
            # HIP-event time of the scan kernel of this step (events sit on the launch stream;
            # reading them waits for the kernel, which the step's result needs anyway)
            scans.append(ix.last_timing()["scan_ms"])
        barrier()
        el = time.perf_counter() - t0
        ix.check()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, scans, D, I

    # ---- headline: the metric's workload, timed exactly K steps ----
    if args.storage == "bf16scan" and args.headline == "f32":
        ix.set_param("f32_scan", 1)
    elapsed, scan_ms, D, I = timed(Q, args.steps, args.warmup)
    ms_per_step = elapsed * 1e3 / args.steps
    stage(f"headline timed: {ms_per_step:.2f} ms per step")
    assert int(I.min()) >= 0 and int(I.max()) < n and bool((D[:, :-1] >= D[:, 1:]).all())
    geo, bf, tm = ix.last_geometry(), ix.last_bf16_stats(), ix.last_timing()
    is_bf16 = bf["candidates"] > 0

    # ---- extras (not the headline): the bf16 scan on the same index, and small batches ----
    extra = {}
    if args.storage == "bf16scan" and args.headline == "f32" and not args.no_extras and world == 1:
        ix.set_param("f32_scan", 0)
        el2, sc2, D2, I2 = timed(Q, args.steps, 1)
        bf2, tm2, geo2 = ix.last_bf16_stats(), ix.last_timing(), ix.last_geometry()
        same = bool(torch.equal(I, I2)) and bool(torch.equal(D.view(torch.int32), D2.view(torch.int32)))
        sc2a = sum(sc2) / len(sc2)
        extra["bf16scan"] = {
            "what": "same index, same queries: bf16 MFMA scan + exact fp32 rescoring; results compared bitwise with the headline run",
            "bit_identical_to_f32_scan": same, "queries_per_s": nq / (el2 / args.steps), "ms_per_step": el2 * 1e3 / args.steps,
            "scan_ms": sc2a, "merge_ms": round(tm2["merge_ms"], 3), "rescore_ms": round(bf2["rescore_ms"], 3),
            "candidates": bf2["candidates"], "repeated_queries": bf2["fallback_queries"],
            "workgroups": geo2["workgroups"], "splits": geo2["splits"],
            "roofline": {"bound": "mfma", "achieved": 2.0 * nq * (hi - lo) * d / (sc2a * 1e-3) / 1e12,
                         "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": 2.0 * nq * (hi - lo) * d / (sc2a * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                         "kernel": "scan_bf16_w16_kernel<1, true> (16x16x32 MFMA)"}}
        # the reference's default depth (--depth 1000, src/pretrain_finetune1.sh:218-228): one query part, 2k candidates
        k_deep = 1000
        old_k = k
        k = k_deep
        el4, sc4, D4, I4 = timed(Q, 2, 1)
        bf4 = ix.last_bf16_stats()
        ix.set_param("f32_scan", 1)
        el5, sc5, D5, I5 = timed(Q, 2, 1)
        ix.set_param("f32_scan", 0)
        k = old_k
        extra["depth_1000"] = {
            "what": "top-1000 on the same index (the reference's default --depth): bf16 scan keeping `candidates` per query + exact fp32 rescoring vs the fp32 scan",
            "bit_identical": bool(torch.equal(I4, I5)) and bool(torch.equal(D4.view(torch.int32), D5.view(torch.int32))),
            "bf16scan_queries_per_s": nq / (el4 / 2), "bf16scan_ms": el4 * 1e3 / 2, "bf16_scan_kernel_ms": sum(sc4) / len(sc4),
            "candidates": bf4["candidates"], "repeated_queries": bf4["fallback_queries"], "rescore_ms": round(bf4["rescore_ms"], 3),
            "f32_queries_per_s": nq / (el5 / 2), "f32_ms": el5 * 1e3 / 2, "f32_scan_kernel_ms": sum(sc5) / len(sc5)}
        small = []
        for bnq in (1, 16, 64):
            el3, sc3, _, _ = timed(Q[:bnq].contiguous(), 20, 3)
            sc3a = sum(sc3) / len(sc3)
            bytes_scan = (hi - lo) * d * 2
            small.append({"queries": bnq, "ms_per_search": el3 * 1e3 / 20, "queries_per_s": bnq / (el3 / 20),
                          "scan_ms": sc3a, "corpus_bytes_streamed": bytes_scan,
                          "roofline": {"bound": "hbm", "achieved": bytes_scan / (sc3a * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                                       "unit": "GB/s", "frac": bytes_scan / (sc3a * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                       "kernel": "scan_bf16_kernel<1>"}})
        extra["small_batch_bf16_copy"] = small
        # the same batches against the fp32 rows themselves (what a storage-0 index serves)
        ix.set_param("f32_scan", 1)
        small32 = []
        for bnq in (1, 16, 64):
            el3, sc3, _, _ = timed(Q[:bnq].contiguous(), 20, 3)
            sc3a = sum(sc3) / len(sc3)
            bytes_scan = (hi - lo) * d * 4
            # the binding roofline: streaming the fp32 rows once, or the fp32 matrix work of the batch — whichever
            # takes longer at peak (64 queries x 8.8M x 768: 5.5 ms of MFMA > 3.4 ms of HBM)
            flops = 2.0 * bnq * (hi - lo) * d
            t_hbm, t_mfma = bytes_scan / (PEAK_HBM_GBS * 1e9), flops / (PEAK_F32_MFMA_TFLOPS * 1e12)
            if t_mfma > t_hbm:
                roof = {"bound": "mfma", "achieved": flops / (sc3a * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": t_mfma / (sc3a * 1e-3)}
            else:
                roof = {"bound": "hbm", "achieved": bytes_scan / (sc3a * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": t_hbm / (sc3a * 1e-3)}
            roof["kernel"] = "scan_f32_small_kernel<%d>" % (1 if bnq <= 32 else 2)
            roof["hbm_time_ms_at_peak"], roof["mfma_time_ms_at_peak"] = t_hbm * 1e3, t_mfma * 1e3
            small32.append({"queries": bnq, "ms_per_search": el3 * 1e3 / 20, "queries_per_s": bnq / (el3 / 20),
                            "scan_ms": sc3a, "corpus_bytes_streamed": bytes_scan, "roofline": roof})
        ix.set_param("f32_scan", 0)
        extra["small_batch_f32_rows"] = small32

    full = (n, d) == (8_800_000, 768)
    if not args.no_extras and (full or args.extras_scale > 0):
        # the headline index goes first; then the measurements that build their own (smaller) indexes
        del index, ix
        torch.cuda.empty_cache()
        if world == 1:
            sc = args.extras_scale if args.extras_scale > 0 else 1.0
            extra["cfg4"] = run_cfg4(torch, dev, d, n=int(1_800_000 * sc), nq=max(64, int(8192 * min(1.0, 4 * sc))))
            extra["projected_8gpu"] = run_projected_8gpu(torch, dev, d, n, nq)
            extra["norm_spread"] = run_norm_spread(torch, dev, d, rows=int(1_000_000 * sc), nq=nq)
    if not args.no_extras and full:
        # BASELINE.json configs[4] (its own shard per rank, weak scaling), any number of ranks
        stage("cfg5 shard")
        extra["cfg5"] = run_cfg5(torch, dist, dev, local_rank, rank, world, d, 100, args.cfg5_rows)
        stage("cfg5 done")

    if rank == 0:
        scan_avg = sum(scan_ms) / len(scan_ms)
        flops_launch = 2.0 * nq * (hi - lo) * d                      # algorithmic flops of one scan launch
        achieved = flops_launch / (scan_avg * 1e-3) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if is_bf16 else PEAK_F32_MFMA_TFLOPS
        traffic, traffic_source = None, None
        prof = os.path.join(ROOT, "profiles", "latest_scan_f32_pmc.json")
        if not is_bf16 and (n, nq, world) == (8_800_000, 6980, 1) and os.path.exists(prof):
            # HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command
            # (FETCH_SIZE*1024*2 + WRITE_SIZE*1024, MI355X_MICROARCH.md §HBM); not collectable live. Only quoted
            # while the kernel's source is the one that was profiled.
            with open(prof) as f:
                pj = json.load(f)
            if pj.get("kernel_source_sha256") == kernel_source_hash():
                traffic = pj.get("traffic_bytes_per_launch")
                traffic_source = (f"profiles/latest_scan_f32_pmc.json (rocprofv3 --pmc passes of this command at git "
                                  f"{pj.get('git_head', '?')}, ipx_kernels.h sha256 {pj['kernel_source_sha256'][:12]}); not measured in this run")
            else:
                traffic_source = "none: profiles/latest_scan_f32_pmc.json was taken on an older ipx_kernels.h"
        out = {
            "metric": "QPS exact top-100 IP search, 8.8M×768 corpus @1/8 GPUs; % roofline",
            "value": nq / (ms_per_step * 1e-3),
            "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16 scan + f32 rescoring" if is_bf16 else "f32",
            "data": "synthetic: torch.randn on device, corpus seed 1234+block, query seed 4321",
            "config": {"workload": f"{n}x{d} fp32 corpus, {nq} queries, top-{k} (BASELINE.json configs[2] corpus, "
                                   f"headline depth 100)", "rows": n, "dim": d, "queries": nq, "k": k,
                       "parallelism": f"row-shard x{world}" if world > 1 else "single GPU",
                       "workgroups": geo["workgroups"], "splits": geo["splits"], "queue_cap": geo["queue_cap"],
                       "index_build_s": round(t_build, 2), "storage": args.storage, "headline_scan": "bf16" if is_bf16 else "f32",
                       "bf16_candidates": bf["candidates"], "repeated_queries": bf["fallback_queries"],
                       "rescore_ms": round(bf["rescore_ms"], 3), "merge_ms": round(tm["merge_ms"], 3)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "scan_bf16_w16_kernel<1>" if is_bf16 else "scan_f32_kernel", "kernel_ms": scan_avg,
                         "flops_per_launch": flops_launch},
        }
        if sample is not None and world == 1:
            Xs = np.concatenate(sample, axis=0)
            out["cpu_baseline"] = cpu_baseline(Xs, Q.cpu().numpy(), k, n)
        elif world == 1:
            out["cpu_baseline"] = None
        if extra:
            out["extra"] = extra
        emit(out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
