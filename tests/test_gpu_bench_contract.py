"""bench.py's output contract (one JSON line on stdout with the keys the driver reads), on a small corpus: the
headline mode with its roofline and cpu_baseline objects, and the cfg5 mode (BASELINE.json configs[4]) on a 1-rank
RCCL group. Runs bench.py as a child process, as the driver does."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                   # ONE line
    return json.loads(lines[0])


def test_headline_line_has_the_contract_keys():
    j = _bench("--rows", "400000", "--queries", "600", "--steps", "2", "--warmup", "1", "--cpu-sample-rows", "50000",
               "--extras-scale", "0.1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["unit"] == "queries/s" and j["dtype"] == "f32" and j["vs_baseline"] is None and "workload" in j["config"]
    assert abs(j["value"] - 600 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - 2.0 * 600 * 400000 * 768 / (r["kernel_ms"] * 1e-3) / 1e12) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None and "traffic_source" in r            # the PMC figure belongs to the full-size workload
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "queries/s" and c["value"] > 0 and c["cores"] >= 1 and "threads" in c["sample"]
    b = j["extra"]["bf16scan"]                                       # same index, bf16 scan: identical bits
    assert b["bit_identical_to_f32_scan"] is True and b["roofline"]["bound"] == "mfma"
    assert j["extra"]["depth_1000"]["bit_identical"] is True
    assert {x["queries"] for x in j["extra"]["small_batch_f32_rows"]} == {1, 16, 64}
    # BASELINE.json configs[3] (aspect queries fused on the device, bf16 scan + rescoring) with its roofline
    c4 = j["extra"]["cfg4"]
    assert c4["bit_identical_to_f32_scan"] is True and c4["prefused_ids_equal_frac"] > 0.99 and c4["prefused_max_rel_score_diff"] < 1e-4
    assert c4["roofline"]["bound"] == "mfma" and c4["roofline"]["peak"] == 2500.0 and c4["queries_per_s"] > 0
    # the 8-GPU figure is a labelled projection: measured shard step + measured merge + modelled all-gather
    p8 = j["extra"]["projected_8gpu"]
    assert p8["projection"] is True and p8["ranks"] == 8 and p8["rows_per_rank"] == 400000 // 8
    for key in ("top100", "top1000"):
        r8 = p8[key]
        tot = r8["f32_shard_search_ms"] + r8["modelled_allgather_ms"] + r8["merge_ms"]
        assert abs(r8["f32_projected_ms_per_step"] - tot) <= 1e-6 * tot
        assert abs(r8["modelled_allgather_ms"] - (r8["block_bytes_per_rank"] / 153e9 + 20e-6) * 1e3) < 1e-9
    ns = j["extra"]["norm_spread"]
    for kind in ("gauss", "lognormal", "l2norm", "dups"):
        assert ns[kind]["bit_identical"] is True and ns[kind]["candidates"] > 0, kind


def test_cfg5_line():
    j = _bench("--workload", "cfg5", "--cfg5-rows", "600000", "--steps", "3", "--warmup", "1")
    assert j["scaling"] == "weak" and j["n_gpus"] == 1 and j["unit"] == "queries/s" and j["roofline"]["bound"] == "hbm"
    rows = j["extra"]["cfg5"]["batches"]
    assert [r["queries"] for r in rows] == [1, 2, 4, 8, 16, 32, 64]
    for r in rows:
        assert r["ms_per_search"] > 0 and r["scan_ms"] > 0 and r["ms_per_search"] >= r["scan_ms"] * 0.9
        assert abs(r["roofline"]["achieved"] - 600000 * 768 * 2 / (r["scan_ms"] * 1e-3) / 1e9) <= 1e-6 * r["roofline"]["achieved"]
    assert abs(j["value"] - rows[-1]["queries_per_s"]) <= 1e-9 * j["value"]


def test_bench_under_torch_distributed_run_with_one_rank():
    """The driver launches N > 1 ranks with torch.distributed.run; the same launcher with ONE rank must work too (the
    1-rank group then comes from the launcher's store: a tcp:// rendezvous of bench.py's own used to wait forever for a
    server that nobody starts — found late in round 3)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--rows", "300000", "--queries", "300", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["value"] > 0 and j["roofline"]["bound"] == "mfma"
