"""One rank of the CLI's multi-process mode with the CPU oracle injected as the per-rank engine (test helper of
tests/test_sharded_gloo.py::test_one_command_spawns_the_ranks; started by _spawn_ranks, never imported by product code)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mural_amd.faiss_retriever import __main__ as drv          # noqa: E402
from test_sharded_gloo import NumpyOracleEngine                 # noqa: E402

if __name__ == "__main__":
    assert drv._is_rank_of_a_launch()
    drv._main_sharded(drv._parse_args(sys.argv[1:]), engine_factory=NumpyOracleEngine)
