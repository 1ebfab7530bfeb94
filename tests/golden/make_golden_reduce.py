"""Generates the reduce_results fixtures. Run in the BUILD container only:

    cd /root/repo && python tests/golden/make_golden_reduce.py

Inputs are synthetic ranking files (what `python -m tevatron.faiss_retriever --save_text` writes:
float32 scores printed through Python floats); the expected outputs are what the reference's own
/root/reference/scripts/reduce_results.py (plain Python, no dependencies) writes for them. Every
file lists the queries in the same order and no score ties cross files, so the expectation does not
depend on os.listdir order (ties inside one file exercise the stable sort). Only data is written.
"""
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scripts/reduce_results.py"


def write_case(name, n_files, nq, per_file, depth, f32_scores, ragged):
    rng = np.random.default_rng(len(name) * 1000 + n_files)
    indir = os.path.join(HERE, f"{name}_in")
    shutil.rmtree(indir, ignore_errors=True)
    os.makedirs(indir)
    qids = [f"q{7 * i + 3}" if i % 2 else str(1000 + i) for i in range(nq)]
    for s in range(n_files):
        with open(os.path.join(indir, f"{s:02d}"), "w") as f:
            for qi, qid in enumerate(qids):
                m = per_file - (qi + s) % 3 if ragged else per_file
                sc = rng.standard_normal(m) * 10 + (1000.0 if qi == 1 else 0.0)
                sc = sc.astype(np.float32) if f32_scores else np.round(sc.astype(np.float64), 7)
                sc = np.sort(sc)[::-1].copy()
                if m > 4:
                    sc[3] = sc[2]                       # a tie inside the file: stable order decides
                sc = sc + (s * 1e-3 if not f32_scores else 0)   # keep ties from crossing files
                for j, v in enumerate(sc):
                    sep = "\t" if (j + qi) % 5 else "  "         # any whitespace splits
                    f.write(f"{qid}{sep}B{s}X{qi}_{j}\t{float(v)}\n")
    out = os.path.join(HERE, f"{name}_expected.tsv")
    subprocess.check_call([sys.executable, REF, "--results_dir", indir, "--output", out, "--depth", str(depth)],
                          stdout=subprocess.DEVNULL)
    print(name, "->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    write_case("R1", n_files=3, nq=5, per_file=12, depth=10, f32_scores=True, ragged=False)
    write_case("R2", n_files=4, nq=4, per_file=9, depth=100, f32_scores=True, ragged=True)     # depth > hits
    write_case("R3", n_files=2, nq=3, per_file=6, depth=5, f32_scores=False, ragged=True)      # not float32: host path
