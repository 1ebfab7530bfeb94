#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every large-batch bf16 scan launch of scripts/bf16_variants.py (one PMC pass each),
# to compare the memory-side traffic of mapping / ring variants. Run on the GPU box.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_variants
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/scripts/bf16_variants.py 8800000 6980 1 > $OUT/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $R/scripts/bf16_variants.py 8800000 6980 1 > $OUT/sq.log 2>&1
echo "sq pass done"
python3 - <<PY
import csv, glob
for t in ("fetch", "sq"):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % t)[0]
    rows = [r for r in csv.DictReader(open(f)) if "scan_bf16" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:70]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(by):
        print(t, k[0], k[1], {c: "%.4g" % v for c, v in by[k].items()})
PY
