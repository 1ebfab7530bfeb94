#!/bin/bash
# LDS-side counters of the large-batch bf16 scan variants (one PMC pass): bank-conflict cycles, LDS-array cycles,
# MFMA busy. Run on the GPU box: gpurun -- 'bash scripts/pmc_lds.sh "s32;ring3"'
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
export VARIANTS=${1:-"s32"}
export STORAGE=${STORAGE:-bf16scan}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_lds
rm -rf $OUT && mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/scripts/bf16_variants.py 8800000 6980 1 > $OUT/sq.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/sq/*/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "scan_bf16" in r["Kernel_Name"]]
by = {}
for r in rows:
    by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"][:60]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(by):
    print(k[0], k[1], {c: "%.4g" % v for c, v in by[k].items()})
PY
