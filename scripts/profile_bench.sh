#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box via gpurun). Each pass is its own run:
# kernel-trace/stats first, then one PMC pass per counter group (FETCH_SIZE and WRITE_SIZE do
# not fit one pass). Outputs land under gpurun_out/prof_<tag>/.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
ARGS=${2:-"--steps 2 --warmup 1 --no-cpu-baseline"}
EXTRA=${3:-""}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS $EXTRA > $OUT/stats.log 2>&1
echo "stats pass done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $EXTRA > $OUT/pmc_fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $EXTRA > $OUT/pmc_write.log 2>&1
echo "write pass done"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $EXTRA > $OUT/pmc_sq.log 2>&1
echo "sq pass done"
find $OUT -name "*.csv" | head -40
