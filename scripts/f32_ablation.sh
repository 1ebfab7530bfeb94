#!/bin/bash
# A/B builds of the fp32 headline scan (ipx_kernels.h): ablations (IPX_F32_ABL=N, results wrong, timing only: what each
# part of the stage costs) and optimisation switches (IPX_F32_OPT=N, same bits).
#   bash scripts/f32_ablation.sh build "abl2 abl32 opt0 opt1"     # here (CPU): mural_amd/libipx_<name>.so
#   bash scripts/f32_ablation.sh build trace                       # the counting build for scripts/trace_f32_hits.py
#   gpurun -- 'bash scripts/f32_ablation.sh run "abl2 abl32 opt0 opt1"'
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=$1
LIST=${2:-"abl2 abl32"}
if [ "$MODE" = build ]; then
    for n in $LIST; do
        case $n in
            abl*) D="-DIPX_F32_ABL=${n#abl}" ;;
            opt*) D="-DIPX_F32_OPT=${n#opt}" ;;
            trace) D="-DIPX_F32_TRACE_HITS" ;;          # libipx_trace.so: what scripts/trace_f32_hits.py loads
        esac
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread $D \
            $R/mural_amd/csrc/ipx.hip $R/mural_amd/csrc/ranking_writer.cpp -o $R/mural_amd/libipx_$n.so &
    done
    wait
    ls -la $R/mural_amd/libipx_*.so
else
    cd $R
    timeout -k 10 120 python -u scripts/f32_headline.py ${K:-100} 3
    for n in $LIST; do
        IPX_LIB=libipx_$n.so timeout -k 10 120 python -u scripts/f32_headline.py ${K:-100} 3 || exit 1
    done
    timeout -k 10 120 python -u scripts/f32_headline.py ${K:-100} 3
fi
