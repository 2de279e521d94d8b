#!/bin/bash
# Collects everything under profiles/ that comes from the GPU box, from the CURRENT build.  Run on the box from the repository
# root (e.g. `gpurun --timeout 1200 -- 'bash profiles/collect.sh r02'`), then condense here with
#   python profiles/summarize_pmc.py r02 gpurun_out/prof_r02 gpurun_out/pmc_r02_fetch gpurun_out/pmc_r02_write gpurun_out/pmc_r02_sq
# and copy gpurun_out/prof_<tag>_e2e/*/*_kernel_stats.csv, prof_<tag>.json, prof_<tag>_e2e.json (see profiles/summarize_pmc.py).
# Counters run in their own passes, one TCC counter per pass, with no tracing beside them (MI355X_MICROARCH.md, HBM section).
set -e -o pipefail
tag=${1:-r02}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-precisions"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precisions > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- $B > /dev/null 2> gpurun_out/pmc_${tag}_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- $B > /dev/null 2> gpurun_out/pmc_${tag}_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_sq -- $B > /dev/null 2> gpurun_out/pmc_${tag}_sq.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_e2e -- python3 bench.py --config e2e --steps 10 --warmup 3 > gpurun_out/prof_${tag}_e2e.json 2> gpurun_out/prof_${tag}_e2e.err
python3 bench.py > gpurun_out/bench_$tag.log 2> gpurun_out/bench_$tag.err
python3 bench.py --config fwd > gpurun_out/fwd_$tag.log 2>> gpurun_out/bench_$tag.err
python3 bench.py --config e2e > gpurun_out/e2e_$tag.log 2>> gpurun_out/bench_$tag.err
echo collected $tag
