#!/bin/bash
# Collects everything under profiles/ that comes from the GPU box, from the CURRENT build.  Run on the box from the repository
# root, e.g.   gpurun --timeout 1200 -- 'bash profiles/collect.sh r04'
# then condense here with   bash profiles/collect.sh --summarize r04   (copies the small per-kernel tables into profiles/).
#
# Rules this script encodes (MI355X_MICROARCH.md, rocprofv3 PMC slots; the round-1 and round-2 aborts "error code 38: Request
# exceeds the capabilities of the hardware to collect" came from several TCC-derived counters in one --pmc pass):
#   * every TCC counter gets a pass of its OWN (FETCH_SIZE alone takes 3 of the 4 TCC slots); SQ / GRBM counters may share one;
#   * a counter pass carries no tracing flag; the kernel-trace pass carries no counters;
#   * the program follows `--` directly (no env / bash -c / launcher in between);
#   * a counter the installed rocprofv3 does not know is skipped (its pass fails fast with a message), never retried.
set -o pipefail
if [ "$1" = "--summarize" ]; then
    tag=${2:-r04}
    here=$(cd "$(dirname "$0")" && pwd)
    for cfg in train e2e fwd; do
        sfx=$([ $cfg = train ] && echo "" || echo "_$cfg")
        dirs=$(ls -d gpurun_out/pmc_${tag}_${cfg}_* 2>/dev/null | tr '\n' ' ')
        [ -d gpurun_out/prof_${tag}_${cfg} ] && python3 "$here/summarize_pmc.py" ${tag}${sfx} gpurun_out/prof_${tag}_${cfg} $dirs
        [ -f gpurun_out/prof_${tag}_${cfg}.json ] && cp gpurun_out/prof_${tag}_${cfg}.json "$here/${tag}${sfx}_bench_under_rocprof.json"
    done
    exit 0
fi
tag=${1:-r04}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p gpurun_out
# (round 3 also collected TCC_EA0_RDREQ* / WRREQ* / BUBBLE: RDREQ_DRAM equals RDREQ on every kernel -- it does not separate Infinity-Cache
# hits from HBM reads -- so they were dropped; profiles/r03_pmc_summary.json keeps them)
TCC_COUNTERS="FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
SQ_PASS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run_cfg() {     # $1 = name, rest = bench.py arguments of the short (counter) run; the traced run uses --steps 10 --warmup 3
    cfg=$1; shift
    echo "== $cfg: kernel trace"
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_${cfg} -- python3 bench.py "$@" --steps 10 --warmup 3 \
        > gpurun_out/prof_${tag}_${cfg}.json 2> gpurun_out/prof_${tag}_${cfg}.err || echo "   trace pass failed (see gpurun_out/prof_${tag}_${cfg}.err)"
    for c in $TCC_COUNTERS; do
        echo "== $cfg: counter $c"
        rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_${cfg}_$c -- python3 bench.py "$@" --steps 3 --warmup 1 \
            > /dev/null 2> gpurun_out/pmc_${tag}_${cfg}_$c.err || echo "   skipped $c (not collectable here: gpurun_out/pmc_${tag}_${cfg}_$c.err)"
    done
    echo "== $cfg: SQ / GRBM pass"
    rocprofv3 --pmc $SQ_PASS --output-format csv -d gpurun_out/pmc_${tag}_${cfg}_sq -- python3 bench.py "$@" --steps 3 --warmup 1 \
        > /dev/null 2> gpurun_out/pmc_${tag}_${cfg}_sq.err || echo "   SQ pass failed"
}
run_cfg train --no-cpu-baseline --no-other-precisions --no-other-configs
run_cfg e2e --config e2e
echo "== fwd: kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_fwd -- python3 bench.py --config fwd --steps 10 --warmup 3 \
    > gpurun_out/prof_${tag}_fwd.json 2> gpurun_out/prof_${tag}_fwd.err || echo "   fwd trace failed"
echo "== the default line (what the driver records)"
python3 bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo collected $tag
