#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the bench command + FETCH_SIZE / WRITE_SIZE PMC passes
# (separate runs, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/prof_$1/.
# usage: bash tools/profile_round.sh r01
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-r03}
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o $tag --output-format csv -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary > $out/trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-secondary > $out/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE -d $out/write -o write --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-secondary > $out/write.log 2>&1 &&
python tools/summarize_pmc.py $out/fetch $out/write $out/pmc_gemm_traffic.json && cp $out/trace/${tag}_kernel_stats.csv $out/kernel_stats.csv && rm -f $out/trace/*_kernel_trace.csv
