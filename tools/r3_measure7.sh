#!/bin/bash
# round 3, call 7: kernel stats of the DiT-XL/2 fp8 and DiT-B/2 steps (where does the non-GEMM time go?) + the round profile
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r3m7; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/xl -o xl --output-format csv -- python bench.py --model DiT-XL/2 --batch 192 --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary > $out/xl.log 2>&1 &&
rm -f $out/xl/*_kernel_trace.csv &&
rocprofv3 --kernel-trace --stats -d $out/b2 -o b2 --output-format csv -- python bench.py --model DiT-B/2 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary > $out/b2.log 2>&1 &&
rm -f $out/b2/*_kernel_trace.csv &&
bash tools/profile_round.sh r03 &&
python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "rc=$?"
