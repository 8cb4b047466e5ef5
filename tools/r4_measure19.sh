#!/bin/bash
# kernel stats of the DiT-XL/2 fp8 and bf16 steps (batch 192), grouped by (kernel, grid)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r4m19; rm -rf $out; mkdir -p $out
for dt in fp8 bf16; do
rocprofv3 --kernel-trace --stats -d $out/xl_$dt -o xl --output-format csv -- python bench.py --model DiT-XL/2 --batch 192 --dtype $dt --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary > $out/xl_$dt.log 2>&1 &&
python tools/trace_shapes.py $out/xl_$dt/xl_kernel_trace.csv > $out/xl_${dt}_shapes.txt 2>&1
rm -f $out/xl_$dt/xl_kernel_trace.csv
done
head -40 $out/xl_fp8_shapes.txt | cut -c1-180
