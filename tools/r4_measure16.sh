#!/bin/bash
# kernel stats of the SDXL-shape UNet step WITHOUT the side stream (UWU_UNET_FORK=0), grouped by (kernel, grid)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export UWU_UNET_FORK=0
out=gpurun_out/r4m16; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/unet -o unet --output-format csv -- python bench.py --model SDXL-UNet --latent 128 --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary > $out/unet.log 2>&1 &&
python tools/trace_shapes.py $out/unet/unet_kernel_trace.csv > $out/unet_shapes.txt 2>&1
rm -f $out/unet/unet_kernel_trace.csv
grep -o '"value": [0-9.]*, "unit": "images/s"' $out/unet.log | head -1
head -45 $out/unet_shapes.txt | cut -c1-170
