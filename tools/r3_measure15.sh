#!/bin/bash
# round 3, call 15: kernel stats of the SDXL-shape UNet step WITHOUT the side stream (UWU_UNET_FORK=0): per-kernel durations that are
# not inflated by the weight gradients running beside the main chain
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export UWU_UNET_FORK=0
out=gpurun_out/r3m15; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/unet -o unet --output-format csv -- python bench.py --model SDXL-UNet --latent 128 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary > $out/unet.log 2>&1 &&
python tools/trace_shapes.py $out/unet/unet_kernel_trace.csv > $out/unet_shapes.txt 2>&1
rm -f $out/unet/unet_kernel_trace.csv
grep -o '"value": [0-9.]*, "unit": "images/s"' $out/unet.log | head -1
