#!/bin/bash
# round 3, call 13: kernel stats of the SDXL-shape UNet step at 4x128x128, batch 12 (final kernels)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r3m13; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/unet -o unet --output-format csv -- python bench.py --model SDXL-UNet --latent 128 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary > $out/unet.log 2>&1 &&
python tools/trace_shapes.py $out/unet/unet_kernel_trace.csv > $out/unet_shapes.txt 2>&1
rm -f $out/unet/unet_kernel_trace.csv
echo rc=$?
