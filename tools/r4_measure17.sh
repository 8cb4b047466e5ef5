#!/bin/bash
# in-situ per-shape timings of the UNet step's GEMM launches + the same shapes alone
cd "$(dirname "$0")/.."
python tools/probe_unet_wgrad_insitu.py > gpurun_out/r4_m17_insitu.txt 2>&1 &&
python tools/probe_wgrad_unet.py > gpurun_out/r4_m17_alone.txt 2>&1
head -60 gpurun_out/r4_m17_insitu.txt | cut -c1-150; cat gpurun_out/r4_m17_alone.txt
