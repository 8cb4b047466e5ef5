#!/bin/bash
# round 3, call 12: SDXL-shape UNet, the few-row fp32 Linears on the matrix-vector kernels: tests + A/B at 4x128x128, batch 12
cd "$(dirname "$0")/.."
out=gpurun_out/r3m12; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_unet_gpu.py -q -x -m gpu > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
for e in 1 0 1 0; do
  UWU_UNET_SKINNY=$e python bench.py --model SDXL-UNet --latent 128 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SKINNY=$e', d['value'], d['ms_per_step'], d['final_loss'])" | tee -a $out/ab.txt
done
