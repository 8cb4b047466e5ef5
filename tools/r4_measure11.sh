#!/bin/bash
# DiT-S/2 (K = 384) on the 8-phase kernel?  headline shapes at B = 768
mkdir -p gpurun_out
for k in 512 384; do
  echo "== UWU_P8_KMIN=$k"
  UWU_P8_KMIN=$k UWU_BENCH_ONLY=qkv_fwd,fc1_fwd,fc1_dgrad,qkv_dgrad,fc2_fwd timeout -k 10 200 python tools/bench_gemm_shapes.py 384 768 2>&1 | grep -v "amdgpu.ids\|block total" || exit 1
done > gpurun_out/r4_m11_s2.txt
cat gpurun_out/r4_m11_s2.txt
