#!/bin/bash
# round 4, call 4: is the epilogue's cost the chip-wide store burst?  fewer CUs -> smaller bursts
mkdir -p gpurun_out
for grid in 256 128 64; do
for abl in 0 1; do
  echo "== UWU_P8_GRID=$grid UWU_P8_ABL=$abl"
  UWU_P8_GRID=$grid UWU_P8_ABL=$abl UWU_BENCH_ONLY=qkv_fwd,fc1_fwd,fc2_fwd timeout -k 10 200 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v "amdgpu.ids\|gelu\|block total" || exit 1
done; done > gpurun_out/r4_m4_grid.txt
cat gpurun_out/r4_m4_grid.txt
