#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8" > gpurun_out/r4_m5_tests.txt 2>&1 || { tail -30 gpurun_out/r4_m5_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m5_tests.txt
for grid in 256 128; do
for abl in 0 1; do
  echo "== UWU_P8_GRID=$grid UWU_P8_ABL=$abl"
  UWU_P8_GRID=$grid UWU_P8_ABL=$abl UWU_BENCH_ONLY=qkv_fwd,proj_fwd,fc1_fwd,fc2_fwd timeout -k 10 200 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v "amdgpu.ids\|block total" || exit 1
done; done > gpurun_out/r4_m5_grid.txt
cat gpurun_out/r4_m5_grid.txt
