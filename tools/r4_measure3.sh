#!/bin/bash
# round 4, call 3: persistent p8 kernel -- tests, shapes, ablations
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8" > gpurun_out/r4_m3_tests.txt 2>&1 || { tail -30 gpurun_out/r4_m3_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m3_tests.txt
timeout -k 10 300 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_m3_shapes_D768.txt || exit 1
cat gpurun_out/r4_m3_shapes_D768.txt
timeout -k 10 300 python tools/bench_gemm_shapes.py 1152 192 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_m3_shapes_D1152.txt || exit 1
cat gpurun_out/r4_m3_shapes_D1152.txt
for abl in 1 2 4 5 6; do
  echo "== UWU_P8_ABL=$abl"
  UWU_P8_ABL=$abl UWU_BENCH_ONLY=qkv_fwd,proj_fwd,fc1_fwd,fc2_fwd timeout -k 10 200 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v "amdgpu.ids\|gelu\|block total" || exit 1
done > gpurun_out/r4_m3_abl.txt
cat gpurun_out/r4_m3_abl.txt
