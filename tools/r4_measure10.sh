#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8w" > gpurun_out/r4_m10_tests.txt 2>&1 || { tail -40 gpurun_out/r4_m10_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m10_tests.txt
for w in 0 x; do
  echo "== UWU_GEMM_P8W=$w"
  env $([ $w = 0 ] && echo UWU_GEMM_P8W=0 || echo UWU_DUMMY=1) UWU_BENCH_ONLY=qkv_wgrad,proj_wgrad,fc1_wgrad,fc2_wgrad timeout -k 10 200 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v "amdgpu.ids" || exit 1
  env $([ $w = 0 ] && echo UWU_GEMM_P8W=0 || echo UWU_DUMMY=1) UWU_BENCH_ONLY=qkv_wgrad,proj_wgrad,fc1_wgrad,fc2_wgrad timeout -k 10 200 python tools/bench_gemm_shapes.py 1152 192 2>&1 | grep -v "amdgpu.ids" || exit 1
done > gpurun_out/r4_m10_wgrad.txt
cat gpurun_out/r4_m10_wgrad.txt
