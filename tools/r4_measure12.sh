#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8n" > gpurun_out/r4_m12_tests.txt 2>&1 || { tail -40 gpurun_out/r4_m12_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m12_tests.txt
for rep in 1 2; do
for w in 0 192 128; do
  echo "== p8n $w (rep $rep)"
  env $([ $w = 0 ] && echo UWU_GEMM_P8N=0 || echo UWU_P8N_ROWS=$w) UWU_BENCH_ONLY=qkv_fwd,proj_fwd,fc2_fwd,qkv_dgrad,proj_dgrad,fc1_dgrad timeout -k 10 200 python tools/bench_gemm_shapes.py 384 768 2>&1 | grep -v "amdgpu.ids" || exit 1
done; done > gpurun_out/r4_m12_p8n.txt
cat gpurun_out/r4_m12_p8n.txt
