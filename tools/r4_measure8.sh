#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
for c in 1 0; do
  echo "== UWU_P8_CONT=$c (rep $rep)"
  UWU_P8_CONT=$c UWU_BENCH_ONLY=qkv_fwd,proj_fwd,fc1_fwd,fc2_fwd,qkv_dgrad,fc1_dgrad,proj_dgrad timeout -k 10 200 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v "amdgpu.ids\|block total" || exit 1
done; done > gpurun_out/r4_m8_cont.txt
cat gpurun_out/r4_m8_cont.txt
