#!/bin/bash
mkdir -p gpurun_out
for nt in 0 4; do
  echo "== UWU_GEMM_NT_C=$nt"
  UWU_GEMM_NT_C=$nt UWU_BENCH_ONLY=qkv_fwd,proj_fwd,fc1_fwd,fc2_fwd timeout -k 10 200 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v "amdgpu.ids\|block total" || exit 1
done > gpurun_out/r4_m6_nt.txt
cat gpurun_out/r4_m6_nt.txt
