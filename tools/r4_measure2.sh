#!/bin/bash
# round 4, call 2: ablations of the p8 kernel (timing-only builds) on the plain forwards of DiT-B/2
mkdir -p gpurun_out
for abl in 0 1 2 3 4; do
  echo "== UWU_P8_ABL=$abl"
  UWU_P8_ABL=$abl UWU_BENCH_ONLY=qkv_fwd,proj_fwd,fc1_fwd,fc2_fwd timeout -k 10 200 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v amdgpu.ids || exit 1
done > gpurun_out/r4_m2_abl.txt
cat gpurun_out/r4_m2_abl.txt
