#!/bin/bash
# round 4, call 1: p8 kernel correctness + per-shape A/B against the round-3 kernels and the vendor GEMM
set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8" > gpurun_out/r4_m1_tests.txt 2>&1 || { tail -30 gpurun_out/r4_m1_tests.txt; exit 1; }
tail -3 gpurun_out/r4_m1_tests.txt
for D in 768 1152; do
  B=256; [ $D = 1152 ] && B=192
  UWU_GEMM_P8=0 timeout -k 10 300 python tools/bench_gemm_shapes.py $D $B vendor > gpurun_out/r4_m1_shapes_D${D}_old.txt 2>&1 &&
  timeout -k 10 300 python tools/bench_gemm_shapes.py $D $B > gpurun_out/r4_m1_shapes_D${D}_p8.txt 2>&1 || exit 1
done
tail -20 gpurun_out/r4_m1_shapes_D768_old.txt gpurun_out/r4_m1_shapes_D768_p8.txt
