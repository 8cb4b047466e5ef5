#!/bin/bash
# timing-only build (-DUWU_TEST_MUL_AUX: dgelu_tanh_f4(x) = x): what would a stashed GELU derivative buy the dGELU kernels?
mkdir -p gpurun_out
UWU_BENCH_ONLY=fc2_dgrad timeout -k 10 200 python tools/bench_kernels.py 768 gemm 2>&1 | grep -v amdgpu > gpurun_out/r4_m14_mulaux.txt
UWU_BENCH_ONLY=fc2_dgrad timeout -k 10 200 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v amdgpu >> gpurun_out/r4_m14_mulaux.txt
cat gpurun_out/r4_m14_mulaux.txt
