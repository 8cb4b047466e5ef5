#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8" > gpurun_out/r4_m7_tests.txt 2>&1 || { tail -30 gpurun_out/r4_m7_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m7_tests.txt
timeout -k 10 300 python tools/bench_gemm_shapes.py 768 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_m7_shapes_D768.txt || exit 1
cat gpurun_out/r4_m7_shapes_D768.txt
timeout -k 10 300 python tools/bench_gemm_shapes.py 1152 192 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_m7_shapes_D1152.txt || exit 1
cat gpurun_out/r4_m7_shapes_D1152.txt
timeout -k 10 100 python tools/probe_p8_stamps.py 65536 3072 768 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_m7_stamps_fc1.txt; cat gpurun_out/r4_m7_stamps_fc1.txt
