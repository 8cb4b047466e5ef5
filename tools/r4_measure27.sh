#!/bin/bash
# LayerNorm backward, wide rows: column partial sums in LDS (ds_add_f32) against the register form (UWU_LN_BWD_LACC=0); tests first
cd "$(dirname "$0")/.."
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "ln or norm or modulate" > gpurun_out/r4_m27_tests.txt 2>&1 || { tail -30 gpurun_out/r4_m27_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m27_tests.txt
for rep in 1 2; do
for cfg in "1152 192" "768 256"; do
  set -- $cfg
  for on in 1 0; do
    echo "== D=$1 B=$2 UWU_LN_BWD_LACC=$on"
    UWU_BENCH_D=$1 UWU_LN_BWD_LACC=$on python tools/bench_kernels.py $2 ln 2>&1 | grep "ln_bwd"
  done
done; done > gpurun_out/r4_m27_ln.txt 2>&1
cat gpurun_out/r4_m27_ln.txt
