#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r3m5; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "ln" > $o/ln_test.log 2>&1; rc=$?; echo "ln tests rc=$rc"; tail -4 $o/ln_test.log
for f in 1 0 1 0; do echo "UWU_LN_ROW16=$f"; UWU_LN_ROW16=$f timeout -k 10 100 python tools/bench_kernels.py 768 ln 2>&1 | grep ln_; done
for f in 1 0; do echo "B=256 UWU_LN_ROW16=$f"; UWU_LN_ROW16=$f timeout -k 10 100 python tools/bench_kernels.py 256 ln 2>&1 | grep ln_fwd; done
