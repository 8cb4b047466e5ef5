#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r3m3; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > $o/attn_test.log 2>&1; rc=$?; echo "attn tests rc=$rc"; tail -4 $o/attn_test.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/bench_attn.py 768 > $o/bench_attn.txt 2>&1; cat $o/bench_attn.txt
for abl in 1 2 3 4; do
  UWU_P256_ABL=$abl timeout -k 10 100 python tools/bench_attn.py 768 2>&1 | grep -E "bwd p256|p256 wave" | sed "s/^/ABL=$abl /" | tee -a $o/abl.txt
done
