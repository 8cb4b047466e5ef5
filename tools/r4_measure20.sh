#!/bin/bash
# static wave priority in the persistent attention kernels (UWU_P256_PRIO = 0 / 1 / 2), d = 64 at 4608 heads and d = 72 at 3072
cd "$(dirname "$0")/.."
for p in 0 1 2 0 1 2; do
  echo "== UWU_P256_PRIO=$p"
  UWU_P256_PRIO=$p python tools/bench_attn.py 768 2>&1 | grep -v amdgpu.ids | grep p256
  UWU_P256_PRIO=$p python tools/bench_attn72.py 192 2>&1 | grep -v amdgpu.ids | grep "p256 "
done > gpurun_out/r4_m20_prio.txt 2>&1
cat gpurun_out/r4_m20_prio.txt
