#!/bin/bash
# fp8 weight amax passes: two workgroups per CU against the uncapped grid (UWU_FP8_AMAX_WIDE=1), DiT-XL/2 fp8 step, alternating
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for wide in 0 1; do
  echo "== UWU_FP8_AMAX_WIDE=$wide (rep $rep)"
  UWU_FP8_AMAX_WIDE=$wide timeout -k 10 300 python bench.py --model DiT-XL/2 --batch 192 --dtype fp8 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'final loss', d.get('final_loss'))
" || exit 1
done; done > gpurun_out/r4_m23_amax.txt 2>&1
cat gpurun_out/r4_m23_amax.txt
