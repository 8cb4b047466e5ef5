#!/bin/bash
# dGELU input gradient on the 8-phase kernel (UWU_P8_DGELU=1) against the 256 x 256 two-stage kernel, whole steps, alternating
cd "$(dirname "$0")/.."
for rep in 1 2; do
for on in 1 0; do
  for cfg in "DiT-XL/2 192 bf16" "DiT-B/2 256 bf16"; do
    set -- $cfg
    echo "== UWU_P8_DGELU=$on $cfg (rep $rep)"
    UWU_P8_DGELU=$on timeout -k 10 300 python bench.py --model $1 --batch $2 --dtype $3 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'gemm family frac', r.get('frac'))
" || exit 1
  done
done; done > gpurun_out/r4_m21_dgelu.txt 2>&1
cat gpurun_out/r4_m21_dgelu.txt
