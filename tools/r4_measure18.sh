#!/bin/bash
# whole-step A/B of the head-dim-72 persistent attention kernels on DiT-XL/2 (same box, alternating) + the new tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "head_dim_72 or p256" > gpurun_out/r4_m18_tests.txt 2>&1 || { tail -20 gpurun_out/r4_m18_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m18_tests.txt
for rep in 1 2; do
for on in 1 0; do
  for cfg in "DiT-XL/2 192 bf16" "DiT-XL/2 192 fp8"; do
    set -- $cfg
    echo "== d72 persistent attention=$on $cfg (rep $rep)"
    UWU_ATTN_P256_D72=$on UWU_ATTN_P256F_D72=$on timeout -k 10 300 python bench.py --model $1 --batch $2 --dtype $3 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'whole-step mfma frac', d.get('mfma_frac_whole_step'))
" || exit 1
  done
done; done > gpurun_out/r4_m18_step_ab.txt 2>&1
cat gpurun_out/r4_m18_step_ab.txt
