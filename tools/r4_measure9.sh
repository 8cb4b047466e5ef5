#!/bin/bash
# whole-step A/B of the p8 kernel on the secondary configs (same box, alternating)
mkdir -p gpurun_out
for rep in 1 2; do
for p8 in 1 0; do
  for cfg in "DiT-B/2 256 bf16" "DiT-XL/2 192 bf16" "DiT-XL/2 192 fp8"; do
    set -- $cfg
    echo "== UWU_GEMM_P8=$p8 $cfg (rep $rep)"
    env $([ $p8 = 0 ] && echo UWU_GEMM_P8=0 || echo UWU_DUMMY=1) timeout -k 10 300 python bench.py --model $1 --batch $2 --dtype $3 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'gemm family frac', r.get('frac'), 'gemm ms', r.get('gemm_ms_per_step'))
        for k in r.get('kernels', []): print('   ', k.get('name'), k.get('launches'), k.get('avg_us'), k.get('tflops'))
" || exit 1
  done
done; done > gpurun_out/r4_m9_step_ab.txt 2>&1
cat gpurun_out/r4_m9_step_ab.txt
