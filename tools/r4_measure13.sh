#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k "p8n" > gpurun_out/r4_m13_tests.txt 2>&1 || { tail -40 gpurun_out/r4_m13_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m13_tests.txt
for rep in 1 2 3; do
for w in 0 x; do
  echo "== UWU_GEMM_P8N=$w (rep $rep)"
  env $([ $w = 0 ] && echo UWU_GEMM_P8N=0 || echo UWU_DUMMY=1) timeout -k 10 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'gemm family frac', r.get('frac'), 'gemm ms', r.get('gemm_ms_per_step'))
        for k in r.get('kernels', []): print('   ', k)
" || exit 1
done; done > gpurun_out/r4_m13_step_ab.txt 2>&1
cat gpurun_out/r4_m13_step_ab.txt | cut -c1-250
