#!/bin/bash
# fp8 GEMMs of a DiT-XL/2 block: the 8-phase kernel against the two-stage kernel, then the whole fp8 step, alternating
cd "$(dirname "$0")/.."
for on in 1 0 1 0; do
  echo "== UWU_GEMM_P8F=$on UWU_GEMM_P8F_PART=$on"
  UWU_GEMM_P8F=$on UWU_GEMM_P8F_PART=$on python tools/bench_gemm_fp8_shapes.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r4_m24_p8f_shapes.txt 2>&1
cat gpurun_out/r4_m24_p8f_shapes.txt
for rep in 1 2; do
for on in 1 0; do
  echo "== fp8 8-phase kernels $on (rep $rep)"
  env $([ $on = 0 ] && echo UWU_GEMM_P8F=0 UWU_GEMM_P8F_PART=0 || echo UWU_DUMMY=1) timeout -k 10 300 python bench.py --model DiT-XL/2 --batch 192 --dtype fp8 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'final loss', d.get('final_loss'))
" || exit 1
done; done > gpurun_out/r4_m24_step.txt 2>&1
cat gpurun_out/r4_m24_step.txt
