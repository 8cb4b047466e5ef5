#!/bin/bash
# SDXL-shape UNet at 4x128x128, batch 12: the 8-phase kernel from how many tiles on?
mkdir -p gpurun_out
for rep in 1 2; do
for t in 256 160 0off; do
  echo "== UWU_P8_MINTILES=$t (rep $rep)"
  env $([ $t = 0off ] && echo UWU_GEMM_P8=0 || echo UWU_P8_MINTILES=$t) timeout -k 10 400 python bench.py --model SDXL-UNet --latent 128 --batch 12 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'gemm family frac', r.get('frac'), 'gemm ms', r.get('gemm_ms_per_step'))
        for k in r.get('kernels', [])[:5]: print('   ', {a: k[a] for a in ('kernel', 'launches_per_step', 'avg_launch_us', 'ms_per_step', 'tflops')})
" || exit 1
done; done > gpurun_out/r4_m15_unet.txt 2>&1
cat gpurun_out/r4_m15_unet.txt | cut -c1-220
