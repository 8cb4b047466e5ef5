#!/bin/bash
# fp8 tests, then the DiT-XL/2 fp8 step with / without the unmasked GELU-emit epilogue for full tiles
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_fp8_gpu.py -x -q > gpurun_out/r4_m25_tests.txt 2>&1 || { tail -30 gpurun_out/r4_m25_tests.txt; exit 1; }
tail -2 gpurun_out/r4_m25_tests.txt
for rep in 1 2 3; do
for on in 1 0; do
  echo "== UWU_F8_EMIT_FULL=$on (rep $rep)"
  UWU_F8_EMIT_FULL=$on timeout -k 10 300 python bench.py --model DiT-XL/2 --batch 192 --dtype fp8 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'final loss', d.get('final_loss'))
" || exit 1
done; done > gpurun_out/r4_m25_step.txt 2>&1
cat gpurun_out/r4_m25_step.txt
