#!/bin/bash
# (1) tests touched by the oracle-build shortcut and by the batched weight amax passes; (2) DiT-XL/2 fp8 step before / after is a
# cross-call comparison (the old library is gone): same box, three runs for the spread
cd "$(dirname "$0")/.."
timeout -k 10 700 python -m pytest tests/test_fp8_gpu.py tests/test_configs_gpu.py tests/test_train_gpu.py -x -q -m gpu --durations=8 > gpurun_out/r4_m22_tests.txt 2>&1 || { tail -30 gpurun_out/r4_m22_tests.txt; exit 1; }
tail -14 gpurun_out/r4_m22_tests.txt
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --model DiT-XL/2 --batch 192 --dtype fp8 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'final loss', d.get('final_loss'))
" || exit 1
done > gpurun_out/r4_m22_fp8.txt 2>&1
cat gpurun_out/r4_m22_fp8.txt
