#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r3m6; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_dit_gpu.py tests/test_train_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "not sdxl and not full_depth_sdxl" > $o/test.log 2>&1; echo "tests rc=$?"; tail -3 $o/test.log
for f in 1 0 1 0; do
  UWU_DIT_MOD_BF16=$f timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MOD_BF16=$f', d['value'], d['ms_per_step'], d['final_loss'])"
done
