#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r3m4; mkdir -p $o
for f in 1 0 1 0; do
  UWU_ATTN_P256F=$f timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('P256F=$f', d['value'], d['ms_per_step'], [(k['kernel'][:13],k['avg_launch_us']) for k in d['roofline']['kernels'] if 'attn' in k['kernel']])"
done
