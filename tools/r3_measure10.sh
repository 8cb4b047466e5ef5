#!/bin/bash
# round 3, call 10: fp8 GEMM epilogues that emit the next operand (fc1 -> gelu(u) e4m3, fc2 dgrad -> du e5m2): tests + A/B
cd "$(dirname "$0")/.."
out=gpurun_out/r3m10; rm -rf $out; mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_fp8_gpu.py -q -x -m gpu > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
for e in 1 0 1 0; do
  UWU_F8_EMIT=$e python bench.py --model DiT-XL/2 --batch 192 --dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | tail -1 > $out/b.json &&
  python - "$e" <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.load(open('gpurun_out/r3m10/b.json'))
ks = {k['kernel'][:22]: (k['avg_launch_us'], k['ms_per_step']) for k in d['roofline']['kernels']}
print('EMIT=' + sys.argv[1], d['value'], d['ms_per_step'], 'loss', d['final_loss'], ks.get('gemm fc1 + bias + GELU'), ks.get('gemm fc2 dgrad + dGELU'))
PY
done
