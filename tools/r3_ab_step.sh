#!/bin/bash
# same-box A/B of the whole step: this round's kernels (attention_p256, four-rows-per-wave LayerNorm forward) against round 2's
cd "$(dirname "$0")/.."
for rep in 1 2; do
for cfg in "new:1:1:1" "old:0:0:0"; do
  IFS=: read name a f l <<< "$cfg"
  UWU_ATTN_P256=$a UWU_ATTN_P256F=$f UWU_LN_ROW16=$l timeout -k 10 200 python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-sweep --no-secondary $EXTRA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], [(k['kernel'][:12],k['avg_launch_us']) for k in d['roofline']['kernels'][5:]])"
done; done
