#!/bin/bash
# round 3, call 9: LayerNorm backward with packed row copies (D >= 768), four-rows-per-wave forward limited to D <= 768
cd "$(dirname "$0")/.."
out=gpurun_out/r3m9; rm -rf $out; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "ln or norm or layernorm" > $out/tests.txt 2>&1 || { tail -20 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
for cfg in "--model DiT-XL/2 --batch 192 --dtype fp8" "--model DiT-B/2 --batch 256" "--model DiT-XL/2 --batch 192"; do
  python bench.py $cfg --steps 8 --warmup 3 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | tail -1 > $out/b.json &&
  python - "$cfg" <<'PY' | tee -a $out/summary.txt
import json, sys
d = json.load(open('gpurun_out/r3m9/b.json'))
ks = {k['kernel'][:14]: k['avg_launch_us'] for k in d['roofline']['kernels']}
print(sys.argv[1], d['value'], d['ms_per_step'], 'ln_fwd', ks.get('add_ln_mod_fwd'), 'ln_bwd', ks.get('add_ln_mod_bwd'), 'loss', d['final_loss'])
PY
done
