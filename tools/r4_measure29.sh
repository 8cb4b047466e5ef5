#!/bin/bash
# kernel stats of the batch-16 step (the reference YAML's batch): where do its 3.3 ms go?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r4m29; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/b16 -o t --output-format csv -- python bench.py --batch 16 --steps 50 --warmup 10 --no-cpu-baseline --no-sweep --no-secondary > $out/b16.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4m29/b16/t_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
steps=[int(r['Calls']) for r in rows if 'adamw' in r['Name']][0]
print('steps',steps,'kernel time per step (ms)',tot/steps/1e6, 'launches per step', sum(int(r['Calls']) for r in rows)/steps)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:28]:
    print(f"{100*float(r['TotalDurationNs'])/tot:5.1f}% n/step={int(r['Calls'])/steps:6.1f} avg={float(r['AverageNs'])/1e3:7.1f}us  {r['Name'][:100]}")
PY
rm -f $out/b16/t_kernel_trace.csv
grep -o '"value": [0-9.]*' $out/b16.log | head -1
