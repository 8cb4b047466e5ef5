#!/bin/bash
# round 3, call 8: kernel trace of the batch-16 step (the reference YAML's batch): launches per step, kernel time against wall time
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r3m8; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/b16 -o b16 --output-format csv -- python bench.py --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary > $out/b16.log 2>&1 &&
python tools/trace_shapes.py $out/b16/*/b16_kernel_trace.csv > $out/b16_shapes.txt 2>&1
python - <<'PY' > gpurun_out/r3m8/b16_timeline.txt 2>&1
import csv, glob
f = glob.glob('gpurun_out/r3m8/b16/**/b16_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 310-ish kernels: print name, start offset, duration, gap to previous end
last = rows[-700:]
t0 = int(last[0]['Start_Timestamp']); prev_end = t0
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:80]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    prev_end = max(prev_end, e)
PY
rm -f $out/b16/*/b16_kernel_trace.csv
echo "rc=$?"
