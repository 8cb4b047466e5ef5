"""Group a rocprofv3 kernel trace by (kernel, grid, workgroup) and print the heaviest groups.
Usage: python tools/trace_shapes.py <*_kernel_trace.csv> [substring]"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if sub and sub not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    acc[key][0] += 1
    acc[key][1] += d
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / tot * 100:5.1f}%  n={n:5d}  avg={t / n:9.1f} us  grid=({k[1]},{k[2]},{k[3]}) wg={k[4]}  {k[0]}")
