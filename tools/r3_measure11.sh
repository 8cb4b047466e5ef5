#!/bin/bash
# round 3, call 11: gemm_wide_kernel after the restructuring for the (opt-in) persistent form: does one-workgroup-per-tile
# still run as before?  Same box: this build, then the library built from the previous gemm.hip (tools/_libuwu_old.so).
cd "$(dirname "$0")/.."
out=gpurun_out/r3m11; rm -rf $out; mkdir -p $out
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt
  cp uwudiff_amd/libuwu_hip.so $out/new.so && cp tools/_libuwu_old.so uwudiff_amd/libuwu_hip.so
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt
  cp $out/new.so uwudiff_amd/libuwu_hip.so
done
rm -f $out/new.so
