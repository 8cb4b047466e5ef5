#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r3m2; mkdir -p $o
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > $o/attn_test.log 2>&1; echo "attn tests rc=$?"; tail -4 $o/attn_test.log
timeout -k 10 200 python -m pytest tests/test_gemm_gpu.py -m gpu -q -k "exact_integers_all" > $o/gemm_as_test.log 2>&1; echo "gemm_as rc=$?"; tail -2 $o/gemm_as_test.log
timeout -k 10 200 python tools/bench_attn.py 768 256 > $o/bench_attn.txt 2>&1; cat $o/bench_attn.txt
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary > $o/bench.json 2> $o/bench.err; python -c "
import json; d=json.loads(open('$o/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); [print('  ',k['kernel'][:30],k['avg_launch_us'],k['frac'],k['frac_hbm']) for k in d['roofline']['kernels']]"
