#!/bin/bash
# round 3, call 16: split-K reduce with four slice lanes per output column group (UWU_SPLITK_REDUCE4): GEMM tests + step A/B
cd "$(dirname "$0")/.."
out=gpurun_out/r3m16; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_conv_gpu.py tests/test_fp8_gpu.py -q -x -m gpu > $out/tests.txt 2>&1 || { tail -30 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
for e in 1 0 1 0; do
  UWU_SPLITK_REDUCE4=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=[x for x in d['roofline']['kernels'] if 'wgrad' in x['kernel']][0]; print('REDUCE4=$e', d['value'], d['ms_per_step'], 'wgrad', k['avg_launch_us'], k['ms_per_step'])" | tee -a $out/ab.txt
done
for e in 1 0; do
  UWU_SPLITK_REDUCE4=$e python bench.py --model SDXL-UNet --latent 128 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('UNet REDUCE4=$e', d['value'], d['ms_per_step'])" | tee -a $out/ab.txt
done
