#!/bin/bash
# Round-4 profiles (run once at the end of the round on the final kernels; copy the summaries to profiles/):
#   1. the round profile of the default bench command (kernel stats + FETCH / WRITE passes)          -> gpurun_out/prof_r04/
#   2. PMC over the headline GEMM family (bench_kernels.py 768 gemm)                                  -> gpurun_out/r04_pmc_gemm_family.txt
#   3. PMC over the 8-phase kernels: bf16 on DiT-B/2 shapes, fp8 on DiT-XL/2 shapes                     -> r04_pmc_gemm_p8.txt, r04_pmc_gemm_p8f.txt
#   4. PMC over the attention kernels, head dim 64 (B = 768) and 72 (B = 192)                           -> r04_pmc_attention.txt, r04_pmc_attention_d72.txt
#   5. kernel stats grouped by (kernel, grid): DiT-XL/2 fp8 / bf16, DiT-B/2, SDXL-shape UNet without the side stream
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
step=${1:-all}
if [ $step = all ] || [ $step = 1 ]; then
  bash tools/profile_round.sh r04 || exit 1
  echo "[1] round profile done"
fi
if [ $step = all ] || [ $step = 2 ]; then
  PMC_SCRIPT="tools/bench_kernels.py 768 gemm" PMC_MATCH="gemm_|splitk" PMC_NAME=r04_pmc_gemm_family bash tools/pmc_attn.sh > /dev/null 2>&1 || exit 1
  echo "[2] gemm family pmc done"
fi
if [ $step = all ] || [ $step = 3 ]; then
  PMC_SCRIPT="tools/bench_gemm_shapes.py 768 256" PMC_MATCH="gemm_p8|gemm_trw|gemm_big" PMC_NAME=r04_pmc_gemm_p8 bash tools/pmc_attn.sh > /dev/null 2>&1 || exit 1
  PMC_SCRIPT="tools/bench_gemm_fp8_shapes.py 1152 192" PMC_MATCH="gemm_p8f|gemm_f8" PMC_NAME=r04_pmc_gemm_p8f bash tools/pmc_attn.sh > /dev/null 2>&1 || exit 1
  echo "[3] 8-phase pmc done"
fi
if [ $step = all ] || [ $step = 4 ]; then
  PMC_NAME=r04_pmc_attention bash tools/pmc_attn.sh 768 > /dev/null 2>&1 || exit 1
  PMC_SCRIPT="tools/bench_attn72.py 192" PMC_MATCH="attn" PMC_NAME=r04_pmc_attention_d72 bash tools/pmc_attn.sh > /dev/null 2>&1 || exit 1
  echo "[4] attention pmc done"
fi
if [ $step = all ] || [ $step = 5 ]; then
  out=gpurun_out/r04_shapes; rm -rf $out; mkdir -p $out
  for cfg in "DiT-XL/2 192 fp8 xl2_fp8" "DiT-XL/2 192 bf16 xl2_bf16" "DiT-B/2 256 bf16 b2_bf16"; do
    set -- $cfg
    rocprofv3 --kernel-trace --stats -d $out/$4 -o t --output-format csv -- python bench.py --model $1 --batch $2 --dtype $3 --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary > $out/$4.log 2>&1 &&
    python tools/trace_shapes.py $out/$4/t_kernel_trace.csv > $out/r04_shapes_$4.txt 2>&1
    rm -rf $out/$4
  done
  UWU_UNET_FORK=0 rocprofv3 --kernel-trace --stats -d $out/unet -o t --output-format csv -- python bench.py --model SDXL-UNet --latent 128 --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-secondary > $out/unet.log 2>&1 &&
  python tools/trace_shapes.py $out/unet/t_kernel_trace.csv > $out/r04_shapes_unet_nofork.txt 2>&1
  rm -rf $out/unet
  echo "[5] shape tables done"
fi
ls -la gpurun_out/prof_r04 gpurun_out/r04_shapes 2>/dev/null | head -30
