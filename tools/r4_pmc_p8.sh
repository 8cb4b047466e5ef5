#!/bin/bash
# PMC passes over the 8-phase GEMM kernels on DiT-B/2 shapes (and the round-3 kernels they replace, UWU_GEMM_P8=0 / P8W=0)
export UWU_BENCH_ONLY=fc1_fwd,fc2_fwd,fc1_dgrad,fc1_wgrad,fc2_wgrad
PMC_SCRIPT="tools/bench_gemm_shapes.py 768 256" PMC_MATCH="gemm_p8|gemm_trw|splitk" PMC_NAME=r4_pmc_p8 bash tools/pmc_attn.sh > /dev/null 2>&1
UWU_GEMM_P8W=0 PMC_SCRIPT="tools/bench_gemm_shapes.py 768 256" PMC_MATCH="gemm_trw" PMC_NAME=r4_pmc_trw bash tools/pmc_attn.sh > /dev/null 2>&1
cat gpurun_out/r4_pmc_p8.txt gpurun_out/r4_pmc_trw.txt | grep -v "^#"
