#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/pmc_attn.sh 768 > /dev/null 2>&1; echo "attn pmc rc=$?"
PMC_SCRIPT="tools/bench_kernels.py 768 ln" PMC_MATCH=ln_mod PMC_NAME=pmc_ln bash tools/pmc_attn.sh > /dev/null 2>&1; echo "ln pmc rc=$?"
head -60 gpurun_out/pmc_ln.txt
