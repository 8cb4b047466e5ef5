#!/bin/bash
# PMC passes (rocprofv3, separate runs per counter group) over the attention kernels at T = 256, head dim 64, B = 768:
# the per-head kernels of attention_mfma.hip and the persistent ones of attention_p256.hip, from ONE command
# (tools/bench_attn.py runs both variants).  Summary -> gpurun_out/pmc_attn.txt (copy to profiles/).
# Other kernels: PMC_SCRIPT="tools/bench_kernels.py 768 ln" PMC_MATCH=ln_mod PMC_NAME=pmc_ln bash tools/pmc_attn.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/${PMC_NAME:-pmc_attn}
export PMC_MATCH=${PMC_MATCH:-attn}
rm -rf $out; mkdir -p $out
B=${1:-768}
script=${PMC_SCRIPT:-tools/bench_attn.py $B}
run() { rocprofv3 --pmc $2 -d $out/$1 -o $1 --output-format csv -- python $script > $out/$1.log 2>&1; }
run p1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" &&
run p2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" &&
run p3 "FETCH_SIZE" &&
run p4 "WRITE_SIZE" &&
run p5 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
python tools/summarize_pmc_kernels.py $out "$PMC_MATCH" "rocprofv3 --pmc passes over: python $script" | tee $out.txt
