#!/bin/bash
# PMC passes over one GEMM shape of tools/bench_kernels.py (UWU_BENCH_ONLY=<name>); summaries -> gpurun_out/pmc_<name>.txt
# usage: bash tools/pmc_gemm.sh qkv_fwd
cd "$(dirname "$0")/.."
export TMPDIR=/tmp UWU_BENCH_ONLY=$1
out=gpurun_out/pmc_$1
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $out/p1 -o p1 --output-format csv -- python tools/bench_kernels.py 256 ${PMC_WHICH:-gemm} > $out/p1.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $out/p2 -o p2 --output-format csv -- python tools/bench_kernels.py 256 ${PMC_WHICH:-gemm} > $out/p2.log 2>&1 &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM -d $out/p3 -o p3 --output-format csv -- python tools/bench_kernels.py 256 ${PMC_WHICH:-gemm} > $out/p3.log 2>&1 &&
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE -d $out/p4 -o p4 --output-format csv -- python tools/bench_kernels.py 256 ${PMC_WHICH:-gemm} > $out/p4.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ.get("PMC_MATCH", "gemm") not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
with open(out + ".txt", "w") as fo:
    for (kn, c), (v, n) in sorted(agg.items()):
        line = f"{kn:60s} {c:34s} avg/launch {v / n:16.1f}  (n={n})"
        print(line); fo.write(line + "\n")
PY
