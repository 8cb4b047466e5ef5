#!/bin/bash
# PMC passes (rocprofv3, separate runs per counter group) over the attention kernels at T = 256, head dim 64, B = 768:
# the per-head kernels of attention_mfma.hip and the persistent ones of attention_p256.hip, from ONE command
# (tools/bench_attn.py runs both variants).  Summary -> gpurun_out/pmc_attn.txt (copy to profiles/).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/pmc_attn
rm -rf $out; mkdir -p $out
B=${1:-768}
run() { rocprofv3 --pmc $2 -d $out/$1 -o $1 --output-format csv -- python tools/bench_attn.py $B > $out/$1.log 2>&1; }
run p1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" &&
run p2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" &&
run p3 "FETCH_SIZE" &&
run p4 "WRITE_SIZE" &&
run p5 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
kern = sorted({k for k, _ in agg})
with open(out + ".txt", "w") as fo:
    def w(s):
        print(s); fo.write(s + "\n")
    w("# rocprofv3 --pmc passes over `python tools/bench_attn.py 768` (T = 256, 6 heads x 64, B = 768: 4608 heads), per launch averages")
    for kn in kern:
        c = {cn: v / n for (k2, cn), (v, n) in agg.items() if k2 == kn}
        w(f"\n== {kn}")
        for cn in sorted(c):
            w(f"   {cn:28s} {c[cn]:18.1f}")
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            w(f"   -> wave time parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)            {c.get('SQ_WAIT_ANY', 0) / wc:6.1%}")
            w(f"   -> issue stalls (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)           {c.get('SQ_WAIT_INST_ANY', 0) / wc:6.1%}")
            w(f"   -> issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)              {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1%}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
            w(f"   -> MFMA pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x SQ_BUSY_CYCLES... see note)) raw ratio {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            w(f"   -> LDS bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:6.1%}")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            w(f"   -> HBM bytes per launch: read 2 x FETCH_SIZE KiB = {2 * c['FETCH_SIZE'] * 1024 / 1e6:8.1f} MB, written WRITE_SIZE KiB = {c['WRITE_SIZE'] * 1024 / 1e6:8.1f} MB"
              f"  (gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled, MI355X_MICROARCH.md)")
PY
