"""Per-kernel summary of rocprofv3 --pmc passes (one counter group per run; tools/pmc_attn.sh): wait / stall / issue shares,
MFMA pipe utilisation, LDS bank conflicts and HBM bytes per launch with the gfx950 corrections of MI355X_MICROARCH.md.
usage: python tools/summarize_pmc_kernels.py <dir with p*/..counter_collection.csv> <kernel-name regex> [title]"""
import collections
import csv
import glob
import re
import sys

out, match = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w*(?:" + match + r")\w*?)(?:I[A-Za-z0-9_]*E[vE]\w*)?(<[^>]*>)?(?:\(|$)", r["Kernel_Name"])
        if not m:
            continue
        name = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", m.group(1)) + (m.group(2) or "")
        k = (name, r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
print(f"# {title}")
print("# rocprofv3 --pmc, one counter group per run, per-launch averages.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles,\n"
      "# SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32); GRBM_GUI_ACTIVE is summed over the 8 XCDs.")
for kn in sorted({k for k, _ in agg}):
    c = {cn: v / n for (k2, cn), (v, n) in agg.items() if k2 == kn}
    print(f"\n== {kn}   ({int(agg[(kn, 'SQ_WAVES')][1]) if (kn, 'SQ_WAVES') in agg else '?'} launches sampled)")
    for cn in sorted(c):
        print(f"   {cn:28s} {c[cn]:18.1f}")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        print(f"   -> wave time parked at s_waitcnt / s_barrier (SQ_WAIT_ANY / SQ_WAVE_CYCLES)   {c.get('SQ_WAIT_ANY', 0) / wc:6.1%}")
        print(f"   -> issue stalls (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)                           {c.get('SQ_WAIT_INST_ANY', 0) / wc:6.1%}"
              f"   of which LDS issue {c.get('SQ_WAIT_INST_LDS', 0) / wc:6.1%}")
        print(f"   -> issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)                              {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1%}")
    if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        print(f"   -> kernel ~ {cyc:.3e} shader cycles; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) = "
              f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):6.1%}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   -> LDS bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)              {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:6.1%}")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        print(f"   -> HBM per launch: read 2 x FETCH_SIZE KiB = {2 * c['FETCH_SIZE'] * 1024 / 1e6:8.1f} MB, written WRITE_SIZE KiB = "
              f"{c['WRITE_SIZE'] * 1024 / 1e6:8.1f} MB")
