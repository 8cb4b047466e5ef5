"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-launch HBM traffic of the GEMM kernels.

MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB (x1024); on gfx950 FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read -> doubled.  WRITE_SIZE is exact for 16-B/lane stores and fp32 atomics."""
import csv
import glob
import hashlib
import json
import os
import sys


def load(dirname, counter):
    f = glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True)[0]
    tot, n = 0.0, 0
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        reduce = "splitk_reduce" in k  # second kernel of a weight-gradient launch: bytes count, launches do not
        if "gemm" not in k and not reduce:
            continue
        fam = "fp32" if ("IffLb" in k or "<float" in k or "gemm_r3_kernelIf" in k) else "bf16"
        if fam != "bf16":
            continue
        tot += float(r["Counter_Value"])
        n += 0 if reduce else 1
        per[k[:70]] = per.get(k[:70], 0) + float(r["Counter_Value"])
    return tot, n, per


fetch, nf, pf = load(sys.argv[1], "FETCH_SIZE")
write, nw, pw = load(sys.argv[2], "WRITE_SIZE")
_csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "uwudiff_amd", "csrc")
_h = hashlib.sha256()
for _name in ("gemm_shared.h", "gemm.hip", "gemm_p8.hip", "gemm_p8n.hip", "gemm_p8f.hip"):  # = bench.py GEMM_SOURCES
    _h.update(open(os.path.join(_csrc, _name), "rb").read())
out = {
    "gemm_src_sha16": _h.hexdigest()[:16],  # bench.py reports the number only for these sources
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline",
    "kernel_family": "bf16 MFMA GEMM family: every kernel with gemm in its name (+ splitk_reduce bytes), fwd + dgrad + wgrad launches",
    "launches": nf,
    "fetch_kib_sum_raw": fetch, "write_kib_sum": write,
    "hbm_bytes_per_launch": (2.0 * fetch / nf + write / nw) * 1024.0,
    "read_bytes_per_launch_corrected": 2.0 * fetch / nf * 1024.0,
    "write_bytes_per_launch": write / nw * 1024.0,
    "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), units KiB -> x1024",
}
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[3], "w"), indent=1)


# ---- per-kernel table (every kernel of the step, not only the GEMM family): counter-check of the algorithmic GB/s figures
def by_kernel(dirname, counter):
    import re

    f = glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(\w+_kernel|attn_\w+|\w+kernel\w*)(<[^>]*>)?", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]
        a = acc.setdefault(k[:72], [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return acc


fk, wk = by_kernel(sys.argv[1], "FETCH_SIZE"), by_kernel(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(fk) | set(wk)):
    rd = 2.0 * fk.get(k, [0, 1])[0] / max(fk.get(k, [0, 1])[1], 1) * 1024 / 1e6
    wr = wk.get(k, [0, 1])[0] / max(wk.get(k, [0, 1])[1], 1) * 1024 / 1e6
    rows.append((rd + wr, k, fk.get(k, [0, 0])[1], rd, wr))
rows.sort(reverse=True)
table = sys.argv[3].replace(".json", "_by_kernel.txt")
with open(table, "w") as fo:
    fo.write("# HBM traffic per launch by kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over `python bench.py`;\n"
             "# read = 2 x FETCH_SIZE KiB (gfx950 tallies 128-B requests at 64 B), written = WRITE_SIZE KiB; MB = 1e6 bytes)\n")
    fo.write(f"{'kernel':72s} {'launches':>8s} {'read MB':>10s} {'written MB':>11s} {'total MB':>10s}\n")
    for tot, k, n, rd, wr in rows[:40]:
        fo.write(f"{k:72s} {n:8d} {rd:10.1f} {wr:11.1f} {tot:10.1f}\n")
print(open(table).read())
