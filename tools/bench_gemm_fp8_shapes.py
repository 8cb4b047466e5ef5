"""fp8 GEMM launches of a DiT block at width D (forward / input gradient / weight gradient of qkv, proj, fc1, fc2), one by one.
Usage: python tools/bench_gemm_fp8_shapes.py [D] [B]      (default 1152 192: DiT-XL/2)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import lib as L  # noqa: E402
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1152
B = int(sys.argv[2]) if len(sys.argv) > 2 else 192
M = B * 256
one = torch.ones(1, device="cuda")
tot = 0.0


def rnd(r, c):
    return torch.randint(0, 120, (r, c), device="cuda", dtype=torch.uint8)  # small positive e4m3 / e5m2 bytes


for name, m, n, k, epi, fa in [("qkv fwd", M, 3 * D, D, L.EPI_NONE, ops.FP8_E4M3), ("proj fwd", M, D, D, L.EPI_NONE, ops.FP8_E4M3),
                               ("fc2 fwd", M, D, 4 * D, L.EPI_NONE, ops.FP8_E4M3), ("qkv dgrad", M, D, 3 * D, L.EPI_NONE, ops.FP8_E5M2),
                               ("proj dgrad", M, D, D, L.EPI_NONE, ops.FP8_E5M2), ("fc1 dgrad", M, D, 4 * D, L.EPI_NONE, ops.FP8_E5M2),
                               ("qkv wgrad", 3 * D, D, M, L.EPI_ACCUM, ops.FP8_E5M2), ("proj wgrad", D, D, M, L.EPI_ACCUM, ops.FP8_E5M2),
                               ("fc1 wgrad", 4 * D, D, M, L.EPI_ACCUM, ops.FP8_E5M2), ("fc2 wgrad", D, 4 * D, M, L.EPI_ACCUM, ops.FP8_E5M2)]:
    a, b = rnd(m, k), rnd(n, k)
    out = torch.zeros(m, n, device="cuda", dtype=torch.float32 if epi == L.EPI_ACCUM else torch.bfloat16)
    us = timeit(lambda: ops.gemm_fp8(a, b, one, one, fmt_a=fa, epilogue=epi, out=out))
    tot += us
    print(f"{name:11s} [{m:6d} x {n:5d}] K={k:6d}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s ({2.0 * m * n * k / us / 5e9:.3f} of 5 PF)",
          flush=True)
print(f"sum {tot:.0f} us")
