"""Weight-gradient launches of the SDXL-shape UNet's Linears at 4x128x128 latents (K = tokens), us per launch.
Usage: [UWU_TR_SPLIT=n] python tools/probe_wgrad_unet.py [batch]      (default batch 12: the bench's)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

tot = 0.0
Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 12
T32, T64 = Bt * 1024, Bt * 4096
for (m, n, k, cnt) in [(1280, 1280, T32, 180), (3840, 1280, T32, 60), (10240, 1280, T32, 60), (1280, 5120, T32, 60),
                       (640, 640, T64, 30), (1920, 640, T64, 10), (5120, 640, T64, 10), (640, 2560, T64, 10)]:
    a = torch.randn(k, m, device="cuda").bfloat16()
    b = torch.randn(k, n, device="cuda").bfloat16()
    out = torch.zeros(m, n, device="cuda")
    us = timeit(lambda: ops.gemm_wgrad_shared(a, b, out, blocks=768))
    tot += us * cnt
    print(f"dW[{m},{n}] K={k} x{cnt}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
print(f"sum over one step: {tot / 1e3:.2f} ms")
