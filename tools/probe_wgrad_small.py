"""Weight-gradient launches with a SHORT reduction (cross-attention key/value weights: K = B x 77 tokens), us per launch.
Usage: python tools/probe_wgrad_small.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

for (m, n, k) in [(1280, 2048, 462), (640, 2048, 462), (1280, 2048, 3696), (640, 2048, 3696), (1280, 1280, 6), (320, 1280, 6)]:
    a = torch.randn(k, m, device="cuda").bfloat16()
    b = torch.randn(k, n, device="cuda").bfloat16()
    out = torch.zeros(m, n, device="cuda")
    for blocks in (768, 256, 1):
        us = timeit(lambda: ops.gemm_wgrad(a, b, out, blocks=blocks))
        print(f"dW[{m},{n}] K={k} blocks={blocks}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
