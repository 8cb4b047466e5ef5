"""DiT-S/2 weight gradients at small per-GPU batches (K = 256 x batch tokens), us per launch (slices + reduce).
Usage: [UWU_TR_SPLIT=n] python tools/probe_wgrad_dit_small.py [batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

for batch in [int(a) for a in sys.argv[1:]] or [16, 64]:
    k = 256 * batch
    tot = 0.0
    for (m, n) in [(1152, 384), (384, 384), (1536, 384), (384, 1536)]:
        a = torch.randn(k, m, device="cuda").bfloat16()
        b = torch.randn(k, n, device="cuda").bfloat16()
        out = torch.zeros(m, n, device="cuda")
        bias = torch.zeros(m, device="cuda")
        us = timeit(lambda: ops.gemm_wgrad_shared(a, b, out, blocks=768, bias_grad=bias))
        tot += us
        print(f"batch {batch}: dW[{m},{n}] K={k}: {us:7.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
    print(f"batch {batch}: one layer {tot:.1f} us")
