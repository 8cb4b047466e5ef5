"""UNet weight-gradient launch dW[1280, 1280] over 12288 tokens (and the 640-wide one) with warm and with flushed caches
(a 512 MB fill between launches): the in-step case is the cold one.   Usage: python tools/probe_wgrad_cold.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

big = torch.empty(1 << 29, device="cuda", dtype=torch.uint8)
flush = min(timeit(lambda: big.zero_()) for _ in range(3))
for (m, n, k) in [(1280, 1280, 12288), (3840, 1280, 12288), (640, 640, 49152), (1280, 5120, 12288)]:
    a = torch.randn(k, m, device="cuda").bfloat16()
    b = torch.randn(k, n, device="cuda").bfloat16()
    out = torch.zeros(m, n, device="cuda")
    fn = lambda: ops.gemm_wgrad_shared(a, b, out, blocks=768)

    def cold():
        big.zero_()
        fn()

    w = min(timeit(fn) for _ in range(3))
    c = min(timeit(cold) for _ in range(3)) - flush
    print(f"dW[{m},{n}] K={k}: warm {w:7.1f} us ({2.0 * m * n * k / w / 1e6:6.1f} TFLOP/s)   cold {c:7.1f} us ({2.0 * m * n * k / c / 1e6:6.1f} TFLOP/s)", flush=True)

# sustained: 1500 launches back to back (the step keeps the chip at its loaded clock), and with the fused bias gradient
a = torch.randn(12288, 1280, device="cuda").bfloat16()
b = torch.randn(12288, 1280, device="cuda").bfloat16()
out = torch.zeros(1280, 1280, device="cuda")
bg = torch.zeros(1280, device="cuda")
for name, fn in (("plain", lambda: ops.gemm_wgrad_shared(a, b, out, blocks=768)),
                 ("+bias gradient", lambda: ops.gemm_wgrad_shared(a, b, out, blocks=768, bias_grad=bg)),
                 ("+bias gradient, blocks=256", lambda: ops.gemm_wgrad_shared(a, b, out, blocks=256, bias_grad=bg))):
    us = timeit(fn, iters=1500, warm=20)
    print(f"dW[1280,1280] K=12288 {name:28s}: {us:7.1f} us per launch over 1500 launches", flush=True)
