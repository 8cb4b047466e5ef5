"""fp8 quantisation pass (quant.hip) at the DiT-XL/2 shapes: which output costs what (HIP events).
Usage: python tools/bench_quant.py [B]      (M = B x 256 token rows; K = 1152 and 4608)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import lib as L  # noqa: E402
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    M = B * 256
    for K in (1152, 4608):
        x = torch.randn(M, K, device="cuda").bfloat16()
        scale = torch.ones(1, device="cuda")
        out = torch.empty(M, K, device="cuda", dtype=torch.uint8)
        out_t = torch.empty(K, M, device="cuda", dtype=torch.uint8)
        amax = torch.zeros(1, device="cuda")
        cs = torch.zeros(K, device="cuda")

        def run(o, ot, am, c, fmt):
            L.call("uwu_fp8_quantize", L.ptr(x), L.dt(x), M, K, K, L.ptr(scale), fmt, L.ptr(o), K, L.ptr(ot), M, L.ptr(am), L.ptr(c),
                   L.stream())

        for name, args in (("row", (out, None, None, None, 0)), ("transposed", (None, out_t, None, None, 0)),
                           ("both", (out, out_t, None, None, 0)), ("both+amax", (out, out_t, amax, None, 0)),
                           ("both+amax+colsum e5m2", (out, out_t, amax, cs, 1))):
            us = min(timeit(lambda: run(*args)) for _ in range(3))
            by = M * K * (2 + (args[0] is not None) + (args[1] is not None))
            print(f"M={M} K={K:5d} {name:24s} {us:8.1f} us  {by / us / 1e3:7.1f} GB/s", flush=True)
        us = min(timeit(lambda: out.copy_(out_t.view(M, K))) for _ in range(3))
        print(f"M={M} K={K:5d} {'(torch copy u8)':24s} {us:8.1f} us  {2 * M * K / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
