"""fp8 mode: LayerNorm + modulate followed by the quantising pass against the LayerNorm that emits the fp8 images itself.
Usage: python tools/bench_ln_q8.py [B] [D]      (M = B x 256 token rows)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import lib as L  # noqa: E402
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 1152
    T = 256
    M, ML = B * T, 6 * D
    x = torch.randn(M, D, device="cuda").bfloat16()
    y = torch.randn(M, D, device="cuda").bfloat16()
    mod = torch.randn(B, ML, device="cuda") * 0.5
    g, sh, sc = mod[:, 0:D], mod[:, D:2 * D], mod[:, 2 * D:3 * D]
    qs = torch.ones(1, device="cuda")
    amax = torch.zeros(1, device="cuda")
    x_out = torch.empty_like(x)
    h = torch.empty_like(x)
    q8 = torch.empty(M, D, device="cuda", dtype=torch.uint8)
    q8t = torch.empty(D, M, device="cuda", dtype=torch.uint8)
    mean = torch.empty(M, device="cuda")
    rstd = torch.empty(M, device="cuda")
    big = torch.empty(1 << 28, device="cuda", dtype=torch.uint8)  # 256 MB: flushes the Infinity Cache between runs

    def plain():
        L.call("uwu_add_ln_modulate_fwd", L.ptr(x), L.ptr(y), g.data_ptr(), sh.data_ptr(), sc.data_ptr(), ML, L.ptr(x_out), L.ptr(h),
               L.ptr(mean), L.ptr(rstd), B, T, D, 1e-6, 0, L.dt(x), L.stream())

    def quant():
        L.call("uwu_fp8_quantize", L.ptr(h), L.dt(h), M, D, D, L.ptr(qs), 0, L.ptr(q8), D, L.ptr(q8t), M, L.ptr(amax), None, L.stream())

    def emit():
        L.call("uwu_add_ln_modulate_fwd_q8", L.ptr(x), L.ptr(y), g.data_ptr(), sh.data_ptr(), sc.data_ptr(), ML, L.ptr(x_out), L.ptr(q8), D,
               L.ptr(q8t), M, L.ptr(qs), L.ptr(amax), L.ptr(mean), L.ptr(rstd), B, T, D, 1e-6, L.stream())

    def cold(fn):
        def run():
            big.zero_()
            fn()
        return run

    flush = min(timeit(lambda: big.zero_()) for _ in range(3))
    for name, fn in (("LayerNorm (bf16 h)", plain), ("quantise h", quant), ("LayerNorm -> fp8", emit)):
        warm = min(timeit(fn) for _ in range(3))
        cd = min(timeit(cold(fn)) for _ in range(3)) - flush
        print(f"M={M} D={D} {name:22s} warm {warm:7.1f} us   cold {cd:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
