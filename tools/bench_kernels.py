"""Per-kernel timing on the GPU (HIP events on the launch stream), DiT-S/2 shapes at per-GPU batch B.
Usage: python tools/bench_kernels.py [B] [which...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import lib as L  # noqa: E402
from uwudiff_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    which = set(sys.argv[2:])
    T, D, H, d = 256, int(os.environ.get("UWU_BENCH_D", "384")), 6, 64  # (UWU_BENCH_D=768 / 1152 with "ln": the wide-row LayerNorm forms)
    M = B * T
    bf = torch.bfloat16
    dev = "cuda"
    torch.manual_seed(0)

    def want(k):
        return not which or k in which

    if want("attn"):
        qkv = torch.randn(M, 3 * D, device=dev).to(bf)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        o, lse = ops.attention_fwd(q, k, v, B, T, T, H, d)
        do = torch.randn(M, D, device=dev).to(bf)
        dqkv = torch.empty_like(qkv)
        fl = 4 * T * T * d * B * H
        us = timeit(lambda: ops.attention_fwd(q, k, v, B, T, T, H, d))
        print(f"attn_fwd  {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s")
        us = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, T, T, H, d))
        print(f"attn_bwd  {us:9.1f} us  {2.5 * fl / us / 1e6:8.1f} TFLOP/s")
    if want("gemm"):
        for name, (m, n, k, ta, tb) in {
            "qkv_fwd": (M, 3 * D, D, 0, 0), "proj_fwd": (M, D, D, 0, 0), "fc1_fwd": (M, 4 * D, D, 0, 0),
            "fc2_fwd": (M, D, 4 * D, 0, 0), "qkv_dgrad": (M, D, 3 * D, 0, 1), "fc1_dgrad": (M, D, 4 * D, 0, 1),
            "fc2_dgrad": (M, 4 * D, D, 0, 1), "qkv_wgrad": (3 * D, D, M, 1, 1), "fc1_wgrad": (4 * D, D, M, 1, 1),
            "fc2_wgrad": (D, 4 * D, M, 1, 1), "proj_wgrad": (D, D, M, 1, 1),
        }.items():
            if os.environ.get("UWU_BENCH_ONLY") and name not in os.environ["UWU_BENCH_ONLY"].split(","):
                continue
            a = torch.randn((k, m) if ta else (m, k), device=dev).to(bf)
            b = torch.randn((k, n) if tb else (n, k), device=dev).to(bf)
            fl = 2.0 * m * n * k
            if ta:
                out = torch.zeros(m, n, device=dev)
                target = int(os.environ.get('UWU_WGRAD_BLOCKS', '512'))
                scratch = ops.gemm_wgrad_scratch(m, n, k) if os.environ.get('UWU_WGRAD_SCRATCH', '1') == '1' else None
                fn = lambda: ops.gemm_wgrad(a, b, out, blocks=target, scratch=scratch)
            else:
                out = torch.empty(m, n, device=dev, dtype=bf)
                fn = lambda: ops.gemm(a, b, trans_a=False, trans_b=bool(tb), out=out)
                if name == "fc1_fwd":  # the in-step call: bias + GELU, two outputs
                    bias, out2 = torch.randn(n, device=dev), torch.empty(m, n, device=dev, dtype=bf)
                    us = timeit(lambda: ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU, out=out, out2=out2))
                    print(f"{name + '+gelu':10s} {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s   ({m}x{n}x{k})")
                if name == "fc2_dgrad":  # the in-step call: x gelu'(u) and the fc1 bias gradient
                    u, cs = torch.randn(m, n, device=dev).to(bf), torch.zeros(n, device=dev)
                    us = timeit(lambda: ops.gemm(a, b, trans_b=True, aux=u, epilogue=L.EPI_DGELU, out=out, out2=cs))
                    print(f"{name + '+dg':10s} {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s   ({m}x{n}x{k})")
                    us = timeit(lambda: ops.gemm(a, b, trans_b=True, aux=u, epilogue=L.EPI_DGELU, out=out))
                    print(f"{name + '+dg-cs':10s} {us:9.1f} us  (no column sums)")
            us = timeit(fn)
            print(f"{name:10s} {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s   ({m}x{n}x{k})")
    if want("ln"):
        x = torch.randn(M, D, device=dev).to(bf)
        y = torch.randn(M, D, device=dev).to(bf)
        mod = torch.randn(B, 6 * D, device=dev)
        fn = lambda: ops.add_ln_modulate_fwd(x, B, T, y=y, gate=mod[:, :D], shift=mod[:, D:2 * D], scale=mod[:, 2 * D:3 * D], mod_ld=6 * D)
        us = timeit(fn)
        print(f"ln_fwd    {us:9.1f} us  {4 * M * D * 2 / us / 1e3:8.1f} GB/s")
        xo, h, mean, rstd = fn()
        dmod = torch.zeros(B, 6 * D, device=dev)
        fnb = lambda: ops.add_ln_modulate_bwd(h, xo, mean, rstd, B, T, scale=mod[:, 2 * D:3 * D], dx_in=x, y=y, gate=mod[:, :D],
                                              mod_ld=6 * D, dshift=dmod[:, D:2 * D], dscale=dmod[:, 2 * D:3 * D], dgate=dmod[:, :D])
        us = timeit(fnb)
        print(f"ln_bwd    {us:9.1f} us  {6 * M * D * 2 / us / 1e3:8.1f} GB/s")
        u = torch.randn(M, 4 * D, device=dev).to(bf)
        out = torch.zeros(4 * D, device=dev)
        us = timeit(lambda: ops.colsum(u, out=out, accumulate=True))
        print(f"colsum4D  {us:9.1f} us  {M * 4 * D * 2 / us / 1e3:8.1f} GB/s")


if __name__ == "__main__":
    main()
