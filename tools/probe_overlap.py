"""Can two kernels of the step share the GPU?  Times pairs of launches back to back on one stream and concurrently on
two streams (weight-gradient GEMM next to a LayerNorm backward / an input-gradient GEMM / attention backward).
Usage: python tools/probe_overlap.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    T, D, H, d = 256, 384, 6, 64
    M = B * T
    bf, dev = torch.bfloat16, "cuda"
    torch.manual_seed(0)
    # A: fc1 weight gradient  dW[1536,384] += du[M,1536]^T h2[M,384]
    du = torch.randn(M, 4 * D, device=dev).to(bf)
    h2 = torch.randn(M, D, device=dev).to(bf)
    dw = torch.zeros(4 * D, D, device=dev)
    scratch = ops.gemm_wgrad_scratch(4 * D, D, M)
    A = lambda: ops.gemm_wgrad(du, h2, dw, scratch=scratch)
    # B: LayerNorm/modulate backward
    x = torch.randn(M, D, device=dev).to(bf)
    y = torch.randn(M, D, device=dev).to(bf)
    mod = torch.randn(B, 6 * D, device=dev)
    xo, h, mean, rstd = ops.add_ln_modulate_fwd(x, B, T, y=y, gate=mod[:, :D], shift=mod[:, D:2 * D], scale=mod[:, 2 * D:3 * D], mod_ld=6 * D)
    dmod = torch.zeros(B, 6 * D, device=dev)
    Bk = lambda: ops.add_ln_modulate_bwd(h, xo, mean, rstd, B, T, scale=mod[:, 2 * D:3 * D], dx_in=x, y=y, gate=mod[:, :D], mod_ld=6 * D,
                                         dshift=dmod[:, D:2 * D], dscale=dmod[:, 2 * D:3 * D], dgate=dmod[:, :D])
    # C: fc1 input gradient  dh[M,384] = du[M,1536] W1[1536,384]
    w1 = torch.randn(4 * D, D, device=dev).to(bf)
    dh = torch.empty(M, D, device=dev, dtype=bf)
    C = lambda: ops.gemm(du, w1, trans_b=True, out=dh)
    # E: attention backward
    qkv = torch.randn(M, 3 * D, device=dev).to(bf)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attention_fwd(q, k, v, B, T, T, H, d)
    do = torch.randn(M, D, device=dev).to(bf)
    dqkv = torch.empty_like(qkv)
    E = lambda: ops.attention_bwd(q, k, v, o, do, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, T, T, H, d)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    def seq(f, g):
        def run():
            f()
            g()
        return run

    def par(f, g):
        def run():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                f()
            with torch.cuda.stream(s2):
                g()
            cur.wait_stream(s1)
            cur.wait_stream(s2)
        return run

    names = {"wgrad": A, "ln_bwd": Bk, "dgrad": C, "attn_bwd": E}
    for n, f in names.items():
        print(f"{n:10s} alone {timed(f):8.1f} us")
    for a, b in (("wgrad", "ln_bwd"), ("wgrad", "dgrad"), ("wgrad", "attn_bwd"), ("dgrad", "ln_bwd")):
        print(f"{a:6s} + {b:8s}: sequential {timed(seq(names[a], names[b])):8.1f} us   two streams {timed(par(names[a], names[b])):8.1f} us")


if __name__ == "__main__":
    main()
