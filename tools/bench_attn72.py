"""Attention at T = 256, head dim 72 (DiT-XL/2: 16 heads of 1152): the persistent LDS-DMA kernels with 8-column tails
(attention_p256.hip, UWU_ATTN_P256_D72 / UWU_ATTN_P256F_D72) against the kernels of attention_mfma.hip -- agreement of the
results (and an fp64 check of the first heads), then interleaved timings in ONE process.
Usage: python tools/bench_attn72.py [B ...]      (H = 16; the XL/2 bench batch is 192)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import lib as L  # noqa: E402
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def setflag(k, v):
    os.environ[k] = v
    L.load().uwu_env_refresh()


def main():
    Bs = [int(x) for x in sys.argv[1:]] or [3, 41, 192]
    T, H, d = 256, 16, 72
    D = H * d
    for B in Bs:
        M = B * T
        torch.manual_seed(B)
        qkv = torch.randn(M, 3 * D, device="cuda").bfloat16()
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        do = (torch.randn(M, D, device="cuda") * 0.5).bfloat16()
        outs = {}
        for flag in ("0", "1"):
            setflag("UWU_ATTN_P256_D72", flag)
            setflag("UWU_ATTN_P256F_D72", flag)
            setflag("UWU_ATTN_P256F", "1")  # (the forward at every head count, not only from 1024 heads)
            for rep in range(3 if flag == "1" else 1):
                o, lse = ops.attention_fwd(q, k, v, B, T, T, H, d)
                dqkv = torch.full_like(qkv, float("nan"))
                ops.attention_bwd(q, k, v, o, do, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, T, T, H, d)
                torch.cuda.synchronize()
                if flag == "1" and rep:
                    assert torch.equal(outs["1"][2], dqkv) and torch.equal(outs["1"][0], o), "launches differ"
                outs[flag] = (o, lse, dqkv)
        for i, nm in enumerate(("o", "lse", "dqkv")):
            a, b = outs["0"][i].float(), outs["1"][i].float()
            print(f"B={B} {nm}: max |new - old| = {(a - b).abs().max().item():.3e}  (max |old| {a.abs().max().item():.3e}, "
                  f"identical {torch.equal(a, b)}, finite {bool(torch.isfinite(b).all())})")
        # fp64 reference of the first two batch elements
        nb = min(B, 2)
        qr, kr, vr = (t[: nb * T].double().reshape(nb, T, H, d).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
        att = torch.softmax(qr @ kr.transpose(-1, -2) / math.sqrt(d), -1)
        orf = att @ vr
        orf.backward(do[: nb * T].double().reshape(nb, T, H, d).transpose(1, 2))
        ref = torch.cat([t.grad.transpose(1, 2).reshape(nb * T, D) for t in (qr, kr, vr)], 1)
        for flag in ("0", "1"):
            e = (outs[flag][2][: nb * T].double() - ref).abs().max().item()
            eo = (outs[flag][0][: nb * T].double() - orf.transpose(1, 2).reshape(nb * T, D)).abs().max().item()
            print(f"B={B} flag {flag}: max |dqkv - fp64| = {e:.3e}, max |o - fp64| = {eo:.3e}")
        fl = 4.0 * T * T * d * B * H
        res = {}
        os.environ.pop("UWU_ATTN_P256F")
        L.load().uwu_env_refresh()
        o, lse, dqkv = outs["1"]
        for rnd in range(3):
            for name, flag in (("fwd old", "0"), ("fwd p256", "1")):
                setflag("UWU_ATTN_P256F_D72", flag)
                res.setdefault(name, []).append(timeit(lambda: ops.attention_fwd(q, k, v, B, T, T, H, d)))
            for name, flag in (("bwd old", "0"), ("bwd p256", "1")):
                setflag("UWU_ATTN_P256_D72", flag)
                res.setdefault(name, []).append(timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, dqkv[:, :D], dqkv[:, D:2 * D],
                                                                                  dqkv[:, 2 * D:], B, T, T, H, d)))
        for name, us in res.items():
            f = fl if name.startswith("fwd") else 2.5 * fl
            m = min(us)
            print(f"B={B:5d} {name:9s} min {m:8.1f} us  median {sorted(us)[len(us) // 2]:8.1f} us   {f / m / 1e6:7.1f} TFLOP/s "
                  f"({f / m / 1e6 / 2500:.3f} of MFMA)", flush=True)


if __name__ == "__main__":
    main()
