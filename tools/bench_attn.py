"""Attention at T = 256, head dim 64 (the DiT shapes): interleaved A/B of the kernel variants in ONE process (HIP events).
Usage: python tools/bench_attn.py [B ...]      (H = 6: DiT-S/2; the bench's per-GPU batch is 768)"""
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
    Bs = [int(x) for x in sys.argv[1:]] or [768]
    T, H, d = 256, 6, 64
    D = H * d
    for B in Bs:
        M = B * T
        torch.manual_seed(0)
        qkv = torch.randn(M, 3 * D, device="cuda").bfloat16()
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        o, lse = ops.attention_fwd(q, k, v, B, T, T, H, d)
        do = torch.randn(M, D, device="cuda").bfloat16()
        dqkv = torch.empty_like(qkv)
        fl = 4.0 * T * T * d * B * H
        by_f, by_b = 2.0 * B * H * d * 4 * T, 2.0 * B * H * d * 8 * T
        res = {}
        for rnd in range(3):
            for name, env in (("fwd old", {"UWU_ATTN_P256F": "0"}), ("fwd p256", {"UWU_ATTN_P256F": "1"})):
                for kk, vv in env.items():
                    setflag(kk, vv)
                res.setdefault(name, []).append(timeit(lambda: ops.attention_fwd(q, k, v, B, T, T, H, d)))
            for name, env in (("bwd old", {"UWU_ATTN_P256": "0"}), ("bwd p256", {"UWU_ATTN_P256": "1"})):
                for kk, vv in env.items():
                    setflag(kk, vv)
                res.setdefault(name, []).append(timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, dqkv[:, :D], dqkv[:, D:2 * D],
                                                                                  dqkv[:, 2 * D:], B, T, T, H, d)))
        for name, us in res.items():
            f, by = (fl, by_f) if name.startswith("fwd") else (2.5 * fl, by_b)
            m = min(us)
            print(f"B={B:5d} {name:9s} min {m:8.1f} us  median {sorted(us)[len(us) // 2]:8.1f} us   {f / m / 1e6:7.1f} TFLOP/s ({f / m / 1e6 / 2500:.3f} of MFMA)"
                  f"   {by / m / 1e3:7.1f} GB/s ({by / m / 1e3 / 8000:.3f} of HBM)", flush=True)


if __name__ == "__main__":
    main()
