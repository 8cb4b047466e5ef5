"""Attention backward at T = 256: the one-kernel form against the key-block + query-owner pair (taken with a key bias).
Usage: python tools/probe_attn_paths.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, D, H, d = 256, 384, 6, 64
M = B * T
qkv = torch.randn(M, 3 * D, device="cuda").bfloat16()
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
kb = torch.zeros(B, T, device="cuda")
o, lse = ops.attention_fwd(q, k, v, B, T, T, H, d)
do = torch.randn(M, D, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
for name, bias in (("one-kernel", None), ("two-kernel (bias path)", kb)):
    us = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, T, T, H, d, key_bias=bias))
    print(f"attn_bwd {name}: {us:.1f} us")
    us = timeit(lambda: ops.attention_fwd(q, k, v, B, T, T, H, d, key_bias=bias))
    print(f"attn_fwd {name}: {us:.1f} us")
