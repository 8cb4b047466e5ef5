"""HBM floor for the two-output epilogues: pure writes, copy, and read-1-write-2 at the fc1 + GELU sizes.
Usage: python tools/probe_hbm.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_kernels import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M, N = B * 256, 1536
u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
both = torch.empty(2, M, N, device="cuda", dtype=torch.bfloat16)
mb = u.numel() * 2 / 1e6
us = timeit(lambda: u.fill_(1.0))
print(f"write {mb:.0f} MB: {us:.1f} us  {mb / us:.2f} TB/s")
us = timeit(lambda: both.fill_(1.0))
print(f"write {2 * mb:.0f} MB: {us:.1f} us  {2 * mb / us:.2f} TB/s")
us = timeit(lambda: h.copy_(u))
print(f"copy  {mb:.0f} -> {mb:.0f} MB: {us:.1f} us  {2 * mb / us:.2f} TB/s")
us = timeit(lambda: torch.nn.functional.gelu(u, approximate="tanh", out=None))
print(f"torch gelu (alloc) {mb:.0f} -> {mb:.0f} MB: {us:.1f} us  {2 * mb / us:.2f} TB/s")
x = torch.empty(M, 384, device="cuda", dtype=torch.bfloat16)
us = timeit(lambda: x.fill_(1.0))
print(f"write {x.numel() * 2 / 1e6:.0f} MB: {us:.1f} us  {x.numel() * 2 / 1e6 / us:.2f} TB/s")
