"""LayerNorm + modulate forward / backward at D = 384 (48 of 64 lanes carry 16-byte vectors) against D = 512 (all 64):
does lane occupancy limit the achieved HBM rate?  Usage: python tools/probe_ln_lanes.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
T = 256
M = B * T
for D in (384, 512, 256, 1024):
    x = torch.randn(M, D, device="cuda").bfloat16()
    y = torch.randn(M, D, device="cuda").bfloat16()
    mod = torch.randn(B, 6 * D, device="cuda")
    fn = lambda: ops.add_ln_modulate_fwd(x, B, T, y=y, gate=mod[:, :D], shift=mod[:, D:2 * D], scale=mod[:, 2 * D:3 * D], mod_ld=6 * D)
    us = timeit(fn)
    print(f"D={D}: fwd {us:8.1f} us  {4 * M * D * 2 / us / 1e6:5.2f} TB/s", end="   ")
    xo, h, mean, rstd = fn()
    dmod = torch.zeros(B, 6 * D, device="cuda")
    fnb = lambda: ops.add_ln_modulate_bwd(h, xo, mean, rstd, B, T, scale=mod[:, 2 * D:3 * D], dx_in=x, y=y, gate=mod[:, :D],
                                          mod_ld=6 * D, dshift=dmod[:, D:2 * D], dscale=dmod[:, 2 * D:3 * D], dgate=dmod[:, :D])
    us = timeit(fnb)
    print(f"bwd {us:8.1f} us  {6 * M * D * 2 / us / 1e6:5.2f} TB/s")
