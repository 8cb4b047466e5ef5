"""LayerNorm backward with one weight / bias vector (UNet transformer blocks), us per launch.
Usage: [UWU_LN_AFFINE=0] python tools/probe_ln_affine.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

for (M, D) in [(6144, 1280), (24576, 640), (1536, 1280), (6144, 640)]:
    x = torch.randn(M, D, device="cuda").bfloat16()
    dh = torch.randn(M, D, device="cuda").bfloat16()
    res = torch.randn(M, D, device="cuda").bfloat16()
    w = torch.randn(D, device="cuda")
    b = torch.randn(D, device="cuda")
    dw, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    _, h, mean, rstd = ops.add_ln_modulate_fwd(x, 1, M, shift=b, scale=w, mod_ld=0, eps=1e-5, affine=True)
    for dx_in in (None, res):
        us = timeit(lambda: ops.add_ln_modulate_bwd(dh, x, mean, rstd, 1, M, scale=w, dx_in=dx_in, mod_ld=0, dshift=db,
                                                    dscale=dw, affine=True))
        byt = M * D * 2 * (3 + (dx_in is not None))
        print(f"M={M} D={D} residual={dx_in is not None}: {us:7.1f} us  {byt / us / 1e6:6.2f} TB/s")
