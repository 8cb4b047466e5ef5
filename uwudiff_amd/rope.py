"""Axial RoPE module mirroring the reference's ``duwu.modules.rope`` (src/duwu/modules/rope.py:42-108) on the HIP kernel."""
import math

import torch
import torch.nn as nn

from . import lib as L


def make_axial_pos(h, w, pixel_aspect_ratio=1.0, align_corners=False, dtype=None, device=None):
    """rope.py:10-53: cell-centre coordinates in the [-1,1] bounding box (aspect-corrected), shape [h*w, 2] (y, x)."""
    ar = w / (h * pixel_aspect_ratio)
    y_min, y_max, x_min, x_max = -1.0, 1.0, -1.0, 1.0
    if ar > 1:
        y_min, y_max = -1 / ar, 1 / ar
    elif ar < 1:
        x_min, x_max = -ar, ar

    def axis(lo, hi, n):
        if align_corners:
            return torch.linspace(lo, hi, n, dtype=dtype, device=device)
        e = torch.linspace(lo, hi, n + 1, dtype=dtype, device=device)
        return (e[:-1] + e[1:]) / 2

    g = torch.stack(torch.meshgrid(axis(y_min, y_max, h), axis(x_min, x_max, w), indexing="ij"), dim=-1)
    return g.view(h * w, 2)


class _RopeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, fh, fw, H, d):
        rows = x.shape[0]
        y = torch.empty_like(x)
        L.call("uwu_axial_rope_fwd", L.ptr(x), L.ptr(pos), L.ptr(fh), L.ptr(fw), L.ptr(y), rows, H, d, x.stride(0),
               L.dt(x), L.stream())
        ctx.save_for_backward(x, pos, fh, fw)
        ctx.meta = (H, d)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pos, fh, fw = ctx.saved_tensors
        H, d = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dfh, dfw = torch.zeros_like(fh), torch.zeros_like(fw)
        L.call("uwu_axial_rope_bwd", L.ptr(x), L.ptr(dy), L.ptr(pos), L.ptr(fh), L.ptr(fw), L.ptr(dx), L.ptr(dfh),
               L.ptr(dfw), x.shape[0], H, d, x.stride(0), L.dt(x), L.stream())
        return dx, None, dfh, dfw, None, None


class AxialRoPE(nn.Module):
    """rope.py:83-108.  ``forward(x [..., T, heads, dim], pos [..., T, 2])`` like the reference (start_index = 0)."""

    def __init__(self, dim, n_heads, start_index=0, max_freq=10.0):
        super().__init__()
        if start_index != 0:
            raise NotImplementedError("start_index != 0")
        self.n_heads, self.dim = n_heads, dim
        lf = torch.linspace(math.log(math.pi), math.log(max_freq * math.pi / 2), dim // 4).expand(n_heads, dim // 4)
        self.freqs_h = nn.Parameter(lf.clone())
        self.freqs_w = nn.Parameter(lf.clone())

    def forward(self, x, pos):
        if pos.shape[-1] != 2:
            raise ValueError("input shape must be (..., 2)")
        shp = x.shape
        H, d = shp[-2], shp[-1]
        xf = x.reshape(-1, H * d).contiguous()
        pf = pos.reshape(-1, 2).float().contiguous()
        y = _RopeFn.apply(xf, pf, self.freqs_h.float().contiguous(), self.freqs_w.float().contiguous(), H, d)
        return y.view(shp)
