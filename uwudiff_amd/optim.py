"""Fused flat-buffer AdamW + global-norm clip on the HIP kernels (csrc/optimizer.hip).

Mirrors ``torch.optim.AdamW`` single-tensor semantics (the reference instantiates ``torch.optim.AdamW`` from YAML,
reference src/duwu/trainer/trainer.py:52-74, configs/demo_training_latent.yaml:30-39) and Lightning's
``gradient_clip_val`` (global L2 norm, demo_training.yaml:12).  One launch per flat parameter buffer; the bf16
shadow of the parameters (MFMA operands) is refreshed by the same kernel.
"""
import math

import torch

from . import lib as L


def _zeros_like(t):
    """torch.zeros_like for the flat buffers; on the HIP device the fill is the library's (hipMemsetAsync on the current stream)."""
    if not (t.is_cuda and t.is_contiguous()):
        return torch.zeros_like(t)
    z = torch.empty_like(t)
    L.call("uwu_memset_zero", L.ptr(z), z.numel() * z.element_size(), L.stream())
    return z


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._norm_ws = {}

    def _ws(self, device):
        w = self._norm_ws.get(device)
        if w is None:
            w = (torch.empty(1024, device=device, dtype=torch.float32), torch.ones(2, device=device, dtype=torch.float32))
            self._norm_ws[device] = w
        return w

    @torch.no_grad()
    def grad_norm_clip(self, max_norm, pre_scale=1.0):
        """Launches the global-norm reduction; returns the device tensor [sumsq, clip_coef] (no host sync).
        Only single-buffer models (one flat parameter) are clipped in one launch."""
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if len(ps) != 1:
            raise NotImplementedError("grad_norm_clip expects one flat parameter buffer")
        p = ps[0]
        part, out = self._ws(p.device)
        L.call("uwu_grad_sqnorm_clip", L.ptr(p.grad), p.numel(), float(pre_scale), float(max_norm or 0.0), L.ptr(part),
               L.ptr(out), L.stream())
        return out

    @torch.no_grad()
    def step(self, closure=None, clip=None, pre_scale=1.0, chunks=None, before_chunk=None, zero_grad=False):
        """One AdamW update.  ``chunks`` = [(offset, length), ...] splits the launch over sub-ranges of the flat
        buffer (16-byte aligned offsets); ``before_chunk(i)`` is called first (waits for chunk i's all-reduce).
        ``zero_grad``: the kernel leaves the consumed gradient zeroed (the flat gradient buffer is accumulated into by the
        next backward: no separate fill launch; ``p.grad`` stays allocated -- the ``zero_grad(set_to_none=False)`` state)."""
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = _zeros_like(p.data)
                    st["exp_avg_sq"] = _zeros_like(p.data)
                st["step"] += 1
                shadow = getattr(p, "_uwu_bf16_shadow", None)
                if shadow is not None and shadow.numel() != p.numel():
                    shadow = None
                flat = [p.data.view(-1), p.grad.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1)]
                for i, (off, ln) in enumerate(chunks or [(0, p.numel())]):
                    if before_chunk is not None:
                        before_chunk(i)
                    ptrs = [t.data_ptr() + 4 * off for t in flat]
                    sp = shadow.data_ptr() + 2 * off if shadow is not None else None
                    L.call("uwu_adamw_step", ptrs[0], ptrs[1], ptrs[2], ptrs[3], sp, ln, float(group["lr"]), b1, b2,
                           group["eps"], group["weight_decay"], st["step"], float(pre_scale),
                           L.ptr(clip) if clip is not None else None, int(bool(zero_grad)), L.stream())
        return loss


def cosine_lr(base_lr, step, T_max, eta_min):
    """Closed form of torch.optim.lr_scheduler.CosineAnnealingLR (trainer.py:111-115 defaults)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / T_max)) / 2
