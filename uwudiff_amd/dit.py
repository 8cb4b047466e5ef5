"""DiT (adaLN-Zero diffusion transformer) behind the reference's denoiser slot.

Call contract (reference src/duwu/loss/diffusion.py:172-176, kwargs from trainer.py:267-274):

    unet(sample [B,C,H,W], timestep [B], encoder_hidden_states=..., encoder_attention_mask=...,
         added_cond_kwargs={"time_ids": [B,6], "text_embeds": [B,Dpool]}, cross_attention_kwargs=...)[0]
      -> [B,C,H,W]

Block semantics follow the reference's ``ada_norm_zero`` branch (src/duwu/modules/rope_unet.py:306-309,
344-349, 393-411).  The network is this build's own definition (the reference ships no DiT):
Peebles & Xie sizes (S/B/L/XL = depth 12/12/24/28, width 384/768/1024/1152, heads 6/12/16/16), patch 2,
fixed 2-D sin-cos positions, tanh-GELU MLP x4, conditioning c = MLP(sinusoid(t)) + Linear(pooled text).

MI355X layout: ALL parameters live in one flat fp32 buffer (``self.flat``; 64-element aligned tensors) with a
bf16 shadow for the MFMA operands, gradients accumulate into one flat fp32 buffer (``self.flat.grad``) -> one
AdamW launch and one all-reduce per step.  Forward and backward are each a single C call into
``uwu_dit_forward`` / ``uwu_dit_backward`` (csrc/dit.cpp).
"""
import ctypes
import math
import os
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import lib as L

PRESETS = {
    "DiT-S/2": dict(depth=12, hidden=384, heads=6, patch=2),
    "DiT-S/2-RoPE": dict(depth=12, hidden=384, heads=6, patch=2, rope=True),   # RoPE variant (rope_unet.py's attention)
    "DiT-B/2-RoPE": dict(depth=12, hidden=768, heads=12, patch=2, rope=True),
    "DiT-B/2": dict(depth=12, hidden=768, heads=12, patch=2),
    "DiT-L/2": dict(depth=24, hidden=1024, heads=16, patch=2),
    "DiT-XL/2": dict(depth=28, hidden=1152, heads=16, patch=2),
}


@dataclass
class DiTConfig:
    depth: int = 12
    hidden: int = 384
    heads: int = 6
    patch: int = 2
    sample_size: int = 32
    in_channels: int = 4
    out_channels: int = 4
    mlp_ratio: int = 4
    cond_dim: int = 0          # pooled-text width (SDXL: 1280); 0 = unconditional
    freq_dim: int = 256
    ln_eps: float = 1e-6
    compute_dtype: str = "bf16"  # "bf16" | "fp32" | "fp8" (bf16 activations, fp8 e4m3 / e5m2 operands of the block Linears)
    rope: bool = False           # learnable axial RoPE on q / k of every block (reference modules/rope.py, rope_unet.py:143-147)
    fp8_scaling: str = "delayed"  # "delayed" (amax of the previous step; the first step scales just in time) | "jit"


def _pad64(n):
    return (n + 63) // 64 * 64


def sincos_2d(dim, grid):
    """Fixed 2-D sin-cos position table [grid*grid, dim] (MAE/DiT construction: half the channels per axis)."""
    def one(d, pos):
        omega = 1.0 / (10000 ** (torch.arange(d // 2, dtype=torch.float64) / (d / 2)))
        out = pos.reshape(-1, 1).double() * omega[None]
        return torch.cat([out.sin(), out.cos()], dim=1)

    gh, gw = torch.meshgrid(torch.arange(grid), torch.arange(grid), indexing="ij")
    emb = torch.cat([one(dim // 2, gw), one(dim // 2, gh)], dim=1)  # w first, as get_2d_sincos_pos_embed
    return emb.float()


class _DiTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, model, noisy, t, cond):
        out = model._run_forward(noisy, t, cond)
        ctx.model = model
        ctx.gen = model._fwd_gen
        ctx.cond = cond
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        if ctx.gen != model._fwd_gen:
            raise RuntimeError("DiT.backward: the activation workspace was overwritten by a later forward "
                               "(one outstanding forward per backward)")
        model._run_backward(dout.float().contiguous(), ctx.cond)
        return None, None, None, None, None


class DiT(nn.Module):
    def __init__(self, config: DiTConfig = None, init: str = "dit", **kw):
        super().__init__()
        if config is None:
            config = DiTConfig(**kw)
        self.cfg = c = config
        assert c.hidden % c.heads == 0 and c.sample_size % c.patch == 0
        self.T = (c.sample_size // c.patch) ** 2
        D, r = c.hidden, c.mlp_ratio
        kp, ko = c.in_channels * c.patch ** 2, c.out_channels * c.patch ** 2
        self.mod_total = c.depth * 6 * D + 2 * D
        spec = [("x_embedder.weight", (D, kp)), ("x_embedder.bias", (D,)),
                ("t_embedder.0.weight", (D, c.freq_dim)), ("t_embedder.0.bias", (D,)),
                ("t_embedder.2.weight", (D, D)), ("t_embedder.2.bias", (D,)),
                ("y_embedder.weight", (D, max(c.cond_dim, 8))), ("y_embedder.bias", (D,)),
                ("adaLN.weight", (self.mod_total, D)), ("adaLN.bias", (self.mod_total,)),
                ("final.weight", (ko, D)), ("final.bias", (ko,))]
        if c.rope:  # per-layer, per-head log-frequencies of the two axes (AxialRoPE(head_dim, heads): [heads, head_dim / 4] each)
            hd4 = D // c.heads // 4
            spec += [("rope.freqs_h", (c.depth, c.heads, hd4)), ("rope.freqs_w", (c.depth, c.heads, hd4))]
        for l in range(c.depth):
            spec += [(f"blocks.{l}.qkv.weight", (3 * D, D)), (f"blocks.{l}.qkv.bias", (3 * D,)),
                     (f"blocks.{l}.proj.weight", (D, D)), (f"blocks.{l}.proj.bias", (D,)),
                     (f"blocks.{l}.fc1.weight", (r * D, D)), (f"blocks.{l}.fc1.bias", (r * D,)),
                     (f"blocks.{l}.fc2.weight", (D, r * D)), (f"blocks.{l}.fc2.bias", (D,))]
        self.registry = {}
        off = 0
        for name, shape in spec:
            n = math.prod(shape)
            self.registry[name] = (off, shape)
            off += _pad64(n)
        self.n_flat = off
        self.flat = nn.Parameter(torch.zeros(off, dtype=torch.float32))
        self.register_buffer("pos", sincos_2d(D, c.sample_size // c.patch), persistent=False)
        self.register_buffer("shadow", torch.zeros(0, dtype=torch.bfloat16), persistent=False)
        if c.rope:
            from .rope import make_axial_pos

            g = c.sample_size // c.patch
            self.register_buffer("pos_xy", make_axial_pos(g, g).float().contiguous(), persistent=False)
        self._ws = None
        self._ws_key = None
        self._layer_done = None
        self._layer_events = None
        self._grad_hook = None
        self._grad_groups = None
        self._fwd_gen = 0
        self._desc = None
        self._side = None      # second HIP stream: weight gradients of small batches run beside the input-gradient chain
        self._f8 = None        # (scale, amax, fmt) device tensors of the fp8 mode, 12 roles per block
        self._f8_steps = 0     # forward passes taken in fp8 mode (the first one always scales just in time)
        self._ckpt = False     # block recomputation (enable_gradient_checkpointing)
        self.config = type("cfg", (), dict(in_channels=c.in_channels, sample_size=c.sample_size))()
        self.reset_parameters(init)

    # ------------------------------------------------------------------ parameters
    def view(self, name):
        off, shape = self.registry[name]
        return self.flat.data[off:off + math.prod(shape)].view(shape)

    def grad_view(self, name):
        off, shape = self.registry[name]
        return self.flat.grad[off:off + math.prod(shape)].view(shape)

    @torch.no_grad()
    def reset_parameters(self, init="dit"):
        """init="dit": DiT paper (xavier linears, zero adaLN / final) + the reference's near-zero N(0,1e-5)
        residual-out rule (src/duwu/modules/unet_patch.py:34-45).  init="random": N(0,0.02) everywhere
        (non-zero gates; used by parity tests and the benchmark so no branch is numerically dead)."""
        self.flat.data.zero_()
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        for name, (off, shape) in self.registry.items():
            v = self.view(name)
            if init == "random" and not name.startswith("rope."):
                v.copy_(torch.randn(shape, generator=g) * 0.02)
                continue
            if name.startswith("rope."):  # reference rope.py:74-81 freqs_pixel_log(max_freq=10): linspace(log pi, log 5 pi)
                n = shape[-1]
                v.copy_(torch.linspace(math.log(math.pi), math.log(10.0 * math.pi / 2), n).expand(shape))
                continue
            if name.endswith("bias"):
                continue
            if name.startswith(("adaLN", "final")):
                continue
            if name.endswith(("proj.weight", "fc2.weight")):
                v.copy_(torch.randn(shape, generator=g) * 1e-5)
            elif name.startswith("t_embedder"):
                v.copy_(torch.randn(shape, generator=g) * 0.02)
            else:
                bound = math.sqrt(6.0 / (shape[0] + shape[1]))
                v.copy_((torch.rand(shape, generator=g) * 2 - 1) * bound)
        self.refresh_shadow()

    def named_tensors(self):
        for name in self.registry:
            if name.startswith("y_embedder") and self.cfg.cond_dim == 0:
                continue
            yield name, self.view(name)

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = destination if destination is not None else {}
        for name, v in self.named_tensors():
            sd[prefix + name] = v if keep_vars else v.detach().clone()
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict=True, assign=False):
        missing = []
        for name, v in self.named_tensors():
            if name in state_dict:
                v.copy_(state_dict[name].to(v))
            else:
                missing.append(name)
        if strict and missing:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing}")
        self.refresh_shadow()
        return torch.nn.modules.module._IncompatibleKeys(missing, [])

    @torch.no_grad()
    def refresh_shadow(self):
        """bf16 copy of the flat parameters for the MFMA operands (kept fresh by the fused AdamW afterwards)."""
        if self.cfg.compute_dtype not in ("bf16", "fp8") or not self.flat.is_cuda:
            return
        if self.shadow.numel() != self.n_flat or self.shadow.device != self.flat.device:
            self.shadow = torch.empty(self.n_flat, device=self.flat.device, dtype=torch.bfloat16)
        L.call("uwu_cast_f32_to_bf16", L.ptr(self.flat.data), L.ptr(self.shadow), self.n_flat, L.stream())
        self.flat._uwu_bf16_shadow = self.shadow

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self._desc = None
        self.refresh_shadow()
        return r

    def enable_gradient_checkpointing(self, enabled=True):
        """Reference test_scripts/test_train.py:38-39 (diffusers checkpoints per transformer block, rope_unet.py:484-507).
        The forward then keeps per block only its input and the statistics of its first LayerNorm; the C++ driver reruns
        block l - 1 inside the backward (``uwu_dit_desc.checkpoint``).  Activation memory drops from ``depth`` block slabs
        to one (DiT-S/2, batch 768: 33 GB -> 4.6 GB) for one extra forward of the blocks."""
        self._ckpt = bool(enabled)
        self._ws_key = None

    # ------------------------------------------------------------------ descriptor / workspace
    def _descriptor(self, B):
        c = self.cfg
        if not self.flat.is_cuda:
            raise L.UwuError("DiT runs on the HIP device only; move the module with .cuda() (no CPU fallback)")
        bf = c.compute_dtype in ("bf16", "fp8")
        if bf and self.shadow.numel() != self.n_flat:
            self.refresh_shadow()
        d = L.DitDesc()
        d.B, d.T, d.D, d.H, d.L = B, self.T, c.hidden, c.heads, c.depth
        d.mlp_ratio, d.in_ch, d.out_ch, d.patch, d.img = c.mlp_ratio, c.in_channels, c.out_channels, c.patch, c.sample_size
        d.dtype = L.BF16 if bf else L.F32
        d.cond_dim, d.freq_dim, d.ln_eps, d.mod_total = c.cond_dim, c.freq_dim, c.ln_eps, self.mod_total
        d.w = (self.shadow if bf else self.flat.data).data_ptr()
        d.w32 = self.flat.data.data_ptr()
        d.g32 = self.flat.grad.data_ptr() if self.flat.grad is not None else None
        r = self.registry
        d.off_patch_w, d.off_patch_b = r["x_embedder.weight"][0], r["x_embedder.bias"][0]
        d.off_t_w1, d.off_t_b1 = r["t_embedder.0.weight"][0], r["t_embedder.0.bias"][0]
        d.off_t_w2, d.off_t_b2 = r["t_embedder.2.weight"][0], r["t_embedder.2.bias"][0]
        d.off_y_w, d.off_y_b = r["y_embedder.weight"][0], r["y_embedder.bias"][0]
        d.off_mod_w, d.off_mod_b = r["adaLN.weight"][0], r["adaLN.bias"][0]
        d.off_final_w, d.off_final_b = r["final.weight"][0], r["final.bias"][0]
        d.off_layer0 = r["blocks.0.qkv.weight"][0]
        d.layer_stride = L.load().uwu_dit_layer_param_stride(c.hidden, c.mlp_ratio)
        if c.depth > 1:
            assert r["blocks.1.qkv.weight"][0] - d.off_layer0 == d.layer_stride
        d.pos = self.pos.data_ptr()
        d.side_stream = None
        if bf and B * self.T <= 16384 and os.environ.get("UWU_DIT_FORK", "1") != "0":
            if self._side is None or self._side.device != self.flat.device:
                self._side = torch.cuda.Stream(device=self.flat.device)
            d.side_stream = self._side.cuda_stream
        d.rope = 1 if c.rope else 0
        if c.rope:
            d.off_rope_h, d.off_rope_w = r["rope.freqs_h"][0], r["rope.freqs_w"][0]
            d.pos_xy = self.pos_xy.data_ptr()
        d.fp8 = 0
        if c.compute_dtype == "fp8":
            if self._f8 is None or self._f8[0].device != self.flat.device:
                n = 12 * c.depth
                fmt = torch.tensor(([0] * 4 + [1] * 4 + [0] * 4) * c.depth, dtype=torch.int32, device=self.flat.device)
                self._f8 = (torch.zeros(n, device=self.flat.device), torch.zeros(n, device=self.flat.device), fmt)
                self._f8_steps = 0
            d.fp8 = 1 if (c.fp8_scaling == "jit" or self._f8_steps == 0) else 2
            d.f8_scale, d.f8_amax, d.f8_fmt = (t.data_ptr() for t in self._f8)
        d.checkpoint = 1 if self._ckpt else 0
        key = (B, d.dtype, d.fp8 != 0, self.flat.device, d.checkpoint)
        if self._ws_key != key:
            d.ws, d.ws_bytes = None, 0
            need = L.load().uwu_dit_workspace_bytes(ctypes.byref(d))
            self._ws = torch.empty(need, device=self.flat.device, dtype=torch.uint8)
            self._ws_key = key
        d.ws, d.ws_bytes = self._ws.data_ptr(), self._ws.numel()
        d.layer_done = ctypes.addressof(self._layer_done) if self._layer_done is not None else None
        return d

    # ------------------------------------------------------------------ data-parallel hook
    def set_grad_ready_hook(self, hook, group_layers=4):
        """``hook(flat_grad, [(offset, length, event), ...])`` is called inside backward, right after the C++ driver has
        enqueued the whole pass: each entry is a contiguous slice of ``flat.grad`` (a group of transformer blocks,
        last group first) and the ``torch.cuda.Event`` the driver records once every gradient launch of that group is
        in flight (``uwu_dit_desc.layer_done``).  A data-parallel host reduces those slices on its communication
        stream while the rest of the backward still runs (uwudiff_amd/gradsync.py).  ``hook=None`` switches it off."""
        self._grad_hook = hook
        self._grad_groups = None
        if hook is None:
            self._layer_done = None
            self._layer_events = None
            return
        if not self.flat.is_cuda:
            raise L.UwuError("set_grad_ready_hook: move the module to the HIP device first")
        depth = self.cfg.depth
        self._layer_events = []
        with torch.cuda.device(self.flat.device):
            for _ in range(depth):
                ev = torch.cuda.Event()
                ev.record()  # materialises the hipEvent_t behind the torch object
                self._layer_events.append(ev)
        self._layer_done = (ctypes.c_void_p * depth)(*[ev.cuda_event for ev in self._layer_events])
        off0 = self.registry["blocks.0.qkv.weight"][0]
        stride = L.load().uwu_dit_layer_param_stride(self.cfg.hidden, self.cfg.mlp_ratio)
        groups = []
        hi = depth
        while hi > 0:  # blocks finish in the order depth-1 .. 0: the lowest block of a group is its last
            lo = max(0, hi - group_layers)
            groups.append((off0 + lo * stride, (hi - lo) * stride, self._layer_events[lo]))
            hi = lo
        assert off0 + depth * stride == self.n_flat, "transformer blocks must be the tail of the flat buffer"
        self._grad_groups = groups

    def _run_forward(self, noisy, t, cond):
        B = noisy.shape[0]
        d = self._descriptor(B)
        out = torch.empty(B, self.cfg.out_channels, self.cfg.sample_size, self.cfg.sample_size, device=noisy.device,
                          dtype=torch.float32)
        L.call("uwu_dit_forward", ctypes.byref(d), L.ptr(noisy), L.ptr(t), L.ptr(cond), L.ptr(out), L.stream())
        self._fwd_gen += 1
        self._f8_mode = d.fp8
        if d.fp8:
            self._f8_steps += 1
        return out

    def _run_backward(self, dout, cond):
        if self.flat.grad is None:
            self.flat.grad = torch.empty_like(self.flat.data)
            L.call("uwu_memset_zero", L.ptr(self.flat.grad), self.flat.grad.numel() * 4, L.stream())
        d = self._descriptor(dout.shape[0])
        d.fp8 = getattr(self, "_f8_mode", 0)  # the backward uses the scaling policy its forward used
        L.call("uwu_dit_backward", ctypes.byref(d), L.ptr(dout), L.stream())
        if self._grad_hook is not None:
            self._grad_hook(self.flat.grad, self._grad_groups)
        if cond is not None:
            L.call("uwu_dit_backward_cond", ctypes.byref(d), L.ptr(cond), L.stream())

    # ------------------------------------------------------------------ denoiser slot
    def forward(self, sample, timestep, encoder_hidden_states=None, encoder_attention_mask=None,
                added_cond_kwargs=None, cross_attention_kwargs=None, **kw):
        c = self.cfg
        B = sample.shape[0]
        noisy = sample.float().contiguous()
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=sample.device)
        if timestep.dtype == torch.int64 and timestep.is_cuda and timestep.device == sample.device and timestep.is_contiguous():
            t = torch.empty(timestep.shape, device=sample.device, dtype=torch.float32)  # (the loss draws int64 timesteps)
            L.call("uwu_cast_i64_to_f32", L.ptr(timestep), L.ptr(t), timestep.numel(), L.stream())
            t = t.reshape(-1)
        else:
            t = timestep.to(device=sample.device, dtype=torch.float32).reshape(-1)
        if t.numel() == 1 and B > 1:
            t = t.expand(B)
        t = t.contiguous()
        cond = None
        if c.cond_dim > 0:
            pooled = (added_cond_kwargs or {}).get("text_embeds")
            if pooled is None:
                cond = torch.zeros(B, c.cond_dim, device=sample.device, dtype=torch.float32)
            else:
                cond = pooled.to(device=sample.device, dtype=torch.float32).contiguous()
                if cond.shape != (B, c.cond_dim):
                    raise ValueError(f"text_embeds must be [{B},{c.cond_dim}], got {tuple(cond.shape)}")
        out = _DiTFn.apply(self.flat, self, noisy, t, cond)
        return (out,)

    @classmethod
    def from_config(cls, config, **kw):
        """``_target_: ...DiT.from_config`` with ``config: "DiT-S/2"`` (preset name) or a dict."""
        init = kw.pop("init", "dit")
        if isinstance(config, str):
            if config not in PRESETS:
                raise ValueError(f"unknown DiT preset {config!r}; known: {sorted(PRESETS)}")
            cfg = dict(PRESETS[config])
        else:
            cfg = dict(config)
        cfg.update(kw)
        cfg.pop("subfolder", None)
        return cls(DiTConfig(**cfg), init=init)
