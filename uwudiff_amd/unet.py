"""UNet2DConditionModel-shape denoiser on the HIP kernels.

The reference's denoiser is ``UNet2DFromScratch`` = ``diffusers.UNet2DConditionModel`` + a near-zero init of the
residual-branch output layers (reference src/duwu/modules/unet_patch.py:13-57); both shipped configs build the
SDXL shape.  This module keeps the call contract (diffusion.py:172-176), diffusers' parameter names in
``state_dict()`` and the init rule, and runs every operator through ``libuwu_hip.so``:

  * activations channels-last / token-major ``[B*H*W, C]`` (bf16 in bf16 mode): Linear, attention and LayerNorm
    consume them as they are; a 3x3 convolution (C and Cout multiples of 32, bf16) is an IMPLICIT GEMM -- the ring
    kernels gather their activation operand per (tap, 32-channel chunk) with LDS-DMA, forward, input gradient (flipped
    taps) and weight gradient alike; no im2col matrix exists in HBM.  conv_in (4 channels) and the
    fp32 parity mode keep ``uwu_im2col3x3`` + GEMM / ``uwu_col2im3x3``;
  * GroupNorm(+SiLU), affine LayerNorm, GEGLU, nearest upsample, time-embedding broadcast-add are dedicated
    kernels; self- and cross-attention (77 context tokens, optional key bias) run on the MFMA flash kernels;
  * all parameters live in one flat fp32 buffer (+ bf16 shadow) and gradients accumulate into ``flat.grad`` from
    inside the kernels, so the optimizer and the data-parallel exchange are the same single launches as for DiT.

The graph is composed in Python (one ``autograd.Function`` per fused op, ~4000 launches per step); the host enqueues a
step in about a fifth of the time the GPU needs for it (bench line ``host_enqueue_ms``), so this is not the limiter.
"""
import collections
import math
import os

import torch
import torch.nn as nn

from . import lib as L
from . import ops

SDXL_UNET_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
    addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
    norm_num_groups=32, sample_size=128,
)
# BASELINE.json configs[0]: "tiny-UNet 32x32x3 pixel diffusion" (build-defined plumbing model)
TINY_UNET_CONFIG = dict(
    in_channels=3, out_channels=3, block_out_channels=(64, 128), layers_per_block=1,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"), up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"),
    transformer_layers_per_block=(1, 1), attention_head_dim=(1, 2), cross_attention_dim=2048,
    addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
    norm_num_groups=32, sample_size=32,
)


def _pad8(c):
    return (c + 7) // 8 * 8


def _pad64(n):
    return (n + 63) // 64 * 64


class _Ctx:
    """Shared state of one model instance: flat parameters, shadow, gradient views, compute dtype."""

    def __init__(self):
        self.registry = {}
        self.n = 0
        self.flat = None
        self.shadow = None
        self.bf16 = True
        self.side = None
        self._join_queued = False
        self._held = collections.deque()

    # ---- weight gradients are off the backward's critical path: they run on a side stream, ordered after the kernels
    # that produced their operands; the main stream joins once, when the backward pass has finished
    def on_side(self, fn, *operands):
        side = self.side
        if side is None:
            return fn()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
            done = torch.cuda.Event()
            done.record(side)
        # The operands stay referenced until the side stream has read them: the allocator cannot hand their memory out, and
        # autograd -- which sums a second gradient INTO a tensor it holds the only reference to -- cannot rewrite them
        # (an output gradient returned to two producers was overwritten on the main stream while its wgrad was reading it).
        self._held.append((done, operands))
        while self._held and self._held[0][0].query():
            self._held.popleft()
        # back-pressure: the main stream never waits for the side stream inside the backward, so at large batches the weight
        # gradients fall behind and their operands pile up (batch 48 with recomputation ran out of 288 GB this way)
        while len(self._held) > 16:
            self._held.popleft()[0].synchronize()
        if not self._join_queued:
            self._join_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._join)

    def _join(self):
        self._join_queued = False
        torch.cuda.current_stream().wait_stream(self.side)
        # everything the main stream launches from here on is ordered behind the side stream, so the operands the queued
        # weight gradients read can go back to the allocator now (they were held across the next forward before: several GB)
        self._held.clear()

    def add(self, name, shape):
        self.registry[name] = (self.n, tuple(shape))
        self.n += _pad64(math.prod(shape))

    def _view(self, buf, name):
        if isinstance(name, tuple):  # (first name, n): n equally shaped matrices stored back to back, seen as one
            first, n = name
            off, (rows, cols) = self.registry[first]
            return buf[off:off + n * rows * cols].view(n * rows, cols)
        off, shape = self.registry[name]
        return buf[off:off + math.prod(shape)].view(shape)

    def span(self, names):
        """(first, n) when `names` sit back to back with no padding between them (one stacked GEMM operand), else None"""
        off, shape = self.registry[names[0]]
        for i, nm in enumerate(names):
            o, sh = self.registry[nm]
            if sh != shape or len(sh) != 2 or o != off + i * math.prod(shape):
                return None
        return (names[0], len(names))

    def w32(self, name):
        return self._view(self.flat.data, name)

    def w(self, name):  # GEMM operand copy
        return self._view(self.shadow if self.bf16 else self.flat.data, name)

    def g(self, name):
        if self.flat.grad is None:
            self.flat.grad = torch.zeros_like(self.flat.data)
        return self._view(self.flat.grad, name)

    @property
    def dtype(self):
        return torch.bfloat16 if self.bf16 else torch.float32


def _wgrad_blocks(M, N, K):
    """workgroups the generic split-K weight-gradient kernel aims for (M tokens, N x K weight): ~3 per CU for large
    weights, ~1 per CU when only a few output tiles exist (uwu_gemm_wgrad's fallback path; the streaming kernel picks
    its own number of K slices)."""
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    return 768 if tiles >= 16 else 256


_SKINNY = os.environ.get("UWU_UNET_SKINNY", "1") != "0"  # A/B switch: the generic fp32 GEMM for the few-row Linears


class _LinearFn(torch.autograd.Function):
    """y = x W^T (+ b); W is a [N, K] view of the flat parameters, grads accumulate into the flat grad buffer."""

    @staticmethod
    def forward(ctx, x, P, wname, bname, fp32, anchor):
        # `anchor` (the flat parameter) keeps the op in the autograd graph when x itself needs no gradient
        # (sinusoid features, text context); needs_input_grad[0] then stays False and the dgrad GEMM is skipped
        W = P.w32(wname) if fp32 else P.w(wname)
        b = P.w32(bname) if bname else None
        # fp32 Linears on a handful of rows (time / text-time embedding MLPs, the resblocks' time_emb_proj: M = batch): the
        # matrix-vector kernels of the conditioning path (csrc/skinny.hip) instead of a 128-row MFMA tile with 12 live rows
        skinny = (fp32 and x.dtype == torch.float32 and x.is_contiguous() and x.shape[0] <= 64 and _SKINNY
                  and ops.skinny_linear_ok(x.shape[0], W.shape[0], x.shape[1]))
        if skinny:
            y = ops.skinny_linear_fwd(x, W, b)
        else:
            y = ops.gemm(x, W, bias=b, epilogue=L.EPI_BIAS if b is not None else L.EPI_NONE)
        ctx.save_for_backward(x)
        ctx.meta = (P, wname, bname, fp32, skinny)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        P, wname, bname, fp32, skinny = ctx.meta
        dy = dy.contiguous()
        W = P.w32(wname) if fp32 else P.w(wname)
        M, K = x.shape
        N = W.shape[0]
        if skinny and dy.dtype == torch.float32:
            dx = None
            if ctx.needs_input_grad[0]:  # (the matrix-vector input gradient covers K <= 512: the DiT widths, not 1280)
                dx = ops.skinny_linear_dgrad(dy, W) if K <= 512 else ops.gemm(dy, W, trans_b=True)
            P.on_side(lambda: ops.skinny_linear_wgrad(dy, x, P.g(wname), P.g(bname) if bname else None), dy, x)
            return dx, None, None, None, None, None
        dx = ops.gemm(dy, W, trans_b=True) if ctx.needs_input_grad[0] else None
        # dW += dy^T x and db += colsum(dy) in one launch where the streaming kernel takes the shape
        P.on_side(lambda: ops.gemm_wgrad_shared(dy, x, P.g(wname), blocks=_wgrad_blocks(M, N, K),
                                                bias_grad=P.g(bname) if bname else None), dy, x)
        return dx, None, None, None, None, None


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, P, wname, bname, B, H, W_, C, stride):
        Wt = P.w(wname)
        # implicit GEMM (the ring kernel gathers its activation rows pixel by pixel; no column matrix) where the shape
        # allows it: bf16, channel counts that are multiples of 32 -- every convolution of the SDXL shape except conv_in
        implicit = ops.conv3x3_implicit_ok(x, B, H, W_, C, Wt.shape[0], stride)
        if implicit:
            y = ops.conv3x3_fwd(x, Wt, P.w32(bname), B, H, W_, C, Wt.shape[0], stride)
        else:
            col = ops.im2col3x3(x, B, H, W_, C, stride)
            y = ops.gemm(col, Wt, bias=P.w32(bname), epilogue=L.EPI_BIAS)
        ctx.save_for_backward(x)
        ctx.meta = (P, wname, bname, B, H, W_, C, stride, implicit)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        P, wname, bname, B, H, W_, C, stride, implicit = ctx.meta
        dy = dy.contiguous()
        Wt = P.w(wname)
        dx = None
        if implicit:
            Cout = Wt.shape[0]
            if ctx.needs_input_grad[0]:
                dx = ops.conv3x3_dgrad(dy, Wt, B, H, W_, C, Cout, stride)
            P.on_side(lambda: ops.conv3x3_wgrad(dy, x, P.g(wname), P.g(bname), B, H, W_, C, Cout, stride), dy, x)
            return (dx,) + (None,) * 8
        if ctx.needs_input_grad[0]:
            dcol = ops.gemm(dy, Wt, trans_b=True)
            dx = ops.col2im3x3(dcol, B, H, W_, C, stride)
        col = ops.im2col3x3(x, B, H, W_, C, stride)  # recomputed: cheaper than keeping 9x the activation
        ops.gemm_wgrad(dy, col, P.g(wname), blocks=_wgrad_blocks(col.shape[0], Wt.shape[0], col.shape[1]),
                       bias_grad=P.g(bname))
        return (dx,) + (None,) * 8


class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, P, prefix, B, HW, C, G, eps, silu):
        y, mean, rstd = ops.groupnorm_fwd(x, P.w32(prefix + ".weight"), P.w32(prefix + ".bias"), B, HW, C, G, eps, silu)
        ctx.save_for_backward(x, mean, rstd)
        ctx.meta = (P, prefix, B, HW, C, G, silu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        P, prefix, B, HW, C, G, silu = ctx.meta
        dx = ops.groupnorm_bwd(dy.contiguous(), x, mean, rstd, P.w32(prefix + ".weight"), P.w32(prefix + ".bias"),
                               P.g(prefix + ".weight"), P.g(prefix + ".bias"), B, HW, C, G, silu)
        return (dx,) + (None,) * 8


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, P, prefix, eps):
        M, D = x.shape
        _, h, mean, rstd = ops.add_ln_modulate_fwd(x, 1, M, shift=P.w32(prefix + ".bias"), scale=P.w32(prefix + ".weight"),
                                                   mod_ld=0, eps=eps, affine=True)
        ctx.save_for_backward(x, mean, rstd)
        ctx.meta = (P, prefix)
        return h

    @staticmethod
    def backward(ctx, dh):
        x, mean, rstd = ctx.saved_tensors
        P, prefix = ctx.meta
        M, D = x.shape
        dx, _ = ops.add_ln_modulate_bwd(dh.contiguous(), x, mean, rstd, 1, M, scale=P.w32(prefix + ".weight"), mod_ld=0,
                                        dshift=P.g(prefix + ".bias"), dscale=P.g(prefix + ".weight"), affine=True)
        return dx, None, None, None


class _AddLayerNormFn(torch.autograd.Function):
    """(x, n) = (h + a, LN(h + a)) in one pass; backward folds the residual gradient into the LayerNorm backward
    (dx = dx_res + LN_bwd(dn)), so neither direction runs a separate add."""

    @staticmethod
    def forward(ctx, h, a, P, prefix, eps, ones):
        M, D = h.shape
        x, n, mean, rstd = ops.add_ln_modulate_fwd(h, 1, M, y=a.contiguous(), gate=ones, shift=P.w32(prefix + ".bias"),
                                                   scale=P.w32(prefix + ".weight"), mod_ld=0, eps=eps, affine=True)
        ctx.save_for_backward(x, mean, rstd)
        ctx.meta = (P, prefix)
        return x, n

    @staticmethod
    def backward(ctx, dx_res, dn):
        x, mean, rstd = ctx.saved_tensors
        P, prefix = ctx.meta
        M, D = x.shape
        dx, _ = ops.add_ln_modulate_bwd(dn.contiguous(), x, mean, rstd, 1, M, scale=P.w32(prefix + ".weight"),
                                        dx_in=dx_res.contiguous(), mod_ld=0, dshift=P.g(prefix + ".bias"),
                                        dscale=P.g(prefix + ".weight"), affine=True)
        return dx, dx, None, None, None, None


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, B, Tq, Tk, H, d, key_bias=None):
        o, lse = ops.attention_fwd(q, k, v, B, Tq, Tk, H, d, key_bias=key_bias)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.meta = (B, Tq, Tk, H, d, key_bias)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        B, Tq, Tk, H, d, key_bias = ctx.meta
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ops.attention_bwd(q, k, v, o, do.contiguous(), lse, dq, dk, dv, B, Tq, Tk, H, d, key_bias=key_bias)
        return dq, dk, dv, None, None, None, None, None, None


class _PackedAttentionFn(torch.autograd.Function):
    """attention over stacked projections: `a` is [M, 3D] (q|k|v, self-attention) or q comes separately and `a` is the
    context's [Mk, 2D] (k|v).  The gradient of the stacked tensor is written in place by the backward kernel, so the
    three (two) projections cost one dgrad and one wgrad GEMM and no gradient summation."""

    @staticmethod
    def forward(ctx, q, a, B, Tq, Tk, H, d, key_bias=None):
        D = H * d
        if q is None:
            qv, kv, vv = a[:, :D], a[:, D:2 * D], a[:, 2 * D:]
        else:
            qv, kv, vv = q, a[:, :D], a[:, D:]
        o, lse = ops.attention_fwd(qv, kv, vv, B, Tq, Tk, H, d, key_bias=key_bias)
        ctx.save_for_backward(q, a, o, lse)
        ctx.meta = (B, Tq, Tk, H, d, key_bias)
        return o

    @staticmethod
    def backward(ctx, do):
        q, a, o, lse = ctx.saved_tensors
        B, Tq, Tk, H, d, key_bias = ctx.meta
        D = H * d
        da = torch.empty_like(a)
        if q is None:
            dq = None
            qv, kv, vv = a[:, :D], a[:, D:2 * D], a[:, 2 * D:]
            dqv, dkv, dvv = da[:, :D], da[:, D:2 * D], da[:, 2 * D:]
        else:
            dq = torch.empty_like(q)
            qv, kv, vv = q, a[:, :D], a[:, D:]
            dqv, dkv, dvv = dq, da[:, :D], da[:, D:]
        ops.attention_bwd(qv, kv, vv, o, do.contiguous(), lse, dqv, dkv, dvv, B, Tq, Tk, H, d, key_bias=key_bias)
        return dq, da, None, None, None, None, None, None


class _FlatParamFn(torch.autograd.Function):
    """a named fp32 tensor of the flat parameter buffer as an autograd leaf-like input of an op written for ordinary
    tensors (the axial-RoPE function): its gradient is ADDED into the same slice of ``flat.grad``"""

    @staticmethod
    def forward(ctx, anchor, P, name):
        ctx.P, ctx.name = P, name
        return P.w32(name).clone()

    @staticmethod
    def backward(ctx, d):
        ctx.P.g(ctx.name).add_(d.float().view_as(ctx.P.g(ctx.name)))
        return None, None, None


class _GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hg):
        ctx.save_for_backward(hg)
        return ops.geglu_fwd(hg)

    @staticmethod
    def backward(ctx, dout):
        (hg,) = ctx.saved_tensors
        return ops.geglu_bwd(hg, dout.contiguous())


class _SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.silu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.silu_bwd(x, dy.contiguous())


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, d):
        return d, d


class _AddRowvecFn(torch.autograd.Function):
    """x[b, p, :] + v[b, :]   (ResnetBlock2D time-embedding injection)."""

    @staticmethod
    def forward(ctx, x, v, B, HW, C):
        ctx.meta = (B, HW, C, v.dtype)
        return ops.add_rowvec(x, v.to(x.dtype), B, HW, C)

    @staticmethod
    def backward(ctx, d):
        B, HW, C, vdt = ctx.meta
        d = d.contiguous()
        dv = ops.colsum_batched(d, B, HW, C)
        return d, dv.to(vdt), None, None, None


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B, H, W, C):
        ctx.meta = (B, H, W, C)
        return ops.upsample2x(x, B, H, W, C, backward=False)

    @staticmethod
    def backward(ctx, d):
        B, H, W, C = ctx.meta
        return ops.upsample2x(d.contiguous(), B, H, W, C, backward=True), None, None, None, None


class _ToCL(torch.autograd.Function):
    """NCHW fp32 -> channels-last tokens (channels zero-padded to `cpad`)."""

    @staticmethod
    def forward(ctx, x, cpad, dtype):
        B, C, H, W = x.shape
        if cpad != C:
            x = torch.cat([x, x.new_zeros(B, cpad - C, H, W)], dim=1)
        ctx.meta = (B, C, H, W, cpad)
        return ops.nchw_to_cl(x.contiguous(), dtype)

    @staticmethod
    def backward(ctx, d):
        B, C, H, W, cpad = ctx.meta
        return ops.cl_to_nchw(d.contiguous(), B, cpad, H * W).view(B, cpad, H, W)[:, :C].contiguous(), None, None


class _FromCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B, C, H, W, cpad):
        ctx.meta = (B, C, H, W, cpad, x.dtype)
        return ops.cl_to_nchw(x, B, cpad, H * W).view(B, cpad, H, W)[:, :C].contiguous()

    @staticmethod
    def backward(ctx, d):
        B, C, H, W, cpad, dt = ctx.meta
        if cpad != C:
            d = torch.cat([d, d.new_zeros(B, cpad - C, H, W)], dim=1)
        return ops.nchw_to_cl(d.float().contiguous(), dt), None, None, None, None, None


def sinusoid(t, dim, max_period=10000.0):
    B = t.numel()
    out = torch.empty(B, dim, device=t.device, dtype=torch.float32)
    t = t.float().contiguous()
    L.call("uwu_timestep_embedding", L.ptr(t), B, dim, float(max_period), L.ptr(out), L.F32, L.stream())
    return out


class UNet2DConditionModel(nn.Module):
    def __init__(self, config=None, compute_dtype="bf16", **kw):
        super().__init__()
        # init_weights=False: leave the flat parameter buffer zero (a state dict is loaded next; the default initialisers draw
        # 0.9-2.6 G random numbers on the host)
        init_weights = kw.pop("init_weights", True)
        # device=...: build the flat parameter buffer there and draw the initial weights with that device's generator (2.6 G
        # draws of the SDXL shape take half a minute on the host); default: host, as nn.Module constructors do
        device = kw.pop("device", None)
        cfg = dict(SDXL_UNET_CONFIG)
        cfg.update(config or {})
        cfg.update(kw)
        # rope: RoPEUNet2DConditionModel (reference rope_unet.py:589-608) -- an AxialRoPE per attention of every transformer block,
        # q rotated, k only in self-attention; zero_init: HDUNet2DConditionModel's exact-zero residual-branch outputs (:562-580)
        cfg.setdefault("rope", False)
        cfg.setdefault("zero_init", False)
        self.cfg_dict = cfg
        c = type("cfg", (), cfg)()
        c.compute_dtype = compute_dtype
        self.cfg = c
        self.config = type("cfg", (), dict(in_channels=cfg["in_channels"], sample_size=cfg.get("sample_size", 128)))()
        boc = list(cfg["block_out_channels"])
        self.G = cfg["norm_num_groups"]
        self.temb = boc[0] * 4
        P = self.P = _Ctx()
        self.cin_pad, self.cout_pad = _pad8(cfg["in_channels"]), _pad8(cfg["out_channels"])
        self._conv_meta = {}

        def lin(name, cin, cout, bias=True):
            P.add(name + ".weight", (cout, cin))
            if bias:
                P.add(name + ".bias", (cout,))

        def conv(name, cin, cout, cin_store=None, cout_store=None):
            ci, co = cin_store or cin, cout_store or cout
            self._conv_meta[name] = (cin, cout, ci, co)
            P.add(name + ".weight", (co, 9 * ci))
            P.add(name + ".bias", (co,))

        def norm(name, c):
            P.add(name + ".weight", (c,))
            P.add(name + ".bias", (c,))

        def resnet(name, cin, cout):
            norm(name + ".norm1", cin)
            conv(name + ".conv1", cin, cout)
            lin(name + ".time_emb_proj", self.temb, cout)
            norm(name + ".norm2", cout)
            conv(name + ".conv2", cout, cout)
            if cin != cout:
                lin(name + ".conv_shortcut", cin, cout)

        self._t2d_heads = {}

        def t2d(name, dim, depth, nheads):
            self._t2d_heads[name] = nheads
            norm(name + ".norm", dim)
            lin(name + ".proj_in", dim, dim)
            ctx_dim = cfg["cross_attention_dim"]
            for i in range(depth):
                b = f"{name}.transformer_blocks.{i}"
                norm(b + ".norm1", dim)
                for a, kd in (("attn1", dim), ("attn2", ctx_dim)):
                    lin(f"{b}.{a}.to_q", dim, dim, bias=False)
                    lin(f"{b}.{a}.to_k", kd, dim, bias=False)
                    lin(f"{b}.{a}.to_v", kd, dim, bias=False)
                    lin(f"{b}.{a}.to_out.0", dim, dim)
                    if cfg["rope"]:  # AxialRoPE(head_dim, heads): log-frequencies [heads, head_dim / 4] per axis (rope.py:83-92)
                        nh = self._t2d_heads[name]
                        P.add(f"{b}.{a}.axial_rope.freqs_h", (nh, dim // nh // 4))
                        P.add(f"{b}.{a}.axial_rope.freqs_w", (nh, dim // nh // 4))
                    if a == "attn1":
                        norm(b + ".norm2", dim)
                norm(b + ".norm3", dim)
                lin(b + ".ff.net.0.proj", dim, dim * 8)
                lin(b + ".ff.net.2", dim * 4, dim)
            lin(name + ".proj_out", dim, dim)

        conv("conv_in", cfg["in_channels"], boc[0], cin_store=self.cin_pad)
        lin("time_embedding.linear_1", boc[0], self.temb)
        lin("time_embedding.linear_2", self.temb, self.temb)
        if cfg["addition_embed_type"] == "text_time":
            lin("add_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], self.temb)
            lin("add_embedding.linear_2", self.temb, self.temb)
        heads, depths = list(cfg["attention_head_dim"]), list(cfg["transformer_layers_per_block"])
        self.plan_down, ch = [], boc[0]
        for i, t in enumerate(cfg["down_block_types"]):
            cin, ch = ch, boc[i]
            attn = t.startswith("CrossAttn")
            blk = dict(res=[], attn=[], down=None, heads=heads[i], ch=ch)
            for j in range(cfg["layers_per_block"]):
                n = f"down_blocks.{i}.resnets.{j}"
                resnet(n, cin if j == 0 else ch, ch)
                blk["res"].append((n, cin if j == 0 else ch, ch))
                if attn:
                    a = f"down_blocks.{i}.attentions.{j}"
                    t2d(a, ch, depths[i], heads[i])
                    blk["attn"].append((a, depths[i]))
            if i < len(boc) - 1:
                d = f"down_blocks.{i}.downsamplers.0.conv"
                conv(d, ch, ch)
                blk["down"] = d
            self.plan_down.append(blk)
        mid = boc[-1]
        resnet("mid_block.resnets.0", mid, mid)
        t2d("mid_block.attentions.0", mid, depths[-1], heads[-1])
        resnet("mid_block.resnets.1", mid, mid)
        self.mid_heads, self.mid_depth = heads[-1], depths[-1]
        rev, rh, rd = boc[::-1], heads[::-1], depths[::-1]
        self.plan_up, ch = [], rev[0]
        nl = cfg["layers_per_block"] + 1
        for i, t in enumerate(cfg["up_block_types"]):
            prev, ch = ch, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            attn = t.startswith("CrossAttn")
            blk = dict(res=[], attn=[], up=None, heads=rh[i], ch=ch)
            for j in range(nl):
                skip = cin if j == nl - 1 else ch
                rin = (prev if j == 0 else ch) + skip
                n = f"up_blocks.{i}.resnets.{j}"
                resnet(n, rin, ch)
                blk["res"].append((n, rin, ch))
                if attn:
                    a = f"up_blocks.{i}.attentions.{j}"
                    t2d(a, ch, rd[i], rh[i])
                    blk["attn"].append((a, rd[i]))
            if i < len(boc) - 1:
                u = f"up_blocks.{i}.upsamplers.0.conv"
                conv(u, ch, ch)
                blk["up"] = u
            self.plan_up.append(blk)
        norm("conv_norm_out", boc[0])
        conv("conv_out", boc[0], cfg["out_channels"], cout_store=self.cout_pad)

        self.flat = nn.Parameter(torch.zeros(P.n, dtype=torch.float32, device=device))
        P.flat = self.flat
        P.bf16 = compute_dtype == "bf16"
        self.register_buffer("shadow", torch.zeros(0, dtype=torch.bfloat16), persistent=False)
        if init_weights:
            self.reset_parameters()
        else:
            self.refresh_shadow()

    # ------------------------------------------------------------------ parameters
    @torch.no_grad()
    def reset_parameters(self):
        """torch default inits per layer type + reference unet_patch.py:34-45 (N(0,1e-5) on residual out layers)."""
        dev = self.flat.device
        g = torch.Generator(device=dev).manual_seed(torch.initial_seed() % (2 ** 31))
        zero = self.cfg_dict["zero_init"]
        for name, (off, shape) in self.P.registry.items():
            v = self.P.w32(name)
            if ".axial_rope." in name:  # rope.py:74-81 freqs_pixel_log(max_freq = 10): linspace(log pi, log 5 pi) per head
                v.copy_(torch.linspace(math.log(math.pi), math.log(10.0 * math.pi / 2), shape[-1]).expand(shape))
            elif name.endswith(".bias"):
                v.zero_()
            elif len(shape) == 1:
                v.fill_(1.0)  # norm gains
            else:
                fan_in = shape[1]
                if name.endswith(("conv1.weight", "conv2.weight", "conv.weight", "conv_in.weight", "conv_out.weight")):
                    cin, cout, ci, co = self._conv_meta[name[:-7]]
                    fan_in = 9 * cin
                bound = 1.0 / math.sqrt(fan_in)
                v.copy_((torch.rand(shape, generator=g, device=dev) * 2 - 1) * bound)
                if name.endswith((".conv2.weight", "attn1.to_out.0.weight", "attn2.to_out.0.weight", "ff.net.2.weight",
                                  "conv_out.weight")) and "samplers" not in name:
                    v.copy_(torch.randn(shape, generator=g, device=dev) * 1e-5)
                    if zero:  # rope_unet.py:562-580: exact zeros (the biases of these layers are zero already)
                        v.zero_()
        self._zero_padding()
        self.refresh_shadow()

    @torch.no_grad()
    def _zero_padding(self):
        for cname, (cin, cout, ci, co) in self._conv_meta.items():
            w = self.P.w32(cname + ".weight").view(co, 9, ci)
            if ci != cin:
                w[:, :, cin:] = 0
            if co != cout:
                w[cout:] = 0
                self.P.w32(cname + ".bias")[cout:] = 0

    def named_tensors(self):
        """(diffusers name, tensor in diffusers layout) pairs; conv weights are returned as [Cout, Cin, 3, 3]."""
        for name in self.P.registry:
            v = self.P.w32(name)
            base = name[:-7] if name.endswith(".weight") else name[:-5]
            if base in self._conv_meta:
                cin, cout, ci, co = self._conv_meta[base]
                if name.endswith(".weight"):
                    v = v.view(co, 3, 3, ci)[:cout, :, :, :cin].permute(0, 3, 1, 2)
                else:
                    v = v[:cout]
            elif name.endswith("conv_shortcut.weight"):
                v = v[:, :, None, None]
            yield name, v

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = destination if destination is not None else {}
        for name, v in self.named_tensors():
            sd[prefix + name] = v.detach().clone().contiguous()
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict, strict=True, assign=False):
        missing = []
        for name in self.P.registry:
            if name not in state_dict:
                missing.append(name)
                continue
            src = state_dict[name].float()
            dst = self.P.w32(name)
            base = name[:-7] if name.endswith(".weight") else name[:-5]
            if base in self._conv_meta:
                cin, cout, ci, co = self._conv_meta[base]
                if name.endswith(".weight"):
                    dst.view(co, 3, 3, ci)[:cout, :, :, :cin].copy_(src.permute(0, 2, 3, 1))
                else:
                    dst[:cout].copy_(src)
            else:
                dst.copy_(src.reshape(dst.shape))
        if strict and missing:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing[:5]}...")
        self._zero_padding()
        self.refresh_shadow()
        return torch.nn.modules.module._IncompatibleKeys(missing, [])

    def grad_tensor(self, name):
        """Gradient of a parameter in diffusers layout (for parity checks)."""
        gv = self.P.g(name)
        base = name[:-7] if name.endswith(".weight") else name[:-5]
        if base in self._conv_meta:
            cin, cout, ci, co = self._conv_meta[base]
            return gv.view(co, 3, 3, ci)[:cout, :, :, :cin].permute(0, 3, 1, 2) if name.endswith(".weight") else gv[:cout]
        if name.endswith("conv_shortcut.weight"):
            return gv[:, :, None, None]
        return gv

    @torch.no_grad()
    def refresh_shadow(self):
        if not self.P.bf16 or not self.flat.is_cuda:
            return
        if self.shadow.numel() != self.P.n or self.shadow.device != self.flat.device:
            self.shadow = torch.empty(self.P.n, device=self.flat.device, dtype=torch.bfloat16)
        L.call("uwu_cast_f32_to_bf16", L.ptr(self.flat.data), L.ptr(self.shadow), self.P.n, L.stream())
        self.P.shadow = self.shadow
        self.flat._uwu_bf16_shadow = self.shadow

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self.P.flat = self.flat
        self.refresh_shadow()
        return r

    def enable_gradient_checkpointing(self, enable=True):
        """test_scripts/test_train.py:38-39.  Every resnet and every Transformer2D stack becomes one recomputed segment: only
        segment inputs stay resident (≈ 1 GB instead of ≈ 12 GB of saved activations per 4x128x128 sample at the SDXL
        shape) for one extra forward per segment.  Off by default: batch 12 fits 288 GB without it and runs 1/3 faster."""
        self._ckpt = bool(enable)

    def _seg(self, fn, *args):
        if getattr(self, "_ckpt", False) and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint

            return checkpoint(fn, *args, use_reentrant=False, preserve_rng_state=False)
        return fn(*args)

    # ------------------------------------------------------------------ blocks
    def _ones(self, n, dev):
        c = self.__dict__.setdefault("_ones_cache", {})
        if (n, dev) not in c:
            c[(n, dev)] = torch.ones(n, device=dev, dtype=torch.float32)
        return c[(n, dev)]

    def _linear(self, x, name, bias=True, fp32=False):
        return _LinearFn.apply(x, self.P, name + ".weight", name + ".bias" if bias else None, fp32, self.flat)

    def _conv(self, x, name, B, H, W, C, stride=1):
        return _Conv3x3Fn.apply(x, self.P, name + ".weight", name + ".bias", B, H, W, C, stride)

    def _resnet(self, x, emb_act, name, cin, cout, B, H, W):
        HW = H * W
        h = _GroupNormFn.apply(x, self.P, name + ".norm1", B, HW, cin, self.G, 1e-5, True)
        h = self._conv(h, name + ".conv1", B, H, W, cin)
        t = self._linear(emb_act, name + ".time_emb_proj", fp32=True)  # [B, cout] fp32
        h = _AddRowvecFn.apply(h, t, B, HW, cout)
        h = _GroupNormFn.apply(h, self.P, name + ".norm2", B, HW, cout, self.G, 1e-5, True)
        h = self._conv(h, name + ".conv2", B, H, W, cout)
        if cin != cout:
            x = self._linear(x, name + ".conv_shortcut")
        return _AddFn.apply(x, h)

    def _rope_pos(self, B, hw, dev):
        """make_axial_pos(height, width) of the feature map (rope_unet.py:476-480), one row per token row of the batch"""
        key = (B, hw, dev)
        if getattr(self, "_pos_key", None) != key:
            from .rope import make_axial_pos

            self._pos = make_axial_pos(hw[0], hw[1]).float().to(dev).repeat(B, 1).contiguous()
            self._pos_key = key
        return self._pos

    def _attn(self, x, ctx, name, B, T, Tk, heads, key_bias=None):
        D = x.shape[1]
        if self.cfg_dict["rope"]:
            from .rope import _RopeFn

            pos = self._rope_pos(B, self._hw, x.device)
            fh, fw = _FlatParamFn.apply(self.flat, self.P, name + ".axial_rope.freqs_h"), _FlatParamFn.apply(self.flat, self.P, name + ".axial_rope.freqs_w")
            d = D // heads
            span = self.P.span([f"{name}.to_{c}.weight" for c in "qkv"]) if ctx is None else None
            if span is not None and d == 64 and T % 64 == 0 and T <= 256 and self.P.bf16:
                # self-attention up to 256 tokens (every one at 4x32x32 latents): ONE stacked projection and the rotation applied
                # inside the attention kernels' q / k staging (attn_*_mfma<..., ROPE>, reference rope_unet.py:143-147 rotates
                # between projection and SDPA) -- no rotated copies of q / k, no standalone rotation kernels in the forward
                qkv = _LinearFn.apply(x, self.P, span, None, False, self.flat)
                o = ops.rope_attention(qkv, pos[:T], fh, fw, B, T, heads, d)
                return self._linear(o, name + ".to_out.0")
            # longer sequences / cross-attention: separate projections (the rotation needs q / k as tensors of their own), q
            # rotated, k only in self-attention
            q = self._linear(x, name + ".to_q", bias=False)
            src = x if ctx is None else ctx
            k = self._linear(src, name + ".to_k", bias=False)
            v = self._linear(src, name + ".to_v", bias=False)
            q = _RopeFn.apply(q, pos, fh, fw, heads, D // heads)
            if ctx is None:
                k = _RopeFn.apply(k, pos, fh, fw, heads, D // heads)
            o = _AttentionFn.apply(q, k, v, B, T, Tk, heads, D // heads, key_bias)
            return self._linear(o, name + ".to_out.0")
        # q|k|v (self-attention) and k|v (cross-attention) weights sit back to back in the flat buffer: one stacked
        # projection instead of three (two), forward and backward
        if ctx is None:
            span = self.P.span([f"{name}.to_{c}.weight" for c in "qkv"])
            if span is not None:
                qkv = _LinearFn.apply(x, self.P, span, None, False, self.flat)
                o = _PackedAttentionFn.apply(None, qkv, B, T, Tk, heads, D // heads, key_bias)
                return self._linear(o, name + ".to_out.0")
        else:
            span = self.P.span([f"{name}.to_{c}.weight" for c in "kv"])
            if span is not None:
                q = self._linear(x, name + ".to_q", bias=False)
                kv = _LinearFn.apply(ctx, self.P, span, None, False, self.flat)
                o = _PackedAttentionFn.apply(q, kv, B, T, Tk, heads, D // heads, key_bias)
                return self._linear(o, name + ".to_out.0")
        q = self._linear(x, name + ".to_q", bias=False)
        src = x if ctx is None else ctx
        k = self._linear(src, name + ".to_k", bias=False)
        v = self._linear(src, name + ".to_v", bias=False)
        o = _AttentionFn.apply(q, k, v, B, T, Tk, heads, D // heads, key_bias)
        return self._linear(o, name + ".to_out.0")

    def _t2d(self, x, ctx, name, depth, heads, B, HW, C, Tk):
        self._hw = self._hw_of[HW]
        h = _GroupNormFn.apply(x, self.P, name + ".norm", B, HW, C, self.G, 1e-6, False)
        h = self._linear(h, name + ".proj_in")
        ones = self._ones(C, h.device)
        f = None  # branch output still to be added to the residual stream: folded into the next LayerNorm pass
        for i in range(depth):
            b = f"{name}.transformer_blocks.{i}"
            if f is None:
                n = _LayerNormFn.apply(h, self.P, b + ".norm1", 1e-5)
            else:
                h, n = _AddLayerNormFn.apply(h, f, self.P, b + ".norm1", 1e-5, ones)
            a = self._attn(n, None, b + ".attn1", B, HW, HW, heads)
            h, n = _AddLayerNormFn.apply(h, a, self.P, b + ".norm2", 1e-5, ones)
            a = self._attn(n, ctx, b + ".attn2", B, HW, Tk, heads, self._key_bias)
            h, n = _AddLayerNormFn.apply(h, a, self.P, b + ".norm3", 1e-5, ones)
            f = self._linear(_GegluFn.apply(self._linear(n, b + ".ff.net.0.proj")), b + ".ff.net.2")
        h = _AddFn.apply(h, f)
        h = self._linear(h, name + ".proj_out")
        return _AddFn.apply(h, x)

    # ------------------------------------------------------------------ denoiser slot
    def forward(self, sample, timestep, encoder_hidden_states=None, encoder_attention_mask=None,
                added_cond_kwargs=None, cross_attention_kwargs=None, **kw):
        if not self.flat.is_cuda:
            raise L.UwuError("UNet2DConditionModel runs on the HIP device only (no CPU fallback)")
        cfg, P = self.cfg, self.P
        dt = P.dtype
        B, _, H, W = sample.shape
        dev = sample.device
        if P.bf16 and self.shadow.numel() != P.n:
            self.refresh_shadow()
        if self.flat.grad is None and torch.is_grad_enabled():
            self.flat.grad = torch.zeros_like(self.flat.data)
        if P.side is None and os.environ.get("UWU_UNET_FORK", "1") != "0":
            P.side = torch.cuda.Stream(device=dev)
        if P.side is not None and torch.is_grad_enabled():
            # a backward that raised between its first weight gradient and the engine's final callback never joined
            torch.cuda.current_stream(dev).wait_stream(P.side)
            P._join_queued = False
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=dev)
        t = timestep.to(dev).float().reshape(-1).expand(B).contiguous()
        boc = list(cfg.block_out_channels)
        # conditioning path (M = B rows) in fp32
        emb = self._linear(sinusoid(t, boc[0]), "time_embedding.linear_1", fp32=True)
        emb = self._linear(_SiluFn.apply(emb), "time_embedding.linear_2", fp32=True)
        if cfg.addition_embed_type == "text_time":
            ids = added_cond_kwargs["time_ids"].to(dev).float()
            te = sinusoid(ids.flatten(), cfg.addition_time_embed_dim).reshape(B, -1)
            aug = torch.cat([added_cond_kwargs["text_embeds"].to(dev).float(), te], dim=-1).contiguous()
            aug = self._linear(aug, "add_embedding.linear_1", fp32=True)
            aug = self._linear(_SiluFn.apply(aug), "add_embedding.linear_2", fp32=True)
            emb = _AddFn.apply(emb, aug)
        emb_act = _SiluFn.apply(emb)
        ctx, Tk = None, 0
        self._key_bias = None
        if encoder_hidden_states is not None:
            Tk = encoder_hidden_states.shape[1]
            ctx = encoder_hidden_states.to(device=dev, dtype=dt).reshape(B * Tk, -1).contiguous()
            if encoder_attention_mask is not None:
                # (1 = keep, 0 = discard) -> additive score bias of the cross-attention keys (rope_unet.py:448-453)
                if encoder_attention_mask.ndim != 2 or tuple(encoder_attention_mask.shape) != (B, Tk):
                    raise ValueError(f"encoder_attention_mask must be [B, S] = [{B}, {Tk}]")
                keep = encoder_attention_mask.to(device=dev, dtype=torch.float32)
                self._key_bias = ((1.0 - keep) * -10000.0).contiguous()
        # the input needs no gradient, but every op must see a differentiable input to be recorded
        x = sample.float()
        if torch.is_grad_enabled() and not x.requires_grad:
            x = x.detach().requires_grad_(True)
        x = _ToCL.apply(x, self.cin_pad, dt)
        x = self._conv(x, "conv_in", B, H, W, self.cin_pad)
        skips = [(x, boc[0])]
        h, w = H, W
        self._hw_of = {}  # token count -> (height, width) of the feature maps of this call (positions of the RoPE variant)
        hh, ww = H, W
        for _ in range(len(boc)):
            self._hw_of[hh * ww] = (hh, ww)
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
        for blk in self.plan_down:
            for j, (n, cin, cout) in enumerate(blk["res"]):
                x = self._seg(self._resnet, x, emb_act, n, cin, cout, B, h, w)
                if blk["attn"]:
                    a, depth = blk["attn"][j]
                    x = self._seg(self._t2d, x, ctx, a, depth, blk["heads"], B, h * w, cout, Tk)
                skips.append((x, cout))
            if blk["down"]:
                x = self._conv(x, blk["down"], B, h, w, blk["ch"], stride=2)
                h, w = (h + 1) // 2, (w + 1) // 2
                skips.append((x, blk["ch"]))
        mid = boc[-1]
        x = self._seg(self._resnet, x, emb_act, "mid_block.resnets.0", mid, mid, B, h, w)
        x = self._seg(self._t2d, x, ctx, "mid_block.attentions.0", self.mid_depth, self.mid_heads, B, h * w, mid, Tk)
        x = self._seg(self._resnet, x, emb_act, "mid_block.resnets.1", mid, mid, B, h, w)
        for blk in self.plan_up:
            for j, (n, rin, cout) in enumerate(blk["res"]):
                s, sc = skips.pop()
                x = torch.cat([x, s], dim=1)  # channel concat of token-major tensors (data movement only)
                x = self._seg(self._resnet, x, emb_act, n, rin, cout, B, h, w)
                if blk["attn"]:
                    a, depth = blk["attn"][j]
                    x = self._seg(self._t2d, x, ctx, a, depth, blk["heads"], B, h * w, cout, Tk)
            if blk["up"]:
                x = _UpsampleFn.apply(x, B, h, w, blk["ch"])
                h, w = 2 * h, 2 * w
                x = self._conv(x, blk["up"], B, h, w, blk["ch"])
        x = _GroupNormFn.apply(x, P, "conv_norm_out", B, h * w, boc[0], self.G, 1e-5, True)
        x = self._conv(x, "conv_out", B, h, w, boc[0])
        return (_FromCL.apply(x, B, cfg.out_channels, h, w, self.cout_pad),)

    @classmethod
    def from_config(cls, config, **kw):
        kw.pop("subfolder", None)
        if isinstance(config, str):
            presets = {"sdxl": SDXL_UNET_CONFIG, "stabilityai/stable-diffusion-xl-base-1.0": SDXL_UNET_CONFIG,
                       "tiny-unet": TINY_UNET_CONFIG, "sdxl-rope": dict(SDXL_UNET_CONFIG, rope=True, zero_init=True),
                       "sdxl-hd": dict(SDXL_UNET_CONFIG, zero_init=True)}
            if config not in presets:
                raise ValueError(f"unknown UNet config {config!r}; known: {sorted(presets)}")
            config = presets[config]
        return cls(dict(config), **kw)
