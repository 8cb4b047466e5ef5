"""Oracle UNet2DConditionModel-shape denoiser: plain PyTorch fp32 (CPU).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED on the reference side: the reference's denoiser
is ``diffusers.UNet2DConditionModel`` subclassed by ``UNet2DFromScratch`` (reference
src/duwu/modules/unet_patch.py:13-57); diffusers is an unpinned third-party dependency that is not installed and
has no source on disk, and no reference test pins its outputs.  This file restates the published architecture
(SDXL ``unet/config.json`` values in ``SDXL_UNET_CONFIG`` below) with ordinary torch
ops and diffusers' parameter names, cross-checked structurally by the parameter count 2 567.5 M for the SDXL
config (tests/test_unet_cpu.py).  Init rule: ``unet_patch.py:34-45`` (N(0,1e-5) on residual-branch out layers).

Restated semantics:
  time embedding   sinusoid(block_out_channels[0], flip_sin_to_cos, shift 0) -> Linear -> SiLU -> Linear
  text_time        concat(text_embeds, sinusoid(time_ids.flatten(), 256).view(B,-1)) -> Linear -> SiLU -> Linear; emb = t + aug
  ResnetBlock2D    GN(32,1e-5) SiLU conv3x3 (+ Linear(SiLU(emb))) GN SiLU conv3x3 (+ 1x1 shortcut) ; out = x + h
  Transformer2D    GN(32,1e-6) -> [B,HW,C] -> Linear in -> N x Basic -> Linear out -> + residual   (use_linear_projection)
  BasicTransformer LN -> self-attn(no qkv bias) ; LN -> cross-attn(K/V from ctx) ; LN -> GEGLU(erf) FF x4 ; pre-LN residuals
                   attention = F.scaled_dot_product_attention semantics (reference rope_unet.py:122-166)
  Down/Upsample    conv3x3 stride 2 pad 1 ; nearest x2 then conv3x3
  rope = True      RoPEUNet2DConditionModel (rope_unet.py:589-608): every Attention of a transformer block owns an AxialRoPE
                   (`axial_rope.freqs_h / freqs_w`, rope.py:83-108); q is rotated, k only in self-attention (rope_unet.py:122-147),
                   positions = make_axial_pos(height, width) of the feature map (rope_unet.py:476-480, rope.py:10-53)
  init_weight_zero HDUNet2DConditionModel (rope_unet.py:562-580): exact zeros on the residual-branch output layers
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

SDXL_UNET_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
    transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
    addition_embed_type="text_time", addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816,
    norm_num_groups=32,
)


def sinusoid(t, dim, max_period=10000.0):
    half = dim // 2
    f = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * f[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)  # flip_sin_to_cos=True


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cout)
        self.linear_2 = nn.Linear(cout, cout)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-5)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-5)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, emb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(emb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


def make_axial_pos(h, w):
    """reference src/duwu/modules/rope.py:10-53 (pixel_aspect_ratio 1, align_corners False): cell centres inside the
    [-1, 1] bounding box of the longer side, [h * w, 2] in (y, x) order."""
    ar = w / h
    y_min, y_max, x_min, x_max = -1.0, 1.0, -1.0, 1.0
    if ar > 1:
        y_min, y_max = -1 / ar, 1 / ar
    elif ar < 1:
        x_min, x_max = -ar, ar

    def centers(lo, hi, n):
        e = torch.linspace(lo, hi, n + 1)
        return (e[:-1] + e[1:]) / 2

    g = torch.stack(torch.meshgrid(centers(y_min, y_max, h), centers(x_min, x_max, w), indexing="ij"), dim=-1)
    return g.view(h * w, 2)


def axial_rope(x, pos, log_fh, log_fw):
    """reference rope.py:56-71, 95-108 as written (rotate_half negates the EVEN element of each pair): x [B, T, H, d],
    pos [T, 2], log-frequencies [H, d / 4] per axis."""
    th = torch.cat((pos[:, None, None, 0] * log_fh.exp(), pos[:, None, None, 1] * log_fw.exp()), dim=-1).repeat_interleave(2, -1)
    rh = torch.stack((-x[..., 0::2], x[..., 1::2]), dim=-1).flatten(-2, -1)
    return x * th.cos() + rh * th.sin()


class _AxialRoPE(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        lf = torch.linspace(math.log(math.pi), math.log(10.0 * math.pi / 2), dim // 4).expand(heads, dim // 4)  # rope.py:74-92
        self.freqs_h = nn.Parameter(lf.clone())
        self.freqs_w = nn.Parameter(lf.clone())


class Attention(nn.Module):
    def __init__(self, dim, heads, ctx_dim=None, rope=False):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim or dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim or dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim)])
        if rope:
            self.axial_rope = _AxialRoPE(dim // heads, heads)

    def forward(self, x, ctx=None, pos=None):
        B, T, D = x.shape
        bias = None
        if isinstance(ctx, tuple):  # (encoder states, additive key bias [B,1,1,S]) -- rope_unet.py:106-114, 440-453
            ctx, bias = ctx
        self_attn = ctx is None
        ctx = x if ctx is None else ctx
        q, k, v = self.to_q(x), self.to_k(ctx), self.to_v(ctx)
        q, k, v = [z.view(B, -1, self.heads, D // self.heads) for z in (q, k, v)]
        if pos is not None:  # rope_unet.py:143-147: q always, k only when it comes from the same tokens
            q = axial_rope(q, pos, self.axial_rope.freqs_h, self.axial_rope.freqs_w)
            if self_attn:
                k = axial_rope(k, pos, self.axial_rope.freqs_h, self.axial_rope.freqs_w)
        q, k, v = [z.transpose(1, 2) for z in (q, k, v)]
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias).transpose(1, 2).reshape(B, T, D)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, ctx_dim, rope=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, rope=rope)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, ctx_dim, rope=rope)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx, pos=None):
        x = x + self.attn1(self.norm1(x), pos=pos)
        x = x + self.attn2(self.norm2(x), ctx, pos=pos)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, depth, ctx_dim, groups, rope=False):
        super().__init__()
        self.rope = rope
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, ctx_dim, rope) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_in(h)
        pos = make_axial_pos(H, W) if self.rope else None
        for blk in self.transformer_blocks:
            h = blk(h, ctx, pos)
        h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        return h + x


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, n, groups, attn=None, add_down=True):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups) for i in range(n)])
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, groups=groups, **attn) for _ in range(n)])
        self.has_attn = bool(attn)
        if add_down:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.add_down = add_down

    def forward(self, x, emb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, emb)
            if self.has_attn:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.add_down:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, temb, n, groups, attn=None, add_up=True):
        super().__init__()
        rs = []
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb, groups))
        self.resnets = nn.ModuleList(rs)
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, groups=groups, **attn) for _ in range(n)])
        self.has_attn = bool(attn)
        if add_up:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.add_up = add_up

    def forward(self, x, skips, emb, ctx):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], dim=1), emb)
            if self.has_attn:
                x = self.attentions[i](x, ctx)
        if self.add_up:
            x = self.upsamplers[0](x)
        return x


class MidBlock(nn.Module):
    def __init__(self, c, temb, groups, attn):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups), ResnetBlock2D(c, c, temb, groups)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, groups=groups, **attn)])

    def forward(self, x, emb, ctx):
        x = self.resnets[0](x, emb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, emb)


class UNetOracle(nn.Module):
    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
                 down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                 up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                 transformer_layers_per_block=(1, 2, 10), attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
                 addition_embed_type="text_time", addition_time_embed_dim=256,
                 projection_class_embeddings_input_dim=2816, norm_num_groups=32, rope=False, **_):
        super().__init__()
        boc = list(block_out_channels)
        G = norm_num_groups
        temb = boc[0] * 4
        self.c0 = boc[0]
        self.add_type = addition_embed_type
        self.add_time_dim = addition_time_embed_dim
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        if addition_embed_type == "text_time":
            self.add_embedding = TimestepEmbedding(projection_class_embeddings_input_dim, temb)

        def attn_cfg(i):
            return dict(heads=attention_head_dim[i], depth=transformer_layers_per_block[i], ctx_dim=cross_attention_dim, rope=rope)

        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, t in enumerate(down_block_types):
            cin, ch = ch, boc[i]
            self.down_blocks.append(DownBlock(cin, ch, temb, layers_per_block, G,
                                              attn_cfg(i) if t.startswith("CrossAttn") else None,
                                              add_down=i < len(boc) - 1))
        self.mid_block = MidBlock(boc[-1], temb, G, attn_cfg(len(boc) - 1))
        self.up_blocks = nn.ModuleList()
        rev = boc[::-1]
        rh, rd = list(attention_head_dim)[::-1], list(transformer_layers_per_block)[::-1]
        ch = rev[0]
        for i, t in enumerate(up_block_types):
            prev, ch = ch, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            a = dict(heads=rh[i], depth=rd[i], ctx_dim=cross_attention_dim, rope=rope) if t.startswith("CrossAttn") else None
            self.up_blocks.append(UpBlock(cin, ch, prev, temb, layers_per_block + 1, G, a, add_up=i < len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(G, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @torch.no_grad()
    def init_weight(self):
        """reference unet_patch.py:34-45: near-zero init of every layer followed by a residual connection."""
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock):
                nn.init.normal_(m.attn1.to_out[0].weight, 0.0, 1e-5)
                nn.init.normal_(m.attn2.to_out[0].weight, 0.0, 1e-5)
                nn.init.normal_(m.ff.net[2].weight, 0.0, 1e-5)
            if isinstance(m, ResnetBlock2D):
                nn.init.normal_(m.conv2.weight, 0.0, 1e-5)
        nn.init.normal_(self.conv_out.weight, 0.0, 1e-5)

    @torch.no_grad()
    def init_weight_zero(self):
        """reference rope_unet.py:562-580 (HDUNet2DConditionModel): exact zeros instead of N(0, 1e-5), biases included."""
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock):
                m.attn1.to_out[0].weight.zero_()
                m.attn2.to_out[0].weight.zero_()
                m.ff.net[2].weight.zero_()
                m.ff.net[2].bias.zero_()
            if isinstance(m, ResnetBlock2D):
                m.conv2.weight.zero_()
                m.conv2.bias.zero_()
        self.conv_out.weight.zero_()

    def forward(self, sample, timestep, encoder_hidden_states=None, encoder_attention_mask=None,
                added_cond_kwargs=None, cross_attention_kwargs=None, **kw):
        B = sample.shape[0]
        t = timestep.float().reshape(-1).expand(B)
        emb = self.time_embedding(sinusoid(t, self.c0))
        if self.add_type == "text_time":
            ids = added_cond_kwargs["time_ids"].float()
            te = sinusoid(ids.flatten(), self.add_time_dim).reshape(B, -1)
            emb = emb + self.add_embedding(torch.cat([added_cond_kwargs["text_embeds"].float(), te], dim=-1))
        ctx = encoder_hidden_states.float() if encoder_hidden_states is not None else None
        if encoder_attention_mask is not None and ctx is not None:
            # (1 = keep, 0 = discard) -> additive bias, as the reference's Transformer2D wrapper does (rope_unet.py:448-453)
            ctx = (ctx, ((1 - encoder_attention_mask.float()) * -10000.0)[:, None, None, :])
        x = self.conv_in(sample.float())
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ctx)
            skips += outs
        x = self.mid_block(x, emb, ctx)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, ctx)
        return (self.conv_out(F.silu(self.conv_norm_out(x))),)
