"""Oracle DiT: plain PyTorch fp32 (CPU) restatement of the adaLN-Zero diffusion transformer.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED on the reference side: the reference ships no
DiT and the block arithmetic lives in the absent third-party ``diffusers``; this module restates, in ordinary
torch ops + autograd, the semantics the reference itself spells out:

  * ada_norm_zero block: ``norm1 -> (h, gate_msa, shift_mlp, scale_mlp, gate_mlp)``; ``h += gate_msa * attn1``;
    ``h_norm = norm(h) * (1 + scale_mlp) + shift_mlp``; ``h += gate_mlp * ff``
    (reference src/duwu/modules/rope_unet.py:306-309, 344-349, 393-411);
  * attention: to_q/to_k/to_v -> [B,heads,T,d] -> F.scaled_dot_product_attention(no mask, no dropout) -> merge
    heads -> to_out (rope_unet.py:122-166);
  * call contract ``unet(noisy, t, **kwargs)[0]`` (src/duwu/loss/diffusion.py:172-176).

Sizes are Peebles & Xie Table 1; parameter names equal the product's ``state_dict`` so weights copy 1:1.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def sincos_2d(dim, grid):
    def one(d, pos):
        omega = 1.0 / (10000 ** (torch.arange(d // 2, dtype=torch.float64) / (d / 2)))
        out = pos.reshape(-1, 1).double() * omega[None]
        return torch.cat([out.sin(), out.cos()], dim=1)

    gh, gw = torch.meshgrid(torch.arange(grid), torch.arange(grid), indexing="ij")
    return torch.cat([one(dim // 2, gw), one(dim // 2, gh)], dim=1).float()


def timestep_features(t, dim, max_period=10000.0):
    half = dim // 2
    f = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * f[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def axial_rope(x, pos, log_fh, log_fw):
    """reference src/duwu/modules/rope.py:56-71,95-108 as written (rotate_half negates the EVEN element of each pair):
    x [B, T, H, d], pos [T, 2], log-frequencies [H, d/4] per axis."""
    th = torch.cat((pos[:, None, None, 0] * log_fh.exp(), pos[:, None, None, 1] * log_fw.exp()), dim=-1).repeat_interleave(2, -1)
    rh = torch.stack((-x[..., 0::2], x[..., 1::2]), dim=-1).flatten(-2, -1)
    return x * th.cos() + rh * th.sin()


class Block(nn.Module):
    def __init__(self, D, H, r):
        super().__init__()
        self.H = H
        self.qkv = nn.Linear(D, 3 * D)
        self.proj = nn.Linear(D, D)
        self.fc1 = nn.Linear(D, r * D)
        self.fc2 = nn.Linear(r * D, D)

    def forward(self, x, mod, eps, rope=None):
        B, T, D = x.shape
        sh1, sc1, g1, sh2, sc2, g2 = mod.chunk(6, dim=-1)
        h = F.layer_norm(x, (D,), eps=eps) * (1 + sc1[:, None]) + sh1[:, None]
        q, k, v = self.qkv(h).chunk(3, dim=-1)
        q, k, v = [z.view(B, T, self.H, D // self.H) for z in (q, k, v)]
        if rope is not None:  # rope_unet.py:143-147: RoPE on q and (self-attention) k, then SDPA
            pos, fh, fw = rope
            q, k = axial_rope(q, pos, fh, fw), axial_rope(k, pos, fh, fw)
        q, k, v = [z.transpose(1, 2) for z in (q, k, v)]
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, T, D)
        x = x + g1[:, None] * self.proj(a)
        h = F.layer_norm(x, (D,), eps=eps) * (1 + sc2[:, None]) + sh2[:, None]
        x = x + g2[:, None] * self.fc2(F.gelu(self.fc1(h), approximate="tanh"))
        return x


class DiTOracle(nn.Module):
    def __init__(self, depth=12, hidden=384, heads=6, patch=2, sample_size=32, in_channels=4, out_channels=4,
                 mlp_ratio=4, cond_dim=0, freq_dim=256, ln_eps=1e-6, rope=False, **_):
        super().__init__()
        D = hidden
        self.D, self.L, self.p, self.C, self.Co, self.S = D, depth, patch, in_channels, out_channels, sample_size
        self.freq_dim, self.eps, self.cond_dim = freq_dim, ln_eps, cond_dim
        self.x_embedder = nn.Linear(in_channels * patch * patch, D)
        self.t_embedder = nn.Sequential(nn.Linear(freq_dim, D), nn.SiLU(), nn.Linear(D, D))
        if cond_dim > 0:
            self.y_embedder = nn.Linear(cond_dim, D)
        self.adaLN = nn.Linear(D, depth * 6 * D + 2 * D)
        self.blocks = nn.ModuleList([Block(D, heads, mlp_ratio) for _ in range(depth)])
        self.final = nn.Linear(D, out_channels * patch * patch)
        self.register_buffer("pos", sincos_2d(D, sample_size // patch), persistent=False)
        self.rope = None
        if rope:
            hd4 = D // heads // 4
            init = torch.linspace(math.log(math.pi), math.log(10.0 * math.pi / 2), hd4).expand(depth, heads, hd4)

            class _R(nn.Module):
                pass

            self.rope = _R()
            self.rope.freqs_h = nn.Parameter(init.clone())
            self.rope.freqs_w = nn.Parameter(init.clone())
            g = sample_size // patch
            e = torch.linspace(-1.0, 1.0, g + 1)
            c1 = (e[:-1] + e[1:]) / 2  # rope.py:36-53 make_axial_pos(g, g): cell centres in [-1, 1], (y, x) order
            self.register_buffer("pos_xy", torch.stack(torch.meshgrid(c1, c1, indexing="ij"), dim=-1).view(g * g, 2),
                                 persistent=False)

    def forward(self, sample, timestep, encoder_hidden_states=None, encoder_attention_mask=None,
                added_cond_kwargs=None, cross_attention_kwargs=None, **kw):
        B, p = sample.shape[0], self.p
        # Conv2d(k=p, s=p) patch embedding == unfold (feature order c,ph,pw) + Linear
        tok = F.unfold(sample.float(), kernel_size=p, stride=p).transpose(1, 2)  # [B,T,C*p*p]
        x = self.x_embedder(tok) + self.pos[None]
        c = self.t_embedder(timestep_features(timestep.float().reshape(-1).expand(B), self.freq_dim))
        if self.cond_dim > 0:
            pooled = (added_cond_kwargs or {}).get("text_embeds")
            if pooled is None:
                pooled = torch.zeros(B, self.cond_dim)
            c = c + self.y_embedder(pooled.float())
        mod = self.adaLN(F.silu(c))
        D = self.D
        for l, blk in enumerate(self.blocks):
            rope = None if self.rope is None else (self.pos_xy, self.rope.freqs_h[l], self.rope.freqs_w[l])
            x = blk(x, mod[:, l * 6 * D:(l + 1) * 6 * D], self.eps, rope)
        shf, scf = mod[:, self.L * 6 * D:].chunk(2, dim=-1)
        h = F.layer_norm(x, (D,), eps=self.eps) * (1 + scf[:, None]) + shf[:, None]
        out = self.final(h)  # [B,T,Co*p*p]
        out = F.fold(out.transpose(1, 2), output_size=(self.S, self.S), kernel_size=p, stride=p)
        return (out,)
