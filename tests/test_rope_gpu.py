"""GPU: axial RoPE kernel vs the REFERENCE's own output (tests/golden/axial_rope.npz, produced by oracle/make_golden.py
from src/duwu/modules/rope.py) and vs autograd of the same formula for the backward."""
import pytest
import torch

from tests.golden_util import load

pytestmark = pytest.mark.gpu


def ref_formula(x, pos, fh, fw):
    th = torch.cat((pos[..., None, None, 0] * fh.exp(), pos[..., None, None, 1] * fw.exp()), dim=-1).repeat_interleave(2, -1)
    rh = torch.stack((-x[..., 0::2], x[..., 1::2]), dim=-1).flatten(-2, -1)
    return x * th.cos() + rh * th.sin()


def test_axial_rope_matches_reference_golden():
    from uwudiff_amd.rope import AxialRoPE, make_axial_pos

    _, d = load("axial_rope")
    m = AxialRoPE(64, 4).cuda()
    with torch.no_grad():
        m.freqs_h.copy_(d["freqs_h"])
        m.freqs_w.copy_(d["freqs_w"])
    y = m(d["x"].cuda(), d["pos"].cuda())
    torch.testing.assert_close(y.cpu(), d["y"], rtol=2e-5, atol=2e-5)
    # the position grid helper reproduces the reference's make_axial_pos(8, 8) (the golden's pos is its repeat)
    torch.testing.assert_close(make_axial_pos(8, 8), d["pos"][0], rtol=0, atol=1e-7)
    # learnable defaults equal the reference's freqs_pixel_log init
    torch.testing.assert_close(AxialRoPE(64, 4).freqs_h.detach(), d["freqs_h"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_axial_rope_backward(dtype):
    from uwudiff_amd.rope import AxialRoPE, make_axial_pos

    torch.manual_seed(0)
    B, h, w, H, dim = 2, 4, 6, 3, 32
    m = AxialRoPE(dim, H).cuda()
    pos = make_axial_pos(h, w).repeat(B, 1, 1)
    x = torch.randn(B, h * w, H, dim).to(dtype)
    g = torch.randn(B, h * w, H, dim).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    fh, fw = m.freqs_h.detach().cpu().clone().requires_grad_(True), m.freqs_w.detach().cpu().clone().requires_grad_(True)
    ref_formula(xr, pos, fh, fw).backward(g.float())
    xd = x.detach().cuda().requires_grad_(True)
    m(xd, pos.cuda()).backward(g.cuda())
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, **tol)
    t2 = dict(rtol=1e-3, atol=1e-3) if dtype == torch.float32 else dict(rtol=5e-2, atol=0.5)
    torch.testing.assert_close(m.freqs_h.grad.cpu(), fh.grad, **t2)
    torch.testing.assert_close(m.freqs_w.grad.cpu(), fw.grad, **t2)


@pytest.mark.parametrize("B,T,H", [(2, 256, 6), (3, 64, 2), (1, 128, 3)])
def test_rope_fused_into_attention_matches_composed_reference(B, T, H):
    """rope_unet.py:143-153 composed (RoPE on q and k, then SDPA) in fp32 on the CPU, with the rotation of the reference's
    rope.py (ref_formula above is pinned by its golden) -- against the attention kernels that rotate q / k while staging
    them.  Forward, and the gradients wrt q, k, v and both log-frequency tables."""
    from uwudiff_amd import ops
    from uwudiff_amd.rope import make_axial_pos

    torch.manual_seed(B * 100 + T)
    d, D = 64, 64 * H
    side = int(T ** 0.5) if int(T ** 0.5) ** 2 == T else None
    pos = make_axial_pos(side, side) if side else make_axial_pos(T // 8, 8)
    qkv = (torch.randn(B * T, 3 * D) * 0.7).bfloat16()
    fh = (torch.linspace(1.1, 2.7, d // 4).expand(H, d // 4) + torch.randn(H, d // 4) * 0.05).contiguous()
    fw = (torch.linspace(1.1, 2.7, d // 4).expand(H, d // 4) + torch.randn(H, d // 4) * 0.05).contiguous()
    do = (torch.randn(B * T, D) * 0.3).bfloat16()

    x = qkv.float().clone().requires_grad_(True)
    fhr, fwr = fh.clone().requires_grad_(True), fw.clone().requires_grad_(True)
    q, k, v = [t.reshape(B, T, H, d) for t in x.split(D, dim=1)]
    qr = ref_formula(q, pos[None].expand(B, T, 2), fhr, fwr).transpose(1, 2)
    kr = ref_formula(k, pos[None].expand(B, T, 2), fhr, fwr).transpose(1, 2)
    o_ref = torch.nn.functional.scaled_dot_product_attention(qr, kr, v.transpose(1, 2)).transpose(1, 2).reshape(B * T, D)
    o_ref.backward(do.float())

    xd = qkv.cuda().requires_grad_(True)
    fhd, fwd = fh.cuda().requires_grad_(True), fw.cuda().requires_grad_(True)
    o = ops.rope_attention(xd, pos.cuda().float().contiguous(), fhd, fwd, B, T, H, d)
    o.backward(do.cuda())

    def rel(a, b):
        a, b = a.float().cpu(), b.float().cpu()
        return ((a - b).norm() / b.norm()).item()

    assert rel(o, o_ref) < 2e-2, rel(o, o_ref)
    assert rel(xd.grad, x.grad) < 4e-2, rel(xd.grad, x.grad)
    assert rel(fhd.grad, fhr.grad) < 6e-2 and rel(fwd.grad, fwr.grad) < 6e-2, (rel(fhd.grad, fhr.grad), rel(fwd.grad, fwr.grad))


def test_rope_dit_preset_matches_oracle():
    """DiT with axial-RoPE attention (the preset behind ``DiT-S/2-RoPE``: per-layer learnable log-frequencies, q / k rotated
    inside the attention kernels) against the CPU oracle's composed block (oracle/dit.py: rope.py's rotation, then SDPA):
    output and every parameter gradient incl. the frequency tables, bf16 tolerances."""
    from oracle.dit import DiTOracle
    from uwudiff_amd.dit import DiT, DiTConfig, PRESETS

    cfg = dict(depth=2, hidden=128, heads=2, patch=2, sample_size=16, in_channels=4, out_channels=4, cond_dim=32, rope=True)
    torch.manual_seed(0)
    ora = DiTOracle(**cfg)
    with torch.no_grad():
        for n, p in ora.named_parameters():
            if not n.startswith("rope."):
                p.copy_(torch.randn_like(p) * (0.05 if p.dim() > 1 else 0.02))
            else:
                p.add_(torch.randn_like(p) * 0.05)
    model = DiT(DiTConfig(compute_dtype="bf16", **cfg), init="dit").cuda()
    model.load_state_dict(ora.state_dict())
    B = 4
    g = torch.Generator().manual_seed(1)
    x, t = torch.randn(B, 4, 16, 16, generator=g), torch.randint(0, 1000, (B,), generator=g)
    pooled, dout = torch.randn(B, 32, generator=g), torch.randn(B, 4, 16, 16, generator=g) / 256
    yo = ora(x, t, added_cond_kwargs={"text_embeds": pooled})[0]
    yo.backward(dout)
    y = model(x.cuda(), t.cuda(), added_cond_kwargs={"text_embeds": pooled.cuda()})[0]
    y.backward(dout.cuda())

    def rel(a, b):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        return ((a - b).norm() / (b.norm() + 1e-30)).item()

    assert rel(y, yo) < 3e-2, rel(y, yo)
    og = dict(ora.named_parameters())
    bad = {n: rel(model.grad_view(n), og[n].grad) for n, _ in model.named_tensors() if rel(model.grad_view(n), og[n].grad) >= 8e-2}
    assert not bad, bad
    assert PRESETS["DiT-S/2-RoPE"]["rope"] and float(model.grad_view("rope.freqs_h").abs().max()) > 0
