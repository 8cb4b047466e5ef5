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
