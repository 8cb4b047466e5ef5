"""Implicit-GEMM 3x3 convolution (uwu_conv3x3_fwd / _dgrad / _wgrad; the UNet resblock / down-sample / up-sample convs,
reference src/duwu/modules/unet_patch.py:13-57 -> diffusers Conv2d) against torch.nn.functional.conv2d on the CPU.

Integer-valued bf16 operands make the comparison EXACT (every product and partial sum is an integer the fp32 accumulator
holds): a wrong tap offset, padding row, stride rule, channel-chunk order or K-slice shows up as a wrong integer."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, C, Cout, stride
    (2, 8, 8, 32, 32, 1),
    (1, 16, 16, 64, 96, 1),
    (3, 8, 16, 32, 64, 1),      # H != W, M = 384 (not a multiple of the 256-row tile)
    (2, 16, 16, 64, 64, 2),
    (1, 32, 32, 320, 320, 1),   # SDXL level-0 width: 5 channel chunks of 64, Cout = 2.5 column tiles
    (2, 16, 16, 320, 640, 2),   # SDXL down-sample
    (1, 8, 8, 1280, 1280, 1),
]


def _data(B, H, W, C, Cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(-2, 3, (B, C, H, W), generator=g).float()
    w = torch.randint(-1, 2, (Cout, C, 3, 3), generator=g).float()
    keep = torch.rand(Cout, C, 3, 3, generator=g) < 0.25  # sparse weights keep the sums small enough for bf16
    w = w * keep
    bias = torch.randint(-3, 4, (Cout,), generator=g).float()
    return x, w, bias


def _cl(t):  # NCHW -> [B*H*W, C] channels-last bf16 on the device
    B, C, H, W = t.shape
    return t.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().bfloat16().cuda()


def _wk(w):  # [Cout, C, 3, 3] -> [Cout, 9*C] tap-major (this build's parameter layout)
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("B,H,W,C,Cout,stride", CASES)
def test_conv3x3_fwd_exact(B, H, W, C, Cout, stride):
    from uwudiff_amd import ops

    x, w, bias = _data(B, H, W, C, Cout, seed=B + C)
    xd = _cl(x)
    assert ops.conv3x3_implicit_ok(xd, B, H, W, C, Cout, stride)
    y = ops.conv3x3_fwd(xd, _wk(w).bfloat16().cuda(), bias.cuda(), B, H, W, C, Cout, stride)
    want = F.conv2d(x, w, bias, stride=stride, padding=1)
    want = want.permute(0, 2, 3, 1).reshape(-1, Cout)
    assert torch.equal(y.float().cpu(), want.bfloat16().float())


@pytest.mark.parametrize("B,H,W,C,Cout,stride", CASES)
def test_conv3x3_dgrad_exact(B, H, W, C, Cout, stride):
    from uwudiff_amd import ops

    x, w, _ = _data(B, H, W, C, Cout, seed=7 + C)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    g = torch.Generator().manual_seed(3)
    dy = torch.randint(-2, 3, (B, Cout, Ho, Wo), generator=g).float()
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, stride=stride, padding=1).backward(dy)
    dx = ops.conv3x3_dgrad(_cl(dy), _wk(w).bfloat16().cuda(), B, H, W, C, Cout, stride)
    want = xr.grad.permute(0, 2, 3, 1).reshape(-1, C)
    assert torch.equal(dx.float().cpu(), want.bfloat16().float())


@pytest.mark.parametrize("B,H,W,C,Cout,stride", CASES)
def test_conv3x3_wgrad_exact(B, H, W, C, Cout, stride):
    from uwudiff_amd import ops

    x, w, _ = _data(B, H, W, C, Cout, seed=11 + C)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    g = torch.Generator().manual_seed(5)
    dy = torch.randint(-1, 2, (B, Cout, Ho, Wo), generator=g).float()
    wr = w.clone().requires_grad_(True)
    br = torch.zeros(Cout, requires_grad=True)
    F.conv2d(x, wr, br, stride=stride, padding=1).backward(dy)
    dw = torch.ones(Cout, 9 * C, device="cuda")   # accumulates on top of existing contents
    db = torch.full((Cout,), 2.0, device="cuda")
    ops.conv3x3_wgrad(_cl(dy), _cl(x), dw, db, B, H, W, C, Cout, stride)
    assert torch.equal(dw.cpu(), _wk(wr.grad) + 1.0)
    assert torch.equal(db.cpu(), br.grad + 2.0)


def test_conv3x3_random_close_and_fallback_rule():
    from uwudiff_amd import ops

    torch.manual_seed(0)
    B, H, W, C, Cout = 2, 32, 32, 320, 320
    x, w, bias = torch.randn(B, C, H, W), torch.randn(Cout, C, 3, 3) * 0.02, torch.randn(Cout)
    y = ops.conv3x3_fwd(_cl(x), _wk(w).bfloat16().cuda(), bias.cuda(), B, H, W, C, Cout, 1)
    want = F.conv2d(x.bfloat16().float(), w.bfloat16().float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    torch.testing.assert_close(y.float().cpu(), want, rtol=2e-2, atol=2e-2)
    xd = _cl(x)
    assert not ops.conv3x3_implicit_ok(xd, B, H, W, 8, 320, 1)         # conv_in: 4 -> padded 8 channels
    assert not ops.conv3x3_implicit_ok(xd.float(), B, H, W, C, Cout, 1)  # fp32 parity mode keeps im2col + exact-fp32 MFMA
