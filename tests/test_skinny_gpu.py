"""fp32 Linears with at most 64 rows (csrc/skinny.hip: the conditioning path of the denoisers) against an fp64 torch
reference of the same op; shapes of DiT-S/2 (timestep MLP 256 -> 384 -> 384, pooled-text projection 1280 -> 384, the adaLN
modulation Linear 384 -> 12 * 6 * 384 + 2 * 384) plus ragged ones."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(16, 384, 256), (16, 384, 384), (16, 384, 1280), (16, 28416, 384), (64, 28416, 384), (1, 384, 256),
          (5, 70, 36), (33, 130, 132), (64, 384, 256), (8, 1000, 384)]


def ref(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g),
            torch.randn(M, N, generator=g))


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_skinny_forward(M, N, K):
    from uwudiff_amd import lib as L, ops

    assert ops.skinny_linear_ok(M, N, K)
    x, w, b, _ = ref(M, N, K, 1)
    want = x.double() @ w.double().T
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    got = ops.skinny_linear_fwd(xd, wd)
    assert torch.allclose(got.cpu().double(), want, rtol=1e-5, atol=1e-5)
    got = ops.skinny_linear_fwd(xd, wd, bd)
    assert torch.allclose(got.cpu().double(), want + b.double(), rtol=1e-5, atol=1e-5)
    y, h = ops.skinny_linear_fwd(xd, wd, bd, epilogue=L.EPI_BIAS_SILU)
    assert torch.allclose(y.cpu().double(), want + b.double(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(h.cpu().double(), torch.nn.functional.silu(want + b.double()), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [s for s in SHAPES if s[2] <= 512])
def test_skinny_input_gradient(M, N, K):
    from uwudiff_amd import ops

    _, w, _, dy = ref(M, N, K, 2)
    want = dy.double() @ w.double()
    got = ops.skinny_linear_dgrad(dy.cuda(), w.cuda())
    assert torch.allclose(got.cpu().double(), want, rtol=1e-4, atol=1e-4 * float(want.abs().max()))
    # the output is overwritten, not accumulated
    again = ops.skinny_linear_dgrad(dy.cuda(), w.cuda())
    assert torch.allclose(again, got, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_skinny_weight_and_bias_gradient(M, N, K):
    from uwudiff_amd import ops

    x, _, _, dy = ref(M, N, K, 3)
    g = torch.Generator().manual_seed(4)
    dw0, db0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    dw, db = dw0.cuda(), db0.cuda()
    ops.skinny_linear_wgrad(dy.cuda(), x.cuda(), dw, db)
    assert torch.allclose(dw.cpu().double(), dw0.double() + dy.double().T @ x.double(), rtol=1e-5, atol=1e-4)
    assert torch.allclose(db.cpu().double(), db0.double() + dy.double().sum(0), rtol=1e-5, atol=1e-4)
    dw2 = dw0.cuda()
    ops.skinny_linear_wgrad(dy.cuda(), x.cuda(), dw2)  # no bias gradient
    assert torch.equal(dw2, dw)


def test_skinny_rejects_uncovered_shapes():
    from uwudiff_amd import lib as L, ops

    assert not ops.skinny_linear_ok(65, 384, 384) and not ops.skinny_linear_ok(64, 384, 1280)
    with pytest.raises(L.UwuError):
        ops.skinny_linear_fwd(torch.zeros(65, 384, device="cuda"), torch.zeros(384, 384, device="cuda"))
