"""GPU parity of the opt-in 256x384 panel GEMM (gemm_panel.hip, reached through the C ABI uwu_gemm for bf16
K-contiguous operands with whole tiles when UWU_GEMM_PANEL=1) against the 128x128 kernel of gemm.hip, which
tests/test_gemm_gpu.py pins to a CPU fp64 matmul.  Asking uwu_gemm for an fp32 C keeps the call on the 128x128
kernel, so both run on the same operands.

Integer operands: every product and partial sum is exactly representable, so the bf16 output must equal the
bf16 rounding of the exact result bit for bit (catches any fragment / swizzle / lane-exchange / tile-walk error).
Random operands: bf16 output within 1 bf16 ulp-ish (rtol 1e-2) of the fp32-output kernel.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _panel_on(monkeypatch):
    # the panel kernel is opt-in (it measured slower than the 128x128 kernel); the host gate
    # reads the variable on every call
    monkeypatch.setenv("UWU_GEMM_PANEL", "1")

# (M, N, K): >= 128 tiles of 256x384 each so the host gate picks the panel kernel; the cases cover one tile per
# workgroup, several tiles per workgroup (ring running across tile ends), an uneven split (300 tiles on 256
# workgroups) and a long K
SHAPES = [(32768, 384, 128), (16384, 1152, 384), (65536, 1152, 384), (76800, 384, 384), (32768, 384, 1536),
          (16384, 1536, 384)]


def _ops(M, N, K, ints, seed=0):
    g = torch.Generator().manual_seed(seed)
    if ints:
        a = torch.randint(-3, 4, (M, K), generator=g).float()
        b = torch.randint(-3, 4, (N, K), generator=g).float()
        bias = torch.randint(-8, 9, (N,), generator=g).float()
    else:
        a, b, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    return a.bfloat16().cuda(), b.bfloat16().cuda(), bias.cuda()


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_panel_exact_integers(M, N, K):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    a, b, bias = _ops(M, N, K, ints=True)
    ref = ops.gemm(a, b, c_dtype=torch.float32)  # 128x128 kernel, exact on integers
    c = ops.gemm(a, b)
    assert c.dtype == torch.bfloat16
    assert torch.equal(c, ref.bfloat16())
    cb = ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS)
    assert torch.equal(cb, (ref + bias).bfloat16())


@pytest.mark.parametrize("M,N,K", [(16384, 1152, 384), (76800, 384, 384), (16384, 1536, 384)])
def test_panel_random_and_gelu(M, N, K):
    import torch.nn.functional as F

    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    a, b, bias = _ops(M, N, K, ints=False, seed=3)
    ref = ops.gemm(a, b, c_dtype=torch.float32) + bias
    c = ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS)
    torch.testing.assert_close(c.float(), ref, rtol=1e-2, atol=1e-2)
    u, f = ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU)
    assert torch.equal(u, c)
    # the activation is applied to the fp32 accumulator, then rounded
    torch.testing.assert_close(f.float(), F.gelu(ref, approximate="tanh"), rtol=1e-2, atol=1e-2)


def test_panel_dgelu_colsum():
    import torch.nn.functional as F

    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    M, N, K = 16384, 1536, 384
    a, b, _ = _ops(M, N, K, ints=False, seed=5)
    u = (torch.randn(M, N, generator=torch.Generator().manual_seed(6)) * 1.5).bfloat16().cuda()
    ref = ops.gemm(a, b, c_dtype=torch.float32)
    uf = u.float().requires_grad_(True)
    F.gelu(uf, approximate="tanh").backward(ref)
    want = uf.grad
    colsum = torch.zeros(N, device="cuda")
    c = ops.gemm(a, b, aux=u, epilogue=L.EPI_DGELU, out2=colsum)
    c = c[0] if isinstance(c, tuple) else c
    torch.testing.assert_close(c.float(), want, rtol=1e-2, atol=1e-2)
    # the column sums are those of the bf16 values that were stored
    torch.testing.assert_close(colsum, c.float().sum(0), rtol=1e-4, atol=0.05)
