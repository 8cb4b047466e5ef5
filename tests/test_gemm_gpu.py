"""GPU parity of the MFMA GEMM (C ABI uwu_gemm) vs a plain fp32 CPU matmul.

Exact-integer operands make the bf16 and fp32 MFMA paths bit-checkable (products and sums are exactly
representable), which catches any fragment-layout / swizzle / transpose error; random operands then check the
tolerance: fp32 path 1e-5 rel (exact-fp32 MFMA), bf16 path compared against the same bf16-rounded operands.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, b, ta, tb):
    A = a.float().cpu().double()
    B = b.float().cpu().double()
    A = A.t() if ta else A
    B = B if tb else B.t()
    return (A @ B).float()


def _operands(M, N, K, ta, tb, dtype, ints, seed=0):
    g = torch.Generator().manual_seed(seed)
    sa = (K, M) if ta else (M, K)
    sb = (K, N) if tb else (N, K)
    if ints:
        a = torch.randint(-3, 4, sa, generator=g).float()
        b = torch.randint(-3, 4, sb, generator=g).float()
    else:
        a = torch.randn(sa, generator=g)
        b = torch.randn(sb, generator=g)
    return a.to(dtype).cuda(), b.to(dtype).cuda()


SHAPES = [(128, 128, 64), (256, 384, 384), (200, 72, 40), (16, 2304, 384), (1000, 16, 384), (512, 384, 16),
          (130, 132, 200)]
TRANS = [(False, False), (False, True), (True, True)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", TRANS)
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_exact_integers(M, N, K, ta, tb, dtype):
    from uwudiff_amd import ops

    epc = 8 if dtype == torch.bfloat16 else 4
    if (ta and M % epc) or (tb and N % epc) or ((not ta or not tb) and K % epc) or N % 4:
        pytest.skip("shape violates the kernel's 16-byte vector rule for this layout (host check covers it)")
    a, b = _operands(M, N, K, ta, tb, dtype, ints=True)
    c = ops.gemm(a, b, trans_a=ta, trans_b=tb, c_dtype=torch.float32)
    torch.testing.assert_close(c.cpu(), _ref(a, b, ta, tb), rtol=0, atol=0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ta,tb", TRANS)
def test_gemm_random(ta, tb, dtype):
    from uwudiff_amd import ops

    M, N, K = 384, 256, 1536
    a, b = _operands(M, N, K, ta, tb, dtype, ints=False, seed=1)
    c = ops.gemm(a, b, trans_a=ta, trans_b=tb, c_dtype=torch.float32)
    torch.testing.assert_close(c.cpu(), _ref(a, b, ta, tb), rtol=2e-5, atol=2e-4)
    if dtype == torch.bfloat16:
        cb = ops.gemm(a, b, trans_a=ta, trans_b=tb)
        assert cb.dtype == torch.bfloat16
        torch.testing.assert_close(cb.float().cpu(), _ref(a, b, ta, tb).bfloat16().float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    import torch.nn.functional as F

    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    M, N, K = 260, 384, 128
    a, b = _operands(M, N, K, False, False, dtype, ints=False, seed=2)
    bias = torch.randn(N).cuda()
    ref = _ref(a, b, False, False) + bias.cpu()
    tol = dict(rtol=1e-5, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    c = ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS)
    torch.testing.assert_close(c.float().cpu(), ref, **tol)
    u, f = ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU)
    torch.testing.assert_close(u.float().cpu(), ref, **tol)
    torch.testing.assert_close(f.float().cpu(), F.gelu(u.float().cpu(), approximate="tanh"), **tol)
    u, s = ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_SILU)
    torch.testing.assert_close(s.float().cpu(), F.silu(u.float().cpu()), **tol)
    # dgelu: C = (A.B) * gelu'(aux)
    aux = torch.randn(M, N).to(dtype).cuda()
    x = aux.float().cpu().requires_grad_(True)
    F.gelu(x, approximate="tanh").sum().backward()
    d = ops.gemm(a, b, aux=aux, epilogue=L.EPI_DGELU)
    torch.testing.assert_close(d.float().cpu(), _ref(a, b, False, False) * x.grad, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("split", [1, 4, 7])
def test_gemm_wgrad_accumulate(dtype, split):
    """dW += dY^T X with split-K atomics; integer data keeps fp32 atomics order-independent (exact)."""
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    Mtok, Nout, Kin = 2048, 384, 136
    g = torch.Generator().manual_seed(3)
    dy = torch.randint(-2, 3, (Mtok, Nout), generator=g).float().to(dtype).cuda()
    x = torch.randint(-2, 3, (Mtok, Kin), generator=g).float().to(dtype).cuda()
    dw = torch.ones(Nout, Kin, device="cuda")
    ops.gemm(dy, x, trans_a=True, trans_b=True, epilogue=L.EPI_ACCUM, out=dw, split_k=split)
    ref = 1.0 + dy.float().cpu().t() @ x.float().cpu()
    torch.testing.assert_close(dw.cpu(), ref, rtol=0, atol=0)


def test_gemm_rejects_bad_shapes():
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    a = torch.zeros(128, 66, device="cuda", dtype=torch.bfloat16)  # K=66 not a multiple of 8
    b = torch.zeros(128, 66, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(L.UwuError):
        ops.gemm(a, b)


# ---- ring kernel (gemm_r3_kernel): picked by the host for K-contiguous bf16 operands once there are >= 512 tiles of
# 256x128.  It issues the same sequence of 16x16x32 MFMAs per output as the 128x128 kernel, so the two must agree bit
# for bit on any operands; UWU_GEMM_R3=0 (read per call) keeps a call on the 128x128 kernel, which the tests above
# pin to the CPU fp64 matmul.
R3_SHAPES = [(65536, 1152, 384), (52000, 648, 160), (262144, 128, 96), (33000, 1536, 1536)]


def _both(monkeypatch, fn):
    monkeypatch.setenv("UWU_GEMM_R3", "0")
    ref = fn()
    monkeypatch.setenv("UWU_GEMM_R3", "1")
    return ref, fn()


@pytest.mark.parametrize("M,N,K", R3_SHAPES)
@pytest.mark.parametrize("c_dtype", [torch.bfloat16, torch.float32])
def test_gemm_r3_matches_128_kernel(M, N, K, c_dtype, monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=False, seed=7)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(8)).cuda()
    ref, got = _both(monkeypatch, lambda: ops.gemm(a, b, c_dtype=c_dtype))
    assert torch.equal(ref, got)
    ref, got = _both(monkeypatch, lambda: ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS, c_dtype=c_dtype))
    assert torch.equal(ref, got)
    if M * N <= 65536 * 1152:
        (u0, f0), (u1, f1) = _both(monkeypatch, lambda: ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU, c_dtype=c_dtype))
        assert torch.equal(u0, u1) and torch.equal(f0, f1)


def test_gemm_r3_exact_integers_and_dgelu(monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    M, N, K = 65536, 1536, 384
    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=True, seed=9)
    monkeypatch.setenv("UWU_GEMM_R3", "1")
    c = ops.gemm(a, b, c_dtype=torch.float32)
    # exact on integers: compare with a torch fp32 matmul of the same (exactly representable) operands
    want = a.float() @ b.float().t()
    assert torch.equal(c, want)
    u = (torch.randn(M, N, generator=torch.Generator().manual_seed(10)) * 1.5).bfloat16().cuda()

    def run():
        cs = torch.zeros(N, device="cuda")
        out = ops.gemm(a, b, aux=u, epilogue=L.EPI_DGELU, out2=cs)
        return (out[0] if isinstance(out, tuple) else out), cs

    (d0, s0), (d1, s1) = _both(monkeypatch, run)
    assert torch.equal(d0, d1)
    torch.testing.assert_close(s0, s1, rtol=1e-4, atol=1.0)  # fp32 atomics: order differs


def test_gemm_big_tile_matches_128_kernel(monkeypatch):
    """256x256 kernel (gemm_big_kernel) on the two GELU Linears: same bits as the 128x128 kernel."""
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    def both(fn):
        monkeypatch.setenv("UWU_GEMM_R3", "0")
        monkeypatch.setenv("UWU_GEMM_BIG", "0")
        ref = fn()
        monkeypatch.setenv("UWU_GEMM_R3", "1")
        monkeypatch.setenv("UWU_GEMM_BIG", "1")
        return ref, fn()

    for M, N, K in [(65536, 1536, 384), (33000, 768, 192)]:
        a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=False, seed=31)
        bias = torch.randn(N, generator=torch.Generator().manual_seed(32)).cuda()
        (u0, f0), (u1, f1) = both(lambda: ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU))
        assert torch.equal(u0, u1) and torch.equal(f0, f1)
        for kw in ({}, dict(bias=bias, epilogue=L.EPI_BIAS)):
            ref, got = both(lambda: ops.gemm(a, b, **kw))
            assert torch.equal(ref, got)
        # input gradient with the GELU derivative and the bias-gradient column sums: dy [M, K'] x W [K', N]
        Kc = K
        dy, w = _operands(M, N, Kc, False, True, torch.bfloat16, ints=False, seed=33)
        u = (torch.randn(M, N, generator=torch.Generator().manual_seed(34)) * 1.5).bfloat16().cuda()

        def run():
            cs = torch.zeros(N, device="cuda")
            out = ops.gemm(dy, w, trans_b=True, aux=u, epilogue=L.EPI_DGELU, out2=cs)
            return (out[0] if isinstance(out, tuple) else out), cs

        (d0, s0), (d1, s1) = both(run)
        assert torch.equal(d0, d1)
        torch.testing.assert_close(s0, s1, rtol=1e-3, atol=0.5)  # fp32 atomics: order differs
        ref, got = both(lambda: ops.gemm(dy, w, trans_b=True))
        assert torch.equal(ref, got)


def test_gemm_wide_tile_matches_128_kernel(monkeypatch):
    """192x384 kernel (gemm_wide_kernel) on the N = 384 / 1152 Linears: same bits as the 128x128 kernel."""
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    def both(fn):
        monkeypatch.setenv("UWU_GEMM_R3", "0")
        monkeypatch.setenv("UWU_GEMM_WIDE", "0")
        ref = fn()
        monkeypatch.setenv("UWU_GEMM_R3", "1")
        monkeypatch.setenv("UWU_GEMM_WIDE", "1")
        return ref, fn()

    for M, N, K in [(65536, 384, 1536), (65536, 1152, 384), (50000, 384, 384)]:
        a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=False, seed=41)
        bias = torch.randn(N, generator=torch.Generator().manual_seed(42)).cuda()
        for kw in ({}, dict(bias=bias, epilogue=L.EPI_BIAS)):
            ref, got = both(lambda: ops.gemm(a, b, **kw))
            assert torch.equal(ref, got)
    for M, N, K in [(65536, 384, 1152), (50000, 384, 1536), (65536, 1152, 384)]:  # input gradients: dy [M,K] x W [K,N]
        dy, w = _operands(M, N, K, False, True, torch.bfloat16, ints=False, seed=43)
        ref, got = both(lambda: ops.gemm(dy, w, trans_b=True))
        assert torch.equal(ref, got)


def test_gemm_m64_tile_matches_128_kernel(monkeypatch):
    """64x128 kernel (gemm_m64_kernel, small token counts): same bits as the 128x128 kernel."""
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    def both(fn):
        monkeypatch.setenv("UWU_GEMM_M64", "0")
        ref = fn()
        monkeypatch.setenv("UWU_GEMM_M64", "1")
        return ref, fn()

    for M, N, K in [(4096, 384, 384), (4096, 1152, 384), (4000, 1536, 384), (1000, 200, 128)]:
        a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=False, seed=51)
        bias = torch.randn(N, generator=torch.Generator().manual_seed(52)).cuda()
        for kw in ({}, dict(bias=bias, epilogue=L.EPI_BIAS)):
            ref, got = both(lambda: ops.gemm(a, b, **kw))
            assert torch.equal(ref, got)
        (u0, f0), (u1, f1) = both(lambda: ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU))
        assert torch.equal(u0, u1) and torch.equal(f0, f1)
        dy, w = _operands(M, N, K, False, True, torch.bfloat16, ints=False, seed=53)  # input gradient: dy [M,K] x W [K,N]
        ref, got = both(lambda: ops.gemm(dy, w, trans_b=True))
        assert torch.equal(ref, got)
        u = (torch.randn(M, N, generator=torch.Generator().manual_seed(54)) * 1.5).bfloat16().cuda()

        def run():
            cs = torch.zeros(N, device="cuda")
            out = ops.gemm(dy, w, trans_b=True, aux=u, epilogue=L.EPI_DGELU, out2=cs)
            return (out[0] if isinstance(out, tuple) else out), cs

        (d0, s0), (d1, s1) = both(run)
        assert torch.equal(d0, d1)
        torch.testing.assert_close(s0, s1, rtol=1e-3, atol=0.2)  # fp32 atomics: order differs


# ---- K-major x K-major accumulate kernel (gemm_tr_kernel: LDS-DMA + ds_read_b64_tr_b16), used for the weight
# gradients dW += dY^T X with split-K.  Integer operands make every partial sum exact, so the fp32 atomics are
# order-independent and the result must equal both the exact matmul and the 128x128 kernel (UWU_GEMM_TR=0).
TR_SHAPES = [(1536, 384, 65536, 14), (1152, 384, 8192, 7), (384, 1536, 4096, 3), (200, 264, 2048, 5), (768, 768, 2080, 1)]


@pytest.mark.parametrize("M,N,K,split", TR_SHAPES)
def test_gemm_tr_exact_integers(M, N, K, split, monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    a, b = _operands(M, N, K, True, True, torch.bfloat16, ints=True, seed=11)

    def run():
        out = torch.ones(M, N, device="cuda")  # accumulate on top of existing content
        ops.gemm(a, b, trans_a=True, trans_b=True, epilogue=L.EPI_ACCUM, out=out, split_k=split)
        return out

    monkeypatch.setenv("UWU_GEMM_TR", "0")
    ref = run()
    monkeypatch.setenv("UWU_GEMM_TR", "1")
    got = run()
    want = a.float().t() @ b.float() + 1.0
    assert torch.equal(ref, want)
    assert torch.equal(got, want)


def test_gemm_tr_random(monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    M, N, K = 1152, 384, 16384
    a, b = _operands(M, N, K, True, True, torch.bfloat16, ints=False, seed=12)
    monkeypatch.setenv("UWU_GEMM_TR", "1")
    out = torch.zeros(M, N, device="cuda")
    ops.gemm(a, b, trans_a=True, trans_b=True, epilogue=L.EPI_ACCUM, out=out, split_k=9)
    want = (a.double().t() @ b.double()).float()
    torch.testing.assert_close(out, want, rtol=1e-4, atol=2e-2)


# (N or M a multiple of 384 with K >= 4096 and a scratch: the wide one-workgroup-per-CU kernel, 192x384 / 384x192 tiles,
#  half-used last sub-image, ragged last row / column tile, 4-stage ring shorter than the K loop and longer)
@pytest.mark.parametrize("M,N,K", [(1536, 384, 65536), (1152, 384, 8192), (384, 1536, 4096), (200, 264, 2048),
                                   (384, 384, 6144), (200, 384, 4096), (384, 264, 4096), (768, 768, 8192),
                                   (1152, 1152, 4096), (4608, 1152, 4096)])
def test_gemm_wgrad_scratch_path(M, N, K, monkeypatch):
    """uwu_gemm_wgrad with split-K scratch (slices + reduce kernel) == exact matmul on integers, on top of existing C."""
    from uwudiff_amd import ops

    monkeypatch.setenv("UWU_GEMM_TRW", "1")  # the wide kernel also for the short reductions of this list (K >= 4096)

    a, b = _operands(M, N, K, True, True, torch.bfloat16, ints=True, seed=13)
    want = a.float().t() @ b.float() + 2.0
    bsum = a.float().sum(0) + 3.0  # the fused bias gradient: column sums of A on top of existing content
    for scratch in (ops.gemm_wgrad_scratch(M, N, K), None):
        dw = torch.full((M, N), 2.0, device="cuda")
        db = torch.full((M,), 3.0, device="cuda")
        ops.gemm_wgrad(a, b, dw, scratch=scratch, bias_grad=db)
        assert torch.equal(dw, want)
        assert torch.equal(db, bsum)
    # fp32 operands take the 128x128 kernel with atomics (+ the colsum kernel for the bias gradient)
    dw = torch.zeros(M, N, device="cuda")
    db = torch.full((M,), 3.0, device="cuda")
    ops.gemm_wgrad(a.float(), b.float(), dw, bias_grad=db)
    assert torch.equal(dw, want - 2.0)
    assert torch.equal(db, bsum)


# ---- ring kernel with the K-major weight read through ds_read_b64_tr_b16 (input gradients): 128x128 (N = 384
# shapes) and 256x128 (wide N) variants must equal gemm_kernel's register-transposing path bit for bit.
@pytest.mark.parametrize("M,N,K", [(65536, 384, 1152), (65536, 1536, 384), (40000, 392, 160), (70000, 1160, 96)])
def test_gemm_r3_input_gradient_matches_128_kernel(M, N, K, monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    a, b = _operands(M, N, K, False, True, torch.bfloat16, ints=False, seed=21)
    ref, got = _both(monkeypatch, lambda: ops.gemm(a, b, trans_b=True))
    assert torch.equal(ref, got)
    u = (torch.randn(M, N, generator=torch.Generator().manual_seed(22)) * 1.5).bfloat16().cuda()

    def run():
        cs = torch.zeros(N, device="cuda")
        out = ops.gemm(a, b, trans_b=True, aux=u, epilogue=L.EPI_DGELU, out2=cs)
        return (out[0] if isinstance(out, tuple) else out), cs

    (d0, s0), (d1, s1) = _both(monkeypatch, run)
    assert torch.equal(d0, d1)
    torch.testing.assert_close(s0, s1, rtol=1e-3, atol=0.5)  # fp32 atomics: order differs


# ---- many-tile weight gradients: the streaming kernel divides the 8 XCDs between K slices and tile lanes (1 slice from 320
# tiles of 256x128 / 128x256 on, 2 from 140, 4 from 80; reductions of <= 1024 K-steps).  One slice writes C directly
# (read-modify-write, no scratch), 2 / 4 go through the scratch + reduce kernel.  UWU_GEMM_TRW=0 keeps the 192x384 kernel
# out of the way so that every shape exercises gemm_tr_kernel; exact on small integers, on top of existing C, with the fused
# bias gradient, ragged last tiles and both tile orientations (tile lanes along m and along n).
@pytest.mark.parametrize("M,N,K,slices", [
    (10240, 1280, 2048, 1),   # 40 x 10 tiles of 256x128: lanes along m
    (1280, 10240, 2048, 1),   # 10 x 40 tiles of 128x256: lanes along n
    (5000, 2600, 2048, 1),    # ragged in both directions, 20 x 21 tiles
    (3840, 1280, 2048, 2),    # 150 tiles
    (1280, 5120, 4096, 2),    # 200 tiles of 128x256
    (5120, 640, 4096, 4),     # 100 tiles
    (2440, 1272, 2080, 4),    # 10 x 10 ragged tiles, K not a multiple of the slice length
])
def test_gemm_wgrad_many_tiles_xcd_partition(M, N, K, slices, monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    monkeypatch.setenv("UWU_GEMM_TRW", "0")
    a, b = _operands(M, N, K, True, True, torch.bfloat16, ints=True, seed=21)
    want = a.float().t() @ b.float() + 2.0
    bsum = a.float().sum(0) + 3.0
    need = L.load().uwu_gemm_wgrad_scratch_bytes(M, N, K)
    assert need >= slices * M * N * 4, (need, slices)  # (the size covers the other kernels' slice counts too)
    dw = torch.full((M, N), 2.0, device="cuda")
    db = torch.full((M,), 3.0, device="cuda")
    ops.gemm_wgrad(a, b, dw, scratch=ops.gemm_wgrad_scratch(M, N, K), bias_grad=db)
    assert torch.equal(dw, want)
    assert torch.equal(db, bsum)
    # without a scratch: one slice still writes directly, several fall back to 8 slice lanes + fp32 atomics (exact on integers)
    dw = torch.full((M, N), 2.0, device="cuda")
    ops.gemm_wgrad(a, b, dw, scratch=None)
    assert torch.equal(dw, want)


# ---- A-stationary kernel (gemm_as_kernel: K = 384, whole 256-row panels, N in 1024..2048): same MFMA sequence per output as
# the 256x256 kernel it replaces for fc1 + bias + GELU, so both outputs must agree bit for bit (UWU_GEMM_AS=0 = the old path).
@pytest.mark.parametrize("M,N", [(65536, 1536), (65536, 1024), (196608, 1536), (65536, 2048), (1024, 2048)])
def test_gemm_as_bias_gelu_matches_big_kernel(M, N, monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    K = 384
    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=False, seed=31)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(32)).cuda()
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("UWU_GEMM_AS", flag)
        u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        h = torch.empty_like(u)
        for _ in range(3):  # the chunk pipeline must not depend on timing
            ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU, out=u, out2=h)
        outs[flag] = (u.clone(), h.clone())
    assert torch.equal(outs["1"][0], outs["0"][0])
    assert torch.equal(outs["1"][1], outs["0"][1])
    rows = slice(0, min(M, 2048))  # and against fp64 on a slice
    want = a[rows].double() @ b.double().t() + bias.double()
    assert (outs["1"][0][rows].double() - want).abs().max() <= 2e-2 * want.abs().max()


@pytest.mark.parametrize("M,N", [(65536, 1152), (65536, 1024), (196608, 1152), (512, 1024)])
def test_gemm_as_bias_matches_other_kernels(M, N, monkeypatch):
    """the same kernel with the plain bias epilogue (qkv forward) against the 192x384 / 256x256 kernels: bit for bit."""
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    K = 384
    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=False, seed=33)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(34)).cuda()
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("UWU_GEMM_AS_BIAS", flag)
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS, out=y)
        outs[flag] = y.clone()
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("M,N", [(65536, 1536), (65536, 1024), (196608, 1536), (256, 1024)])
def test_gemm_as_dgelu_matches_k_major_kernel(M, N):
    """du = (dy . W2) * gelu'(u) through the A-stationary kernel on the TRANSPOSED weight (uwu_transpose_bf16) against the
    256x256 kernel on the K-major weight: same MFMA sequence per output, bit for bit."""
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    K = 384
    g = torch.Generator().manual_seed(41)
    dy = (torch.randn(M, K, generator=g) * 0.1).bfloat16().cuda()
    w2 = (torch.randn(K, N, generator=g) / K ** 0.5).bfloat16().cuda()   # [out = K, in = N]: K-major for this product
    u = torch.randn(M, N, generator=g).bfloat16().cuda()
    ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(dy, w2, trans_b=True, aux=u, epilogue=L.EPI_DGELU, out=ref)
    w2t = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    L.call("uwu_transpose_bf16", L.ptr(w2), L.ptr(w2t), K, N, N, K, L.stream())
    assert torch.equal(w2t, w2.t().contiguous())
    got = torch.empty_like(ref)
    for _ in range(3):
        ops.gemm(dy, w2t, aux=u, epilogue=L.EPI_DGELU, out=got)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("N", [1024, 1088, 2048])
def test_gemm_as_exact_integers_all_epilogues(N):
    """The A-stationary kernel driven DIRECTLY on exact data: operands in {-1, 0, 1}, so every sum is a small
    integer that bf16 holds exactly -- any fragment / chunk / interleaved-epilogue indexing error shows as a wrong integer.
    All three epilogues of the kernel: bias (qkv forward), bias + GELU (fc1: pre-activation exact, activation within one bf16
    step of tanh-GELU), and dGELU on aux values where gelu' is 0.5 / 1 / 0 (fc2 input gradient).  Reference: fp32 matmul
    (exact on these integers).  M = 65536 rows of K = 384: the shapes for which the host rule picks gemm_as_kernel."""
    import torch.nn.functional as F

    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    M, K = 65536, 384
    g = torch.Generator().manual_seed(77 + N)
    a = torch.randint(-1, 2, (M, K), generator=g).float().cuda()
    b = torch.randint(-1, 2, (N, K), generator=g).float().cuda()
    bias = torch.randint(-8, 9, (N,), generator=g).float().cuda()
    a16, b16 = a.bfloat16(), b.bfloat16()
    ref = a @ b.t()
    assert float(ref.abs().max()) + 8 < 256
    # bias
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a16, b16, bias=bias, epilogue=L.EPI_BIAS, out=y)
    assert torch.equal(y.float(), ref + bias)
    # bias + GELU: two outputs
    u, h = torch.empty_like(y), torch.empty_like(y)
    ops.gemm(a16, b16, bias=bias, epilogue=L.EPI_BIAS_GELU, out=u, out2=h)
    assert torch.equal(u.float(), ref + bias)
    want = F.gelu(ref + bias, approximate="tanh")
    assert float((h.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    big = (ref + bias) >= 8  # x >= 8: gelu(x) is x to fp32 precision; x <= -8: zero (the kernel's x * sigmoid leaves -1e-28)
    assert torch.equal(h.float()[big], (ref + bias)[big])
    assert float(h.float()[(ref + bias) <= -8].abs().max()) < 1e-20
    # dGELU: aux in {0, 10, -10} -> gelu' in {0.5, 1, 0}
    aux = (torch.randint(-1, 2, (M, N), generator=g).float() * 10).bfloat16().cuda()
    d = torch.empty_like(y)
    ops.gemm(a16, b16, aux=aux, epilogue=L.EPI_DGELU, out=d)
    fac = torch.where(aux == 0, 0.5, torch.where(aux > 0, 1.0, 0.0)).float()
    torch.testing.assert_close(d.float(), ref * fac, rtol=0, atol=1e-20)


# ---- 8-phase 256 x 256 kernel (gemm_p8.hip): K >= 512 Linears of DiT-B/2, DiT-XL/2, the UNet ---------------------------------
# UWU_GEMM_P8=1 forces it on any shape it can run, =0 keeps the older kernels.  Its MFMA sequence per output element is the
# 128x128 kernel's (K steps of 64 in order, two 32-deep halves each), so every epilogue must agree BIT FOR BIT on random
# operands; exact-integer operands pin the half-tile ring (a stale or early read of a slot shows as a wrong integer).
P8_SHAPES = [(4096, 1024, 768), (2048, 768, 3072), (1000, 520, 128), (777, 1160, 960), (512, 256, 192), (300, 264, 1088)]


def _p8_both(monkeypatch, fn):
    monkeypatch.setenv("UWU_GEMM_P8", "0")
    ref = fn()
    monkeypatch.setenv("UWU_GEMM_P8", "1")
    return ref, fn()


@pytest.mark.parametrize("M,N,K", P8_SHAPES)
def test_gemm_p8_exact_integers(M, N, K, monkeypatch):
    from uwudiff_amd import ops

    monkeypatch.setenv("UWU_GEMM_P8", "1")
    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=True, seed=51)
    for _ in range(3):  # repeated launches: the ring's ordering must hold under different timings
        c = ops.gemm(a, b)
        assert torch.equal(c.float(), (a.float() @ b.float().t()).bfloat16().float())
    dy, w = _operands(M, N, K, False, True, torch.bfloat16, ints=True, seed=52)
    for _ in range(3):
        c = ops.gemm(dy, w, trans_b=True)
        assert torch.equal(c.float(), (dy.float() @ w.float()).bfloat16().float())


@pytest.mark.parametrize("M,N,K", P8_SHAPES)
def test_gemm_p8_matches_older_kernels(M, N, K, monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=False, seed=53)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(54)).cuda()
    for kw in ({}, dict(bias=bias, epilogue=L.EPI_BIAS)):
        ref, got = _p8_both(monkeypatch, lambda: ops.gemm(a, b, **kw))
        assert torch.equal(ref, got)
    (u0, f0), (u1, f1) = _p8_both(monkeypatch, lambda: ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU))
    assert torch.equal(u0, u1) and torch.equal(f0, f1)
    dy, w = _operands(M, N, K, False, True, torch.bfloat16, ints=False, seed=55)
    ref, got = _p8_both(monkeypatch, lambda: ops.gemm(dy, w, trans_b=True))
    assert torch.equal(ref, got)
    u = (torch.randn(M, N, generator=torch.Generator().manual_seed(56)) * 1.5).bfloat16().cuda()

    def run():
        cs = torch.zeros(N, device="cuda")
        out = ops.gemm(dy, w, trans_b=True, aux=u, epilogue=L.EPI_DGELU, out2=cs)
        return (out[0] if isinstance(out, tuple) else out), cs

    (d0, s0), (d1, s1) = _p8_both(monkeypatch, run)
    assert torch.equal(d0, d1)
    torch.testing.assert_close(s0, s1, rtol=1e-3, atol=0.5)  # fp32 atomics: order differs
    ref, got = _p8_both(monkeypatch, lambda: ops.gemm(dy, w, trans_b=True, aux=u, epilogue=L.EPI_DGELU))
    assert torch.equal(ref, got)


def test_gemm_p8_bench_shape_exact(monkeypatch):
    """DiT-B/2 fc1 at the secondary bench's launch shape (3072 tiles = 12 rounds of CUs), exact integers, default dispatch."""
    from uwudiff_amd import ops

    M, N, K = 65536, 3072, 768
    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=True, seed=57)
    want = (a.float() @ b.float().t()).bfloat16()
    for _ in range(2):
        assert torch.equal(ops.gemm(a, b), want)


# ---- 128 x 384 sibling (gemm_p8n.hip): N a multiple of 384 -- the N = 384 / 1152 Linears of DiT-S/2 and DiT-XL/2 -------------
# UWU_GEMM_P8N=1 forces it on any shape it can run, =0 keeps the older kernels (UWU_GEMM_P8=0 as well: N = 768 / 1536 would go
# to the 256 x 256 kernel first).  Same MFMA sequence per output element as every other bf16 kernel: bit-identical results.
P8N_SHAPES = [(4096, 384, 384), (2048, 1152, 1536), (1000, 768, 256), (777, 384, 1152), (300, 1536, 640), (66000, 384, 384)]


def _p8n_both(monkeypatch, fn):
    monkeypatch.setenv("UWU_GEMM_P8", "0")
    monkeypatch.setenv("UWU_GEMM_P8N", "0")
    ref = fn()
    monkeypatch.setenv("UWU_GEMM_P8N", "1")
    return ref, fn()


@pytest.mark.parametrize("M,N,K", P8N_SHAPES)
def test_gemm_p8n_exact_integers(M, N, K, monkeypatch):
    from uwudiff_amd import ops

    monkeypatch.setenv("UWU_GEMM_P8", "0")
    monkeypatch.setenv("UWU_GEMM_P8N", "1")
    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=True, seed=71)
    for _ in range(3):  # repeated launches: the ring's ordering must hold under different timings
        c = ops.gemm(a, b)
        assert torch.equal(c.float(), (a.float() @ b.float().t()).bfloat16().float())
    dy, w = _operands(M, N, K, False, True, torch.bfloat16, ints=True, seed=72)
    for _ in range(3):
        c = ops.gemm(dy, w, trans_b=True)
        assert torch.equal(c.float(), (dy.float() @ w.float()).bfloat16().float())


@pytest.mark.parametrize("M,N,K", P8N_SHAPES)
def test_gemm_p8n_matches_older_kernels(M, N, K, monkeypatch):
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=False, seed=73)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(74)).cuda()
    for kw in ({}, dict(bias=bias, epilogue=L.EPI_BIAS)):
        ref, got = _p8n_both(monkeypatch, lambda: ops.gemm(a, b, **kw))
        assert torch.equal(ref, got)
    (u0, f0), (u1, f1) = _p8n_both(monkeypatch, lambda: ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU))
    assert torch.equal(u0, u1) and torch.equal(f0, f1)
    dy, w = _operands(M, N, K, False, True, torch.bfloat16, ints=False, seed=75)
    ref, got = _p8n_both(monkeypatch, lambda: ops.gemm(dy, w, trans_b=True))
    assert torch.equal(ref, got)


def test_gemm_p8n_bench_shape_exact():
    """DiT-S/2 qkv forward with bias at the bench's launch shape (4608 tiles: 18 per workgroup), exact integers, default dispatch."""
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    M, N, K = 196608, 1152, 384
    a, b = _operands(M, N, K, False, False, torch.bfloat16, ints=True, seed=77)
    bias = torch.randint(-3, 4, (N,), generator=torch.Generator().manual_seed(78)).float().cuda()
    want = (a.float() @ b.float().t() + bias).bfloat16()
    for _ in range(2):
        assert torch.equal(ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS), want)
