"""GPU parity of the non-GEMM denoiser kernels (C ABI) vs plain PyTorch fp32 CPU references + autograd.

fp32 kernels: rtol 1e-4 (north-star bar is 1e-3).  bf16 kernels: compared at bf16 resolution (2e-2).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def tol(dtype):
    return dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)


def cmp(a, b, **kw):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), **kw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [384, 768, 1152, 64])
def test_add_ln_modulate_fwd_bwd(D, dtype):
    from uwudiff_amd import ops

    torch.manual_seed(0)
    B, T, ML = 3, 64, 6 * D + 8
    x_in = torch.randn(B * T, D).to(dtype)
    y = torch.randn(B * T, D).to(dtype)
    mod = torch.randn(B, ML) * 0.5
    dh = torch.randn(B * T, D).to(dtype)
    dx_in = torch.randn(B * T, D).to(dtype)
    # reference (fp32 math on the same rounded inputs)
    xr, yr = x_in.float().requires_grad_(True), y.float().requires_grad_(True)
    modr = mod.clone().requires_grad_(True)
    gate, shift, scale = modr[:, 0:D], modr[:, D:2 * D], modr[:, 2 * D:3 * D]
    xo = xr + gate.repeat_interleave(T, 0) * yr
    if dtype == torch.bfloat16:
        xo = xo + (xo.detach().bfloat16().float() - xo.detach())  # the kernel stores/normalises the rounded stream
    hr = F.layer_norm(xo, (D,), eps=1e-6) * (1 + scale.repeat_interleave(T, 0)) + shift.repeat_interleave(T, 0)
    (hr * dh.float()).sum().backward(retain_graph=True)
    xo_grad_total = torch.autograd.grad((hr * dh.float()).sum(), xo, retain_graph=True)[0] + dx_in.float()

    md = mod.cuda()
    g, sh, sc = md[:, 0:D], md[:, D:2 * D], md[:, 2 * D:3 * D]
    x_out, h, mean, rstd = ops.add_ln_modulate_fwd(x_in.cuda(), B, T, y=y.cuda(), gate=g, shift=sh, scale=sc, mod_ld=ML)
    cmp(x_out, xo, **tol(dtype))
    cmp(h, hr, **tol(dtype))
    dmod = torch.zeros(B, ML, device="cuda")
    dx, dy = ops.add_ln_modulate_bwd(dh.cuda(), x_out, mean, rstd, B, T, scale=sc, dx_in=dx_in.cuda(), y=y.cuda(),
                                     gate=g, mod_ld=ML, dshift=dmod[:, D:2 * D], dscale=dmod[:, 2 * D:3 * D],
                                     dgate=dmod[:, 0:D])
    cmp(dx, xo_grad_total, **tol(dtype))
    cmp(dy, gate.repeat_interleave(T, 0) * xo_grad_total, **tol(dtype))
    # per-sample modulation grads: dgate uses the *total* residual gradient
    dgate_ref = (xo_grad_total * y.float()).view(B, T, D).sum(1)
    t2 = dict(rtol=1e-3, atol=1e-3) if dtype == torch.float32 else dict(rtol=3e-2, atol=0.3)
    cmp(dmod[:, D:2 * D], modr.grad[:, D:2 * D], **t2)
    cmp(dmod[:, 2 * D:3 * D], modr.grad[:, 2 * D:3 * D], **t2)
    cmp(dmod[:, 0:D], dgate_ref, **t2)
    assert dmod[:, 3 * D:].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,B,T,with_y,affine", [(384, 17, 256, True, False), (768, 16, 259, True, False), (1152, 5, 1000, False, False),
                                                  (128, 33, 128, True, False), (1280, 4, 1027, False, True)])
def test_add_ln_modulate_fwd_four_rows_per_wave(D, B, T, with_y, affine, dtype, monkeypatch):
    """M >= 4096 rows, D a multiple of 128: the forward kernel that gives a row to 16 lanes (four rows per wave, DPP row sums).
    Against the fp32 reference, and against the one-row-per-wave kernel (UWU_LN_ROW16=0) on the same inputs -- the token
    counts make row groups straddle samples and leave a ragged last group (B * T not a multiple of 16)."""
    from uwudiff_amd import ops

    torch.manual_seed(3)
    M, ML = B * T, 3 * D + 8
    x_in = torch.randn(M, D).to(dtype)
    y = torch.randn(M, D).to(dtype) if with_y else None
    mod = torch.randn(1 if affine else B, ML) * 0.5
    md = mod.cuda()
    g, sh, sc = md[:, 0:D], md[:, D:2 * D], md[:, 2 * D:3 * D]
    rep = (lambda v: v.expand(B, -1).repeat_interleave(T, 0)) if affine else (lambda v: v.repeat_interleave(T, 0))
    xo = x_in.float()
    if with_y:
        xo = xo + rep(mod[:, 0:D]) * y.float()
        if dtype == torch.bfloat16:
            xo = xo.bfloat16().float()
    sc_r, sh_r = rep(mod[:, 2 * D:3 * D]), rep(mod[:, D:2 * D])
    hr = F.layer_norm(xo, (D,), eps=1e-6) * (sc_r if affine else 1 + sc_r) + sh_r
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("UWU_LN_ROW16", flag)
        kw = dict(shift=sh, scale=sc, mod_ld=0 if affine else ML, affine=affine)
        if with_y:
            kw.update(y=y.cuda(), gate=g)
        outs[flag] = ops.add_ln_modulate_fwd(x_in.cuda(), B, T, **kw)
    x_out, h, mean, rstd = outs["1"]
    if with_y:
        cmp(x_out, xo, **tol(dtype))
        assert torch.equal(x_out, outs["0"][0])
    cmp(h, hr, **tol(dtype))
    cmp(mean, xo.mean(1), rtol=1e-4, atol=1e-5)
    cmp(rstd, 1 / torch.sqrt(xo.var(1, unbiased=False) + 1e-6), rtol=1e-4, atol=1e-5)
    cmp(h, outs["0"][1], **tol(dtype))
    cmp(mean, outs["0"][2], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_plain_ln_no_residual(dtype):
    from uwudiff_amd import ops

    torch.manual_seed(1)
    B, T, D = 2, 32, 384
    x = torch.randn(B * T, D).to(dtype)
    _, h, mean, rstd = ops.add_ln_modulate_fwd(x.cuda(), B, T)
    cmp(h, F.layer_norm(x.float(), (D,), eps=1e-6), **tol(dtype))
    xr = x.float().requires_grad_(True)
    dh = torch.randn(B * T, D).to(dtype)
    (F.layer_norm(xr, (D,), eps=1e-6) * dh.float()).sum().backward()
    dx, dy = ops.add_ln_modulate_bwd(dh.cuda(), x.cuda(), mean, rstd, B, T)
    assert dy is None
    cmp(dx, xr.grad, **tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Tq,Tk,H,d,packed", [(2, 256, 256, 6, 64, True), (1, 64, 77, 2, 64, False),
                                                 (2, 100, 40, 3, 72, False), (1, 128, 128, 2, 32, True),
                                                 (3, 64, 64, 2, 64, True), (2, 128, 128, 3, 64, True),
                                                 (1, 192, 192, 1, 64, False), (1, 512, 512, 2, 64, True),
                                                 (1, 1024, 1024, 3, 64, True), (2, 320, 320, 2, 64, False),
                                                 (2, 1024, 77, 3, 64, False), (1, 256, 300, 2, 64, False), (2, 64, 13, 1, 64, False),
                                                 # head dims 72 (DiT-XL) and 128 on the MFMA kernels
                                                 (2, 256, 256, 4, 72, True), (1, 128, 77, 2, 72, False),
                                                 (1, 320, 320, 2, 72, False), (2, 256, 256, 2, 128, True),
                                                 (1, 64, 200, 1, 128, False)])
def test_attention_fwd_bwd(B, Tq, Tk, H, d, packed, dtype):
    from uwudiff_amd import ops

    if d == 128 and dtype == torch.float32:
        pytest.skip("head dim 128 exists on the bf16 MFMA kernels only")

    torch.manual_seed(0)
    D = H * d
    if packed:
        qkv = torch.randn(B * Tq, 3 * D).to(dtype)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        qkv_d = qkv.cuda()
        qd, kd, vd = qkv_d[:, :D], qkv_d[:, D:2 * D], qkv_d[:, 2 * D:]
    else:
        q, k, v = (torch.randn(B * Tq, D).to(dtype), torch.randn(B * Tk, D).to(dtype), torch.randn(B * Tk, D).to(dtype))
        qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    do = torch.randn(B * Tq, D).to(dtype)

    def heads(t, T):
        return t.float().reshape(B, T, H, d).transpose(1, 2)

    qr, kr, vr = [heads(t, T).detach().requires_grad_(True) for t, T in ((q, Tq), (k, Tk), (v, Tk))]
    orf = F.scaled_dot_product_attention(qr, kr, vr)
    orf.backward(heads(do, Tq))
    lse_ref = torch.logsumexp(qr.detach() @ kr.detach().transpose(-1, -2) / math.sqrt(d), dim=-1)

    o, lse = ops.attention_fwd(qd, kd, vd, B, Tq, Tk, H, d)
    cmp(o, orf.transpose(1, 2).reshape(B * Tq, D), **tol(dtype))
    cmp(lse, lse_ref, rtol=1e-4, atol=1e-4 if dtype == torch.float32 else 3e-2)
    if packed:
        dqkv = torch.empty_like(qkv_d)
        dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
    else:
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.attention_bwd(qd, kd, vd, o, do.cuda(), lse, dq, dk, dv, B, Tq, Tk, H, d)
    t = tol(dtype) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    cmp(dq, qr.grad.transpose(1, 2).reshape(B * Tq, D), **t)
    cmp(dk, kr.grad.transpose(1, 2).reshape(B * Tk, D), **t)
    cmp(dv, vr.grad.transpose(1, 2).reshape(B * Tk, D), **t)


@pytest.mark.parametrize("B,H,packed", [(1, 1, True), (41, 16, True), (43, 7, False), (192, 16, True)],
                         ids=["1head", "656heads", "301heads_unpacked", "xl2_bench_3072heads"])
def test_attention_p256_head_dim_72(B, H, packed, monkeypatch):
    """T = 256, head dim 72 (DiT-XL/2): the persistent LDS-DMA kernels carry columns 64..71 as 8-column tails (compact
    16-byte-row LDS images; a fifth k step of the score products, a third row tile of the transposed outputs).  Forward: the MFMA
    sequence per output element is the one of the kernel in attention_mfma.hip (UWU_ATTN_P256F_D72=0) -> o and lse BIT FOR BIT.
    Backward: delta = rowsum(dO * O) is summed in another order and rows 64..71 of dK^T / dV^T are accumulated per sub-tile, so
    dq / dk / dv agree with the key-block + dq kernels (UWU_ATTN_P256_D72=0) to 2 bf16 ulps of the largest gradient, and with an
    fp64 reference of the first heads as closely as those do.  Three launches each: the ring / prefetch pipeline must not depend
    on timing; head counts with unequal shares per workgroup (656 = 2.6 per CU, 301) and the XL/2 bench's 3072."""
    from uwudiff_amd import ops

    T, d = 256, 72
    D = H * d
    g = torch.Generator(device="cuda").manual_seed(200 + B)
    if packed:
        qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).bfloat16()
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:
        q, k, v = (torch.randn(B * T, D, device="cuda", generator=g).bfloat16() for _ in range(3))
    do = (torch.randn(B * T, D, device="cuda", generator=g) * 0.5).bfloat16()
    monkeypatch.setenv("UWU_ATTN_P256F", "1")  # the persistent forward at every head count (default: from 1024 heads)
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("UWU_ATTN_P256F_D72", flag)
        monkeypatch.setenv("UWU_ATTN_P256_D72", flag)
        for rep in range(3 if flag == "1" else 1):
            o, lse = ops.attention_fwd(q, k, v, B, T, T, H, d)
            if packed:  # (a gradient has the row stride of its tensor)
                dqkv = torch.full_like(qkv, float("nan"))
                dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
            else:
                dq, dk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
            ops.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, B, T, T, H, d)
            cur = (o, lse, dq, dk, dv)
            if flag == "1" and rep:
                for a, b in zip(outs["1"], cur):
                    assert torch.equal(a, b), "two launches of the persistent kernels differ"
            outs[flag] = cur
    assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])
    for a, b in zip(outs["0"][2:], outs["1"][2:]):
        assert torch.isfinite(b.float()).all()
        ulp = a.float().abs().max().item() * 2.0 ** -8
        assert (a.float() - b.float()).abs().max().item() <= 2 * ulp
    nb = min(B, 2)
    qr, kr, vr = (t[: nb * T].double().reshape(nb, T, H, d).transpose(1, 2).detach().requires_grad_(True) for t in (q, k, v))
    att = torch.softmax(qr @ kr.transpose(-1, -2) / math.sqrt(d), -1)
    (att @ vr).backward(do[: nb * T].double().reshape(nb, T, H, d).transpose(1, 2))
    for i, t in enumerate((qr, kr, vr)):
        ref = t.grad.transpose(1, 2).reshape(nb * T, D)
        e_old = (outs["0"][2 + i][: nb * T].double() - ref).abs().max().item()
        e_new = (outs["1"][2 + i][: nb * T].double() - ref).abs().max().item()
        assert e_new <= 1.25 * e_old + 1e-4, (i, e_new, e_old)


@pytest.mark.parametrize("B,H,packed", [(1, 1, True), (100, 6, True), (43, 7, False), (768, 6, True)],
                         ids=["1head", "600heads", "301heads_unpacked", "bench_4608heads"])
def test_attention_p256_kernels_are_bit_identical_to_the_per_head_kernels(B, H, packed, monkeypatch):
    """T = 256, head dim 64: the persistent LDS-DMA backward (attention_p256.hip: heads streamed through a ring, transposing LDS
    reads instead of transposed images, K / V fragments reloaded per head) issues the same MFMA sequence per output as the
    one-workgroup-per-head kernel (UWU_ATTN_P256=0), which the SDPA comparisons above pin -- so dq / dk / dv must agree BIT FOR
    BIT, for 1 head, for head counts that give workgroups unequal shares (600 = 2.3 per CU, 301) and for the bench's 4608.
    Three launches each: the ring / prefetch pipeline must not depend on timing.  Plus an fp64 check of the first heads."""
    from uwudiff_amd import ops

    T, d = 256, 64
    D = H * d
    g = torch.Generator(device="cuda").manual_seed(100 + B)
    if packed:
        qkv = torch.randn(B * T, 3 * D, device="cuda", generator=g).bfloat16()
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    else:
        q, k, v = (torch.randn(B * T, D, device="cuda", generator=g).bfloat16() for _ in range(3))
    do = (torch.randn(B * T, D, device="cuda", generator=g) * 0.5).bfloat16()
    # forward: the persistent kernel (K / V tiles through a six-slot LDS-DMA ring, V^T by transposing reads) against the
    # two-workgroups-per-head kernel (UWU_ATTN_P256F=0): same arithmetic per element -> o and lse bit for bit
    fo = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("UWU_ATTN_P256F", flag)
        for rep in range(3 if flag == "1" else 1):
            o, lse = ops.attention_fwd(q, k, v, B, T, T, H, d)
            if flag in fo:
                assert torch.equal(o, fo[flag][0]) and torch.equal(lse, fo[flag][1]), "launch-to-launch difference (forward)"
            fo[flag] = (o.clone(), lse.clone())
    assert torch.isfinite(fo["1"][0].float()).all() and torch.isfinite(fo["1"][1]).all()
    assert torch.equal(fo["1"][0], fo["0"][0]), float((fo["1"][0].float() - fo["0"][0].float()).abs().max())
    assert torch.equal(fo["1"][1], fo["0"][1])
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("UWU_ATTN_P256", flag)
        for rep in range(3 if flag == "1" else 1):
            if packed:
                dqkv = torch.full_like(qkv, float("nan"))
                dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
            else:
                dq, dk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
            ops.attention_bwd(q, k, v, o, do, lse, dq, dk, dv, B, T, T, H, d)
            got = (dq.clone(), dk.clone(), dv.clone())
            if flag in outs:
                assert all(torch.equal(x, y) for x, y in zip(got, outs[flag])), "launch-to-launch difference"
            outs[flag] = got
    for name, x, y in zip(("dq", "dk", "dv"), outs["1"], outs["0"]):
        assert torch.isfinite(x.float()).all(), name
        assert torch.equal(x, y), (name, float((x.float() - y.float()).abs().max()))
    # the first two samples against fp64 attention on the host
    nb = min(B, 2)
    def heads(t):
        return t[:nb * T].double().cpu().reshape(nb, T, H, d).transpose(1, 2)
    qr, kr, vr = [heads(t).requires_grad_(True) for t in (q, k, v)]
    F.scaled_dot_product_attention(qr, kr, vr).backward(heads(do))
    for x, ref in zip(outs["1"], (qr.grad, kr.grad, vr.grad)):
        ref = ref.transpose(1, 2).reshape(nb * T, D)
        err = (x[:nb * T].double().cpu() - ref).abs().max().item()
        assert err < 2e-2 * ref.abs().max().item() + 1e-3, err


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Tq,Tk,H,d", [(2, 64, 77, 2, 64), (2, 1024, 77, 3, 64), (1, 256, 256, 2, 64),
                                         (2, 128, 300, 1, 64), (2, 100, 40, 3, 72), (3, 64, 5, 1, 32),
                                         (2, 128, 77, 2, 72), (1, 64, 130, 1, 128)])
def test_attention_key_bias(B, Tq, Tk, H, d, dtype):
    """softmax(scale QK^T + key_bias[b,:]) V -- the reference's encoder_attention_mask path: (1-m)*-10000 per key,
    broadcast over heads and queries (rope_unet.py:106-114,448-453) -- plus a general finite bias."""
    from uwudiff_amd import ops

    if d == 128 and dtype == torch.float32:
        pytest.skip("head dim 128 exists on the bf16 MFMA kernels only")

    torch.manual_seed(1)
    D = H * d
    q, k, v = (torch.randn(B * Tq, D).to(dtype), torch.randn(B * Tk, D).to(dtype), torch.randn(B * Tk, D).to(dtype))
    do = torch.randn(B * Tq, D).to(dtype)
    keep = (torch.rand(B, Tk) > 0.4).float()
    keep[:, 0] = 1
    for bias in ((1 - keep) * -10000.0, torch.randn(B, Tk) * 2):
        def heads(t, T):
            return t.float().reshape(B, T, H, d).transpose(1, 2)

        qr, kr, vr = [heads(t, T).detach().requires_grad_(True) for t, T in ((q, Tq), (k, Tk), (v, Tk))]
        orf = F.scaled_dot_product_attention(qr, kr, vr, attn_mask=bias[:, None, None, :])
        orf.backward(heads(do, Tq))
        qd, kd, vd, bd = q.cuda(), k.cuda(), v.cuda(), bias.cuda()
        o, lse = ops.attention_fwd(qd, kd, vd, B, Tq, Tk, H, d, key_bias=bd)
        cmp(o, orf.transpose(1, 2).reshape(B * Tq, D), **tol(dtype))
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        ops.attention_bwd(qd, kd, vd, o, do.cuda(), lse, dq, dk, dv, B, Tq, Tk, H, d, key_bias=bd)
        for got, ref, T in ((dq, qr.grad, Tq), (dk, kr.grad, Tk), (dv, vr.grad, Tk)):
            ref = ref.transpose(1, 2).reshape(B * T, D)
            if dtype == torch.float32:
                cmp(got, ref, **tol(dtype))
            else:  # bf16 operands, sums over up to 1024 queries: bound the error against the tensor's scale
                err = (got.float().cpu() - ref).abs().max().item()
                assert err < 1.5e-2 * ref.abs().max().item() + 1e-3, (err, ref.abs().max().item())
    with pytest.raises(Exception):
        ops.attention_fwd(qd, kd, vd, B, Tq, Tk, H, d, key_bias=bd[:, :-1].contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(dtype):
    from uwudiff_amd import ops

    torch.manual_seed(0)
    x = torch.randint(-4, 5, (1000, 72)).float().to(dtype)
    out = ops.colsum(x.cuda())
    cmp(out, x.float().sum(0), rtol=0, atol=0)
    out = ops.colsum(x.cuda(), out=out, accumulate=True)
    cmp(out, 2 * x.float().sum(0), rtol=0, atol=0)


def test_embed_and_layout_kernels():
    from uwudiff_amd import lib as L

    torch.manual_seed(0)
    B, dim = 5, 256
    t = torch.tensor([0.0, 1.0, 250.5, 999.0, 37.0])
    out = torch.empty(B, dim, device="cuda")
    td = t.cuda()  # keep device temporaries alive: launches are asynchronous and take raw pointers
    L.call("uwu_timestep_embedding", L.ptr(td), B, dim, 10000.0, L.ptr(out), L.F32, L.stream())
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t[:, None] * f[None]
    cmp(out, torch.cat([torch.cos(a), torch.sin(a)], -1), rtol=1e-4, atol=2e-4)
    # silu fwd / bwd
    x = torch.randn(1024)
    y = torch.empty(1024, device="cuda")
    xd_, ones = x.cuda(), torch.ones(1024, device="cuda")
    L.call("uwu_silu_fwd", L.ptr(xd_), L.ptr(y), 1024, L.F32, L.stream())
    cmp(y, F.silu(x), rtol=1e-5, atol=1e-6)
    xr = x.clone().requires_grad_(True)
    F.silu(xr).sum().backward()
    dx = torch.empty(1024, device="cuda")
    L.call("uwu_silu_bwd", L.ptr(xd_), L.ptr(ones), L.ptr(dx), 1024, L.F32, L.stream())
    cmp(dx, xr.grad, rtol=1e-5, atol=1e-6)
    # patchify == unfold of a stride-p conv; unpatchify inverts it
    Bi, C, H, W, p = 2, 4, 8, 8, 2
    img = torch.randn(Bi, C, H, W)
    tok = torch.empty(Bi * (H // p) * (W // p), C * p * p, device="cuda")
    imgd = img.cuda()
    L.call("uwu_patchify", L.ptr(imgd), L.ptr(tok), Bi, C, H, W, p, L.F32, L.stream())
    ref = F.unfold(img, kernel_size=p, stride=p).transpose(1, 2).reshape(-1, C * p * p)
    cmp(tok, ref, rtol=0, atol=0)
    back = torch.empty(Bi, C, H, W, device="cuda")
    L.call("uwu_unpatchify", L.ptr(tok), L.F32, L.ptr(back), Bi, C, H, W, p, L.stream())
    cmp(back, img, rtol=0, atol=0)
    # add_pos
    T_, D = 16, 8
    xx = torch.randn(Bi * T_, D)
    pos = torch.randn(T_, D)
    xd = xx.cuda()
    posd = pos.cuda()
    L.call("uwu_add_pos", L.ptr(xd), L.ptr(posd), Bi, T_, D, L.F32, L.stream())
    cmp(xd, xx + pos.repeat(Bi, 1), rtol=0, atol=1e-7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,HW,C,silu", [(2, 64, 320, True), (3, 100, 640, False), (1, 37, 2560, True), (2, 1024, 960, True),
                                         (1, 4096, 64, True)])
def test_groupnorm_fwd_bwd(B, HW, C, silu, dtype):
    """uwu_groupnorm_fwd/bwd (channels-last, 32 groups) vs torch.nn.functional.group_norm (+ SiLU) in fp32."""
    from uwudiff_amd import ops

    torch.manual_seed(0)
    G = 32
    x = (torch.randn(B, HW, C) * 1.3 + 0.4).to(dtype)
    gamma, beta = torch.randn(C) * 0.5 + 1.0, torch.randn(C) * 0.3
    dy = torch.randn(B, HW, C).to(dtype)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.group_norm(xr.transpose(1, 2), G, gr, br, eps=1e-5).transpose(1, 2)
    if silu:
        yr = F.silu(yr)
    yr.backward(dy.float())
    xd = x.cuda().reshape(B * HW, C)
    y, mean, rstd = ops.groupnorm_fwd(xd, gamma.cuda(), beta.cuda(), B, HW, C, G, 1e-5, silu)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx = ops.groupnorm_bwd(dy.cuda().reshape(B * HW, C), xd, mean, rstd, gamma.cuda(), beta.cuda(), dg, db, B, HW, C, G, silu)
    t = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    cmp(y.reshape(B, HW, C), yr, **t)
    cmp(dx.reshape(B, HW, C), xr.grad, **t)
    tg = dict(rtol=1e-3, atol=1e-3 * HW ** 0.5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2 * (B * HW) ** 0.5)
    cmp(dg, gr.grad, **tg)
    cmp(db, br.grad, **tg)


def test_c_abi_rejects_bad_arguments_without_launching():
    """Host-side checks of the newer entry points: a bad shape / null pointer is a negative return code (UwuError with
    the library's message), never a device fault."""
    from uwudiff_amd import lib as L
    from uwudiff_amd import ops

    x = torch.randn(8 * 64, 36, device="cuda").bfloat16()  # C = 36: not a multiple of 8
    g, b = torch.ones(36, device="cuda"), torch.zeros(36, device="cuda")
    with pytest.raises(L.UwuError, match="groupnorm_fwd"):
        ops.groupnorm_fwd(x, g, b, 8, 64, 36, 4, 1e-5, True)
    a = torch.randn(2048, 64, device="cuda").bfloat16()
    w = torch.randn(2048, 32, device="cuda").bfloat16()
    dw = torch.zeros(64, 32, device="cuda")
    with pytest.raises(L.UwuError, match="gemm_wgrad"):  # leading dimension smaller than the row length
        L.call("uwu_gemm_wgrad", L.ptr(a), L.ptr(w), L.ptr(dw), None, 64, 32, 2048, 32, 32, 32, L.BF16, 256, None, 0,
               L.stream())
    with pytest.raises(L.UwuError, match="sampler_combine"):
        L.call("uwu_sampler_combine", L.ptr(dw), None, None, None, L.ptr(dw), dw.numel(), 1.0, 0.0, 0.0, 0.0, L.stream())
    with pytest.raises(L.UwuError, match="aggregate_concat"):
        L.call("uwu_aggregate_concat", L.ptr(dw), L.ptr(dw), L.ptr(dw), 2, 2, 24, 3, 0, L.stream())  # elem_size 3
    torch.cuda.synchronize()  # nothing was launched, nothing faulted
