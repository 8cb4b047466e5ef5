"""Thin tensor-level wrappers over the C ABI (shape checks on the host, raw pointers to the kernels)."""
import torch

from . import lib as L


def gemm(a, b, *, trans_a=False, trans_b=False, bias=None, aux=None, epilogue=L.EPI_NONE, out=None, out2=None,
         c_dtype=None, split_k=1, M=None, N=None, K=None):
    """C = op(a) @ op(b)  with op(b) = b^T when trans_b is False (torch Linear weight layout [N,K])."""
    assert a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype
    if M is None:
        M, K_ = (a.shape[1], a.shape[0]) if trans_a else a.shape
        N, Kb = (b.shape[1], b.shape[0]) if trans_b else b.shape
        assert K_ == Kb, (a.shape, b.shape, trans_a, trans_b)
        K = K_
    c_dtype = c_dtype or a.dtype
    if out is None:
        out = (torch.zeros if epilogue == L.EPI_ACCUM else torch.empty)(M, N, device=a.device, dtype=c_dtype)
    if epilogue in (L.EPI_BIAS_GELU, L.EPI_BIAS_SILU) and out2 is None:
        out2 = torch.empty(M, N, device=a.device, dtype=c_dtype)
    L.call("uwu_gemm", L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(out2), L.ptr(bias), L.ptr(aux), M, N, K, a.stride(0),
           b.stride(0), out.stride(0), aux.stride(0) if aux is not None else 0, int(trans_a), int(trans_b), L.dt(a),
           L.dt(out), epilogue, split_k, L.stream())
    return (out, out2) if out2 is not None else out


def add_ln_modulate_fwd(x_in, B, T, *, y=None, gate=None, shift=None, scale=None, mod_ld=0, eps=1e-6, affine=False):
    M, D = x_in.shape
    x_out = torch.empty_like(x_in) if y is not None else x_in
    h = torch.empty_like(x_in)
    mean = torch.empty(M, device=x_in.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    L.call("uwu_add_ln_modulate_fwd", L.ptr(x_in), L.ptr(y), _p(gate), _p(shift), _p(scale), mod_ld, L.ptr(x_out),
           L.ptr(h), L.ptr(mean), L.ptr(rstd), B, T, D, eps, int(affine), L.dt(x_in), L.stream())
    return x_out, h, mean, rstd


def add_ln_modulate_fwd_q8(x_in, B, T, q_scale, *, shift, scale, mod_ld, y=None, gate=None, eps=1e-6, amax=None):
    """fp8 mode: the LayerNorm output as e4m3 bytes, row-major [M, D] and transposed [D, M] (no bf16 h)."""
    M, D = x_in.shape
    x_out = torch.empty_like(x_in) if y is not None else x_in
    q8 = torch.empty(M, D, device=x_in.device, dtype=torch.uint8)
    q8t = torch.empty(D, M, device=x_in.device, dtype=torch.uint8)
    mean = torch.empty(M, device=x_in.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    L.call("uwu_add_ln_modulate_fwd_q8", L.ptr(x_in), L.ptr(y), _p(gate), _p(shift), _p(scale), mod_ld, L.ptr(x_out), L.ptr(q8), D,
           L.ptr(q8t), M, L.ptr(q_scale), L.ptr(amax), L.ptr(mean), L.ptr(rstd), B, T, D, eps, L.stream())
    return x_out, q8, q8t, mean, rstd


def add_ln_modulate_bwd(dh, x, mean, rstd, B, T, *, scale=None, dx_in=None, y=None, gate=None, mod_ld=0,
                        dshift=None, dscale=None, dgate=None, affine=False):
    M, D = x.shape
    dx = torch.empty_like(x)
    dy = torch.empty_like(x) if y is not None else None
    L.call("uwu_add_ln_modulate_bwd", L.ptr(dh), L.ptr(x), L.ptr(mean), L.ptr(rstd), _p(scale), L.ptr(dx_in),
           L.ptr(y), _p(gate), mod_ld, L.ptr(dx), L.ptr(dy), _p(dshift), _p(dscale), _p(dgate), B, T, D, int(affine),
           L.dt(x), L.stream())
    return dx, dy


def _p(t):
    """Pointer of a possibly non-contiguous *view* whose rows are addressed through an explicit stride."""
    if t is None:
        return None
    if not t.is_cuda:
        raise L.UwuError("device tensor required")
    return t.data_ptr()


def _key_bias(key_bias, B, Tk):
    if key_bias.dtype != torch.float32 or tuple(key_bias.shape) != (B, Tk):
        raise L.UwuError(f"key_bias must be fp32 [B, Tk] = [{B}, {Tk}], got {key_bias.dtype} {tuple(key_bias.shape)}")
    return L.ptr(key_bias)


def attention_fwd(q, k, v, B, Tq, Tk, H, d, scale=None, key_bias=None):
    """q/k/v: 2-D views [B*T, >=H*d] with unit inner stride (e.g. column slices of a packed projection).
    key_bias: optional fp32 [B, Tk] added to the scaled scores of every head and query (encoder_attention_mask)."""
    scale = scale if scale is not None else d ** -0.5
    o = torch.empty(B * Tq, H * d, device=q.device, dtype=q.dtype)
    lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
    tail = (L.ptr(o), L.ptr(lse), B, Tq, Tk, H, d, q.stride(0), k.stride(0), v.stride(0), o.stride(0), scale, L.dt(q),
            L.stream())
    if key_bias is None:
        L.call("uwu_attention_fwd", _p(q), _p(k), _p(v), *tail)
    else:
        L.call("uwu_attention_bias_fwd", _p(q), _p(k), _p(v), _key_bias(key_bias, B, Tk), *tail)
    return o, lse


def attention_bwd(q, k, v, o, do, lse, dq, dk, dv, B, Tq, Tk, H, d, scale=None, key_bias=None):
    scale = scale if scale is not None else d ** -0.5
    # the C ABI has one leading dimension per (tensor, gradient) pair: a gradient laid out differently would be written out of bounds
    if (dq.stride(0), dk.stride(0), dv.stride(0), do.stride(0)) != (q.stride(0), k.stride(0), v.stride(0), o.stride(0)):
        raise ValueError("attention_bwd: dq / dk / dv / do must have the row strides of q / k / v / o")
    delta = torch.empty_like(lse)
    tail = (L.ptr(o), L.ptr(do), L.ptr(lse), L.ptr(delta), _p(dq), _p(dk), _p(dv), B, Tq, Tk, H, d, q.stride(0),
            k.stride(0), v.stride(0), o.stride(0), scale, L.dt(q), L.stream())
    if key_bias is None:
        L.call("uwu_attention_bwd", _p(q), _p(k), _p(v), *tail)
    else:
        L.call("uwu_attention_bias_bwd", _p(q), _p(k), _p(v), _key_bias(key_bias, B, Tk), *tail)
    return dq, dk, dv


def colsum(x, out=None, accumulate=False):
    M, N = x.shape
    if out is None:
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    L.call("uwu_colsum", L.ptr(x), L.dt(x), M, N, x.stride(0), L.ptr(out), int(accumulate), L.stream())
    return out


def colsum_batched(x, batch, M, N):
    """x: contiguous [batch*M, N] -> fp32 [batch, N] per-slab column sums."""
    out = torch.empty(batch, N, device=x.device, dtype=torch.float32)
    L.call("uwu_colsum_batched", L.ptr(x), L.dt(x), batch, M, N, N, L.ptr(out), 0, L.stream())
    return out


def im2col3x3(x, B, H, W, C, stride=1):
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    col = torch.empty(B * Ho * Wo, 9 * C, device=x.device, dtype=x.dtype)
    L.call("uwu_im2col3x3", L.ptr(x), L.ptr(col), B, H, W, C, stride, L.dt(x), L.stream())
    return col


def col2im3x3(dcol, B, H, W, C, stride=1):
    dx = torch.empty(B * H * W, C, device=dcol.device, dtype=dcol.dtype)
    L.call("uwu_col2im3x3", L.ptr(dcol), L.ptr(dx), B, H, W, C, stride, L.dt(dcol), L.stream())
    return dx


def conv3x3_implicit_ok(x, B, H, W, C, Cout, stride):
    return x.is_cuda and bool(L.load().uwu_conv3x3_implicit_ok(B, H, W, C, Cout, stride, L.dt(x)))


def conv3x3_fwd(x, w, bias, B, H, W, C, Cout, stride=1):
    """Implicit-GEMM 3x3 convolution: x [B*H*W, C] channels-last, w [Cout, 9*C] (tap-major), -> [B*Ho*Wo, Cout]."""
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty(B * Ho * Wo, Cout, device=x.device, dtype=x.dtype)
    L.call("uwu_conv3x3_fwd", L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), B, H, W, C, Cout, stride, L.dt(x), L.stream())
    return y


def conv3x3_dgrad(dy, w, B, H, W, C, Cout, stride=1):
    dx = torch.empty(B * H * W, C, device=dy.device, dtype=dy.dtype)
    L.call("uwu_conv3x3_dgrad", L.ptr(dy), L.ptr(w), L.ptr(dx), B, H, W, C, Cout, stride, L.dt(dy), L.stream())
    return dx


def conv3x3_wgrad(dy, x, dw, db, B, H, W, C, Cout, stride=1):
    """dw [Cout, 9*C] fp32 += , db [Cout] fp32 += (views of the flat gradient buffer)."""
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    need = L.load().uwu_conv3x3_wgrad_scratch_bytes(C, Cout, B * Ho * Wo)
    sc = shared_scratch(need, dy.device)
    L.call("uwu_conv3x3_wgrad", L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(db), B, H, W, C, Cout, stride, L.dt(dy), L.ptr(sc),
           sc.numel(), L.stream())


def groupnorm_fwd(x, gamma, beta, B, HW, C, G, eps, silu):
    y = torch.empty_like(x)
    mean = torch.empty(B * G, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    L.call("uwu_groupnorm_fwd", L.ptr(x), _p(gamma), _p(beta), L.ptr(y), L.ptr(mean), L.ptr(rstd), B, HW, C, G,
           float(eps), int(silu), L.dt(x), L.stream())
    return y, mean, rstd


def groupnorm_bwd(dy, x, mean, rstd, gamma, beta, dgamma, dbeta, B, HW, C, G, silu):
    dx = torch.empty_like(x)
    ws = torch.empty(2 * B * G, device=x.device, dtype=torch.float32)
    L.call("uwu_groupnorm_bwd", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), _p(gamma), _p(beta), L.ptr(dx),
           _p(dgamma), _p(dbeta), L.ptr(ws), B, HW, C, G, int(silu), L.dt(x), L.stream())
    return dx


def geglu_fwd(hg):
    M, F2 = hg.shape
    out = torch.empty(M, F2 // 2, device=hg.device, dtype=hg.dtype)
    L.call("uwu_geglu_fwd", L.ptr(hg), L.ptr(out), M, F2 // 2, L.dt(hg), L.stream())
    return out


def geglu_bwd(hg, dout):
    dhg = torch.empty_like(hg)
    L.call("uwu_geglu_bwd", L.ptr(hg), L.ptr(dout), L.ptr(dhg), hg.shape[0], hg.shape[1] // 2, L.dt(hg), L.stream())
    return dhg


def silu_fwd(x):
    y = torch.empty_like(x)
    L.call("uwu_silu_fwd", L.ptr(x), L.ptr(y), x.numel(), L.dt(x), L.stream())
    return y


def silu_bwd(x, dy):
    dx = torch.empty_like(x)
    L.call("uwu_silu_bwd", L.ptr(x), L.ptr(dy), L.ptr(dx), x.numel(), L.dt(x), L.stream())
    return dx


def add(a, b):
    out = torch.empty_like(a)
    L.call("uwu_add", L.ptr(a.contiguous()), L.ptr(b.contiguous()), L.ptr(out), a.numel(), L.dt(a), L.stream())
    return out


def add_rowvec(x, v, B, HW, C):
    out = x.clone()
    L.call("uwu_add_rowvec", L.ptr(out), L.ptr(v.contiguous()), B, HW, C, L.dt(x), L.stream())
    return out


def upsample2x(x, B, H, W, C, backward=False):
    n = B * H * W if backward else B * 4 * H * W
    out = torch.empty(n, C, device=x.device, dtype=x.dtype)
    L.call("uwu_upsample2x", L.ptr(x), L.ptr(out), B, H, W, C, int(backward), L.dt(x), L.stream())
    return out


def nchw_to_cl(x, dtype):
    B, C, H, W = x.shape
    out = torch.empty(B * H * W, C, device=x.device, dtype=dtype)
    L.call("uwu_nchw_to_cl", L.ptr(x), L.ptr(out), B, C, H * W, L.dt(out), L.stream())
    return out


def cl_to_nchw(x, B, C, HW):
    out = torch.empty(B, C, HW, device=x.device, dtype=torch.float32)
    L.call("uwu_cl_to_nchw", L.ptr(x), L.ptr(out), B, C, HW, L.dt(x), L.stream())
    return out


def gemm_wgrad(dy, x, dw, blocks=512, scratch=None, bias_grad=None):
    """dw[M,N] (fp32) += dy[K,M]^T @ x[K,N] -- weight gradient of a Linear; ``scratch`` (a byte tensor from
    :func:`gemm_wgrad_scratch`) switches the split-K reduction from fp32 atomics to slices + a reduce kernel;
    ``bias_grad`` (fp32[M]) += column sums of dy."""
    K, M = dy.shape
    K2, N = x.shape
    assert K == K2 and dw.shape == (M, N) and dw.dtype == torch.float32 and dy.dtype == x.dtype
    L.call("uwu_gemm_wgrad", L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(bias_grad), M, N, K, dy.stride(0), x.stride(0), dw.stride(0), L.dt(dy),
           blocks, L.ptr(scratch), scratch.numel() if scratch is not None else 0, L.stream())
    return dw


def gemm_wgrad_scratch(M, N, K, device="cuda"):
    return torch.empty(L.load().uwu_gemm_wgrad_scratch_bytes(M, N, K), dtype=torch.uint8, device=device)


_shared = {}


def shared_scratch(nbytes, device):
    """One growing split-K scratch per device, shared by every weight gradient of a Python-composed graph (launches on
    one stream are ordered, and a weight gradient's reduce kernel has consumed the scratch before the next one starts)."""
    sc = _shared.get(device)
    if sc is None or sc.numel() < nbytes:
        sc = _shared[device] = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
    return sc


def gemm_wgrad_shared(dy, x, dw, blocks=512, bias_grad=None):
    """gemm_wgrad with the per-device shared scratch: split-K slices + reduce instead of fp32 atomics (50 tiles x 8 slices
    of a 1280 x 1280 weight gradient are 52 MB of atomics at ~1.3 TB/s chip-wide -- as long as the K loop itself)."""
    K, M = dy.shape
    N = x.shape[1]
    need = L.load().uwu_gemm_wgrad_scratch_bytes(M, N, K)
    return gemm_wgrad(dy, x, dw, blocks=blocks, scratch=shared_scratch(need, dy.device) if need else None, bias_grad=bias_grad)


# ---------------------------------------------------------------------------------------------------- fp8 (config 5)
FP8_E4M3, FP8_E5M2 = 0, 1
FP8_MAX = {FP8_E4M3: 448.0, FP8_E5M2: 57344.0}


def fp8_amax(x, amax=None):
    if amax is None:
        amax = torch.zeros(1, device=x.device, dtype=torch.float32)
    L.call("uwu_fp8_amax", L.ptr(x), L.dt(x), x.numel(), L.ptr(amax), L.stream())
    return amax


def fp8_update_scales(amax, scale, fmt, margin=1.0):
    """amax / scale: fp32 [n]; fmt: int32 [n] (FP8_E4M3 / FP8_E5M2).  scale = FMT_MAX / (amax * margin); amax <- 0."""
    L.call("uwu_fp8_update_scales", L.ptr(amax), L.ptr(scale), L.ptr(fmt), amax.numel(), float(margin), L.stream())
    return scale


def fp8_quantize(x, scale, fmt=FP8_E4M3, *, rowmajor=True, transposed=False, amax=None, colsum=None):
    """x [M,K] (bf16 / fp32) -> (uint8 [M,K] | None, uint8 [K,M] | None) in one pass over x."""
    M, K = x.shape
    out = torch.empty(M, K, device=x.device, dtype=torch.uint8) if rowmajor else None
    out_t = torch.empty(K, M, device=x.device, dtype=torch.uint8) if transposed else None
    L.call("uwu_fp8_quantize", L.ptr(x), L.dt(x), M, K, x.stride(0), L.ptr(scale), fmt, L.ptr(out), K, L.ptr(out_t), M,
           L.ptr(amax), L.ptr(colsum), L.stream())
    return out, out_t


def gemm_fp8(a8, b8, scale_a, scale_b, *, fmt_a=FP8_E4M3, bias=None, aux=None, epilogue=L.EPI_NONE, out=None, out2=None):
    """C[M,N] = (a8[M,K] . b8[N,K]^T) / (scale_a * scale_b): both operands uint8 fp8 bytes, contraction-contiguous."""
    M, K = a8.shape
    N, Kb = b8.shape
    assert K == Kb and a8.dtype == torch.uint8 and b8.dtype == torch.uint8
    scratch, sbytes = None, 0
    if epilogue == L.EPI_ACCUM:
        if out is None:
            out = torch.zeros(M, N, device=a8.device, dtype=torch.float32)
        sbytes = L.load().uwu_gemm_fp8_scratch_bytes(M, N, K)
        scratch = torch.empty(sbytes, device=a8.device, dtype=torch.uint8)
    elif out is None:
        out = torch.empty(M, N, device=a8.device, dtype=torch.bfloat16)
    if epilogue == L.EPI_BIAS_GELU and out2 is None:
        out2 = torch.empty(M, N, device=a8.device, dtype=torch.bfloat16)
    L.call("uwu_gemm_fp8", L.ptr(a8), L.ptr(b8), L.ptr(out), L.ptr(out2), L.ptr(bias), L.ptr(aux), M, N, K, a8.stride(0),
           b8.stride(0), out.stride(0), aux.stride(0) if aux is not None else 0, fmt_a, epilogue, L.ptr(scale_a),
           L.ptr(scale_b), L.ptr(scratch), sbytes, L.stream())
    return (out, out2) if out2 is not None and epilogue == L.EPI_BIAS_GELU else out


def gemm_fp8_emit(a8, b8, scale_a, scale_b, q_scale, *, epilogue, fmt_a=FP8_E4M3, bias=None, aux=None, colsum=None, amax=None,
                  rowmajor=True, transposed=True):
    """The fp8 GEMM whose epilogue leaves its result as the fp8 operand of the next GEMMs (no quantising pass):
    EPI_BIAS_GELU -> (bf16 pre-activation, e4m3 gelu [M,N], its transpose [N,M]); EPI_DGELU -> (None, e5m2 [M,N], [N,M])."""
    M, K = a8.shape
    N = b8.shape[0]
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and b8.shape[1] == K
    C = torch.empty(M, N, device=a8.device, dtype=torch.bfloat16) if epilogue == L.EPI_BIAS_GELU else None
    q8 = torch.empty(M, N, device=a8.device, dtype=torch.uint8) if rowmajor else None
    q8t = torch.empty(N, M, device=a8.device, dtype=torch.uint8) if transposed else None
    L.call("uwu_gemm_fp8_emit", L.ptr(a8), L.ptr(b8), L.ptr(C), L.ptr(colsum), L.ptr(bias), L.ptr(aux), M, N, K, a8.stride(0),
           b8.stride(0), N, aux.stride(0) if aux is not None else 0, fmt_a, epilogue, L.ptr(scale_a), L.ptr(scale_b), L.ptr(q8), N,
           L.ptr(q8t), M, L.ptr(q_scale), L.ptr(amax), L.stream())
    return C, q8, q8t


# ---------------------------------------------------------------------------------------------------- RoPE in attention
def axial_rope_table(pos, fh, fw, H, d):
    """Factor table m [T, H*d] (fp32) of the reference's axial RoPE for shared positions pos [T, 2]."""
    T = pos.shape[0]
    tab = torch.empty(T, H * d, device=pos.device, dtype=torch.float32)
    L.call("uwu_axial_rope_table", L.ptr(pos), L.ptr(fh), L.ptr(fw), L.ptr(tab), T, H, d, H * d, L.stream())
    return tab


class _RopeAttentionFn(torch.autograd.Function):
    """o = SDPA(rope(q), rope(k), v) with the rotation applied inside the attention kernels' q / k staging."""

    @staticmethod
    def forward(ctx, qkv, pos, fh, fw, B, T, H, d):
        D = H * d
        tab = axial_rope_table(pos, fh, fw, H, d)
        o = torch.empty(B * T, D, device=qkv.device, dtype=qkv.dtype)
        lse = torch.empty(B, H, T, device=qkv.device, dtype=torch.float32)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        L.call("uwu_attention_rope_fwd", _p(q), _p(k), _p(v), L.ptr(tab), L.ptr(o), L.ptr(lse), B, T, H, d, qkv.stride(0),
               qkv.stride(0), qkv.stride(0), D, D, d ** -0.5, L.dt(qkv), L.stream())
        ctx.save_for_backward(qkv, pos, fh, fw, tab, o, lse)
        ctx.meta = (B, T, H, d)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, pos, fh, fw, tab, o, lse = ctx.saved_tensors
        B, T, H, d = ctx.meta
        D = H * d
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
        L.call("uwu_attention_rope_bwd", _p(q), _p(k), _p(v), L.ptr(tab), L.ptr(o), L.ptr(do), L.ptr(lse), _p(dq), _p(dk),
               _p(dv), B, T, H, d, qkv.stride(0), qkv.stride(0), qkv.stride(0), D, D, d ** -0.5, L.dt(qkv), L.stream())
        # dq', dk' (wrt the rotated operands) -> dq, dk in place, log-frequency gradients accumulated
        dfh, dfw = torch.zeros_like(fh), torch.zeros_like(fw)
        for x, dx in ((q, dq), (k, dk)):
            L.call("uwu_axial_rope_bwd_shared", _p(x), _p(dx), L.ptr(pos), T, L.ptr(fh), L.ptr(fw), _p(dx), L.ptr(dfh),
                   L.ptr(dfw), B * T, H, d, qkv.stride(0), L.dt(qkv), L.stream())
        return dqkv, None, dfh, dfw, None, None, None, None


def rope_attention(qkv, pos, fh, fw, B, T, H, d):
    """qkv: packed projection [B*T, 3*H*d] (bf16); pos [T, 2] fp32 (shared by the batch); fh / fw [H, d/4] log-frequencies."""
    return _RopeAttentionFn.apply(qkv, pos, fh, fw, B, T, H, d)


# ------------------------------------------------------------------------- fp32 Linears with <= 64 rows (conditioning path)
def skinny_linear_ok(M, N, K):
    return bool(L.load().uwu_skinny_linear_ok(M, N, K))


def skinny_linear_fwd(x, w, bias=None, epilogue=None):
    """y = x w^T (+ bias); with epilogue L.EPI_BIAS_SILU returns (y, silu(y)).  fp32, x [M <= 64, K], w [N, K]."""
    M, K = x.shape
    N = w.shape[0]
    epi = epilogue if epilogue is not None else (L.EPI_BIAS if bias is not None else L.EPI_NONE)
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    y2 = torch.empty_like(y) if epi == L.EPI_BIAS_SILU else None
    L.call("uwu_skinny_linear_fwd", L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), L.ptr(y2), M, N, K, epi, L.stream())
    return (y, y2) if y2 is not None else y


def skinny_linear_dgrad(dy, w):
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
    L.call("uwu_skinny_linear_dgrad", L.ptr(dy), L.ptr(w), L.ptr(dx), M, N, K, L.stream())
    return dx


def skinny_linear_wgrad(dy, x, dw, db=None):
    M, N = dy.shape
    K = x.shape[1]
    L.call("uwu_skinny_linear_wgrad", L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(db), M, N, K, L.stream())
    return dw
