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
