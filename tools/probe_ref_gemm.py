"""Reference point only (NOT product code): vendor bf16 GEMM (hipBLASLt via torch.matmul) on the denoiser's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_kernels import timeit
M, D = 65536, 384
for name, (m, n, k, ta, tb) in {
    "qkv_fwd": (M, 3 * D, D, 0, 0), "proj_fwd": (M, D, D, 0, 0), "fc1_fwd": (M, 4 * D, D, 0, 0), "fc2_fwd": (M, D, 4 * D, 0, 0),
    "qkv_dgrad": (M, D, 3 * D, 0, 1), "fc2_dgrad": (M, 4 * D, D, 0, 1),
    "qkv_wgrad": (3 * D, D, M, 1, 1), "fc1_wgrad": (4 * D, D, M, 1, 1), "proj_wgrad": (D, D, M, 1, 1),
}.items():
    a = torch.randn((k, m) if ta else (m, k), device="cuda").bfloat16()
    b = torch.randn((k, n) if tb else (n, k), device="cuda").bfloat16()
    A = a.t() if ta else a
    Bm = b if tb else b.t()
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    us = timeit(lambda: torch.matmul(A, Bm, out=out))
    print(f"{name:10s} {us:9.1f} us  {2.0*m*n*k/us/1e6:8.1f} TFLOP/s   ({m}x{n}x{k})")
