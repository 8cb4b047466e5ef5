import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops
from tools.bench_kernels import timeit
M, N, K = 65536, 1152, 384
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); o32 = torch.empty(M, N, device="cuda")
print("bf16 out", timeit(lambda: ops.gemm(a, b, out=o16)))
print("fp32 out", timeit(lambda: ops.gemm(a, b, out=o32, c_dtype=torch.float32)))
x = torch.empty(M * N, device="cuda", dtype=torch.bfloat16)
print("torch fill 151MB", timeit(lambda: x.fill_(1.0)))
y = torch.empty(M * N, device="cuda", dtype=torch.bfloat16)
print("torch copy 151MB", timeit(lambda: y.copy_(x)))
