#!/bin/bash
# fc1 + bias + GELU with fp8 emitted by the 8-phase kernel (UWU_GEMM_P8F_EMIT) against the LDS-staged emit epilogue: the launch alone
# at the DiT-XL/2 shape, then the whole fp8 step, alternating
cd "$(dirname "$0")/.."
cat > /tmp/emit_bench.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from uwudiff_amd import lib as L, ops
from tools.bench_kernels import timeit
M, N, K = 49152, 4608, 1152
a8 = torch.randint(0, 120, (M, K), device="cuda", dtype=torch.uint8)
b8 = torch.randint(0, 120, (N, K), device="cuda", dtype=torch.uint8)
bias = torch.randn(N, device="cuda")
one = torch.ones(1, device="cuda"); qs = torch.tensor([1e-3], device="cuda"); amax = torch.zeros(1, device="cuda")
for rep in range(3):
    for flag in ("0", "1"):
        os.environ["UWU_GEMM_P8F_EMIT"] = flag; L.load().uwu_env_refresh()
        us = timeit(lambda: ops.gemm_fp8_emit(a8, b8, one, one, qs, epilogue=L.EPI_BIAS_GELU, bias=bias, amax=amax))
        print(f"UWU_GEMM_P8F_EMIT={flag}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
PY
python /tmp/emit_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4_m28_emit.txt
cat gpurun_out/r4_m28_emit.txt
for rep in 1 2; do
for on in 1 0; do
  echo "== UWU_GEMM_P8F_EMIT=$on (rep $rep)"
  UWU_GEMM_P8F_EMIT=$on timeout -k 10 300 python bench.py --model DiT-XL/2 --batch 192 --dtype fp8 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --no-sweep 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], 'img/s', d['ms_per_step'], 'ms', 'final loss', d.get('final_loss'))
" || exit 1
done; done > gpurun_out/r4_m28_step.txt 2>&1
cat gpurun_out/r4_m28_step.txt
