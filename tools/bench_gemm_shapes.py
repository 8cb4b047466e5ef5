"""Per-shape GEMM timings for one model width: forward / input-gradient / weight-gradient launches of the four block Linears.
Usage: python tools/bench_gemm_shapes.py D B [vendor]     (M = 256 B tokens; `vendor` adds torch.matmul = hipBLASLt)
Env: the library's A/B switches (UWU_GEMM_P8=0 ...); UWU_BENCH_ONLY=name,... selects shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import lib as L  # noqa: E402
from uwudiff_amd import ops  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    vendor = "vendor" in sys.argv[3:]
    M = B * 256
    bf, dev = torch.bfloat16, "cuda"
    torch.manual_seed(0)
    only = os.environ.get("UWU_BENCH_ONLY")
    shapes = {
        "qkv_fwd": (M, 3 * D, D, 0, 0), "proj_fwd": (M, D, D, 0, 0), "fc1_fwd": (M, 4 * D, D, 0, 0), "fc2_fwd": (M, D, 4 * D, 0, 0),
        "qkv_dgrad": (M, D, 3 * D, 0, 1), "proj_dgrad": (M, D, D, 0, 1), "fc1_dgrad": (M, D, 4 * D, 0, 1), "fc2_dgrad": (M, 4 * D, D, 0, 1),
        "qkv_wgrad": (3 * D, D, M, 1, 1), "proj_wgrad": (D, D, M, 1, 1), "fc1_wgrad": (4 * D, D, M, 1, 1), "fc2_wgrad": (D, 4 * D, M, 1, 1),
    }
    tot = {}
    for name, (m, n, k, ta, tb) in shapes.items():
        if only and name not in only.split(","):
            continue
        a = torch.randn((k, m) if ta else (m, k), device=dev).to(bf)
        b = torch.randn((k, n) if tb else (n, k), device=dev).to(bf)
        fl = 2.0 * m * n * k
        line = f"{name:11s} ({m}x{n}x{k})"
        if ta:
            out = torch.zeros(m, n, device=dev)
            scratch = ops.gemm_wgrad_scratch(m, n, k)
            us = timeit(lambda: ops.gemm_wgrad(a, b, out, blocks=512, scratch=scratch))
        else:
            out = torch.empty(m, n, device=dev, dtype=bf)
            us = timeit(lambda: ops.gemm(a, b, trans_b=bool(tb), out=out))
            if name == "fc1_fwd":
                bias, out2 = torch.randn(n, device=dev), torch.empty(m, n, device=dev, dtype=bf)
                ug = timeit(lambda: ops.gemm(a, b, bias=bias, epilogue=L.EPI_BIAS_GELU, out=out, out2=out2))
                print(f"{'fc1+gelu':11s} ({m}x{n}x{k}) {ug:9.1f} us {fl / ug / 1e6:8.1f} TFLOP/s")
                tot["fc1+gelu"] = (ug, fl)
            if name == "fc2_dgrad":
                u, cs = torch.randn(m, n, device=dev).to(bf), torch.zeros(n, device=dev)
                ug = timeit(lambda: ops.gemm(a, b, trans_b=True, aux=u, epilogue=L.EPI_DGELU, out=out, out2=cs))
                print(f"{'fc2dg+dgelu':11s} ({m}x{n}x{k}) {ug:9.1f} us {fl / ug / 1e6:8.1f} TFLOP/s")
                tot["fc2dg+dgelu"] = (ug, fl)
        line += f" {us:9.1f} us {fl / us / 1e6:8.1f} TFLOP/s"
        if vendor:
            if ta:
                uv = timeit(lambda: torch.matmul(a.t(), b))
            else:
                bb = b if tb else b.t()
                uv = timeit(lambda: torch.matmul(a, bb))
            line += f"   vendor {uv:9.1f} us {fl / uv / 1e6:8.1f} TFLOP/s"
        print(line, flush=True)
        tot[name] = (us, fl)
    # the step's launches: fc1 forward runs with GELU, fc2 dgrad with dGELU
    step = [k for k in tot if k not in ("fc1_fwd", "fc2_dgrad")] if "fc1+gelu" in tot else list(tot)
    us = sum(tot[k][0] for k in step)
    fl = sum(tot[k][1] for k in step)
    print(f"block total (step launches) {us:9.1f} us {fl / us / 1e6:8.1f} TFLOP/s = {fl / us / 1e6 / 2500:.3f} of 2.5 PF")


if __name__ == "__main__":
    main()
