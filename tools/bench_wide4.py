"""192x384 GEMM kernel with 8 waves against the 160x384 four-wave form (UWU_GEMM_WIDE4=1: one wave per SIMD, accumulators in
AGPRs), same process, alternating; first a bit-identity check.   Usage: python tools/bench_wide4.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import lib as L  # noqa: E402
from uwudiff_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402


def setflag(k, v):
    os.environ[k] = v
    L.load().uwu_env_refresh()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    D = 384
    os.environ["UWU_GEMM_WIDE"] = "1"
    os.environ["UWU_GEMM_AS"] = "0"
    L.load().uwu_env_refresh()
    bf = torch.bfloat16
    for M in (160 * 13 + 48, B * 256):
        shapes = {"proj_fwd+bias": (M, D, D, 0, True), "qkv_fwd+bias": (M, 3 * D, D, 0, True), "fc2_fwd+bias": (M, D, 4 * D, 0, True),
                  "qkv_dgrad": (M, D, 3 * D, 1, False), "fc1_dgrad": (M, D, 4 * D, 1, False), "proj_dgrad": (M, D, D, 1, False)}
        for name, (m, n, k, tb, wb) in shapes.items():
            torch.manual_seed(1)
            a = torch.randn(m, k, device="cuda").to(bf)
            b = (torch.randn((k, n) if tb else (n, k), device="cuda") * 0.05).to(bf)
            bias = torch.randn(n, device="cuda") if wb else None
            out = {f: torch.full((m, n), 7.0, device="cuda", dtype=bf) for f in "01"}

            def run(f):
                ops.gemm(a, b, trans_b=bool(tb), bias=bias, epilogue=L.EPI_BIAS if wb else L.EPI_NONE, out=out[f])

            res = {}
            for rnd in range(3):
                for f in "01":
                    setflag("UWU_GEMM_WIDE4", f)
                    res.setdefault(f, []).append(timeit(lambda: run(f)))
            torch.cuda.synchronize()
            same = torch.equal(out["0"], out["1"])
            ref = (a.float() @ (b.float() if tb else b.float().t())) + (bias if wb else 0)
            err = ((out["1"].float() - ref).norm() / ref.norm()).item()
            fl = 2.0 * m * n * k
            print(f"M={m:7d} {name:14s} 8 waves {min(res['0']):7.1f} us   4 waves {min(res['1']):7.1f} us  "
                  f"({fl / min(res['1']) / 1e6:6.1f} TFLOP/s)  identical={same}  rel.err vs fp32 {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
