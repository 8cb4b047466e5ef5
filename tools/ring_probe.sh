#!/bin/bash
# Phase probe of the opt-in ring GEMM: full / no-store / no-MFMA / neither, per tile shape (UWU_GEMM_TILE).
cd "$(dirname "$0")/.."
for tile in ${TILES:-1 3 5}; do
  for dbg in ${DBGS:-0 1 2 3}; do
    echo "== TILE=$tile DEBUG=$dbg"
    UWU_GEMM_RING=1 UWU_GEMM_TILE=$tile UWU_GEMM_DEBUG=$dbg python tools/bench_kernels.py 256 gemm 2>&1 | grep -E "qkv_fwd|proj_fwd|fc1_fwd|fc2_fwd"
  done
done
