#!/bin/bash
# Phase probe of the panel GEMM (UWU_GEMM_DEBUG bits: 1 no epilogue, 2 no MFMA, 4 no A DMA, 8 no B DMA, 16 rotate B tile)
cd "$(dirname "$0")/.."
for dbg in ${PROBE_LIST:-0 1 2 3 7 11 19}; do
  echo "== DEBUG=$dbg"
  UWU_GEMM_DEBUG=$dbg python tools/bench_kernels.py 256 gemm 2>&1 | grep -E "qkv_fwd|proj_fwd|fc1_fwd|fc2_fwd"
done
