"""Where a tile of the persistent 8-phase GEMM kernel spends its time: the timing build (UWU_P8_ABL=9) stamps the 100 MHz clock
at fixed points of every tile; this prints the median intervals over all workgroups and tiles.
Usage: python tools/probe_p8_stamps.py M N K"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 3072, 768)
stamps = torch.zeros(256, 16, 16, dtype=torch.int64, device="cuda")
os.environ["UWU_P8_STAMPS"] = hex(stamps.data_ptr())
os.environ["UWU_P8_ABL"] = "9"
from uwudiff_amd import ops  # noqa: E402

a = torch.randn(M, K, device="cuda").bfloat16()
b = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, b, out=out)
torch.cuda.synchronize()
s = stamps.cpu().double() * 0.01  # us
names = ["tile start", "K step 0 done", "K step 1 done", "K step 3 done", "K loop done", "(drain mode) request", "boundary done"]
for grp in (0, 1):
    g = s[:, :, 8 * grp:8 * grp + 7]
    ntile = int((g[0, :, 0] > 0).sum())
    print(f"wave group {grp}: {ntile} tiles per workgroup; medians over workgroups, tiles 1..{ntile - 2} (us)")
    mid = g[:, 1:ntile - 1]
    for a_, b_ in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 6)):
        d = (mid[:, :, b_] - mid[:, :, a_]).flatten()
        print(f"   {names[a_]:20s} -> {names[b_]:20s} {d.median():7.2f}   (p10 {d.quantile(0.1):6.2f}, p90 {d.quantile(0.9):6.2f})")
    tile = (g[:, 2:ntile, 0] - g[:, 1:ntile - 1, 0]).flatten()
    print(f"   whole tile {tile.median():7.2f}")
    t5 = g[:, min(5, ntile - 1), 4]
    print(f"   spread of 'K loop done' of tile 5 over workgroups: {t5.max() - t5.min():7.2f} us (std {t5.std():6.2f})")
