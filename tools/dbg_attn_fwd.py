import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import lib as L, ops
B, T, H, d = 2, 256, 6, 64
D = H * d
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * D, device="cuda").bfloat16()
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
outs = {}
for flag in ("0", "1"):
    os.environ["UWU_ATTN_P256F"] = flag
    L.load().uwu_env_refresh()
    o, lse = ops.attention_fwd(q, k, v, B, T, T, H, d)
    torch.cuda.synchronize()
    outs[flag] = (o.float().cpu(), lse.cpu())
bad = (outs["0"][0] != outs["1"][0])
print("wrong o elements", int(bad.sum()), "of", bad.numel(), " lse wrong", int((outs["0"][1] != outs["1"][1]).sum()))
idx = bad.nonzero()
import collections
print("rows mod 32:", sorted(collections.Counter((idx[:, 0] % 32).tolist()).items()))
print("token//32 (wave):", sorted(collections.Counter(((idx[:, 0] % 256) // 32).tolist()).items()))
print("d (col mod 64):", sorted(collections.Counter((idx[:, 1] % 64).tolist()).items()))
print("head:", sorted(collections.Counter((idx[:, 1] // 64).tolist()).items()))
print("sample:", sorted(collections.Counter((idx[:, 0] // 256).tolist()).items()))
