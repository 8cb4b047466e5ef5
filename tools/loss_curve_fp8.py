"""DiT loss-curve overlay fp8 vs bf16: the same model trained from identical weights on identical
(latent, noise, timestep) draws with the block Linears in bf16 and in fp8 (delayed scaling).  Writes
gpurun_out/loss_curve_<model>_fp8_vs_bf16.csv and prints the largest relative deviation of the loss.

    python tools/loss_curve_fp8.py [--model DiT-XL/2] [--batch 32] [--steps 40] [--lr 1e-4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from uwudiff_amd.dit import DiT  # noqa: E402
from uwudiff_amd.objective import DiffusionLoss  # noqa: E402
from uwudiff_amd.optim import FusedAdamW  # noqa: E402
from uwudiff_amd.scheduler import EulerDiscreteScheduler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="DiT-XL/2")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--lr", type=float, default=1e-4)
a = ap.parse_args()

torch.manual_seed(0)
ref = DiT.from_config(a.model, cond_dim=1280, init="random", compute_dtype="bf16").cuda()
sd = ref.state_dict()
data = torch.randn(64, 4, 32, 32, device="cuda")
pooled = torch.randn(64, 1280, device="cuda")
g = torch.Generator().manual_seed(1)
draws = [(torch.randn(a.batch, 4, 32, 32, generator=g), torch.randint(0, 1000, (a.batch,), generator=g)) for _ in range(a.steps)]
curves = {}
for dtype in ("bf16", "fp8"):
    m = DiT.from_config(a.model, cond_dim=1280, init="random", compute_dtype=dtype).cuda()
    m.load_state_dict(sd)
    opt = FusedAdamW(m.parameters(), lr=a.lr, weight_decay=0.01)
    lf = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl"))
    m.flat.grad = torch.zeros_like(m.flat.data)
    out = []
    for i, (noise, t) in enumerate(draws):
        off = (i * a.batch) % (64 - a.batch + 1)
        lf.inject(noise=noise.cuda(), timesteps=t.cuda())
        m.flat.grad.zero_()
        loss, _ = lf(data[off:off + a.batch], m, added_cond_kwargs={"text_embeds": pooled[off:off + a.batch]})
        loss.backward()
        opt.step()
        out.append(float(loss))
    curves[dtype] = out
    del m, opt
os.makedirs("gpurun_out", exist_ok=True)
name = a.model.replace("/", "").replace("-", "_").lower()
path = f"gpurun_out/loss_curve_{name}_fp8_vs_bf16.csv"
dev = 0.0
with open(path, "w") as f:
    f.write("step,loss_bf16,loss_fp8,rel_dev\n")
    for i, (lb, l8) in enumerate(zip(curves["bf16"], curves["fp8"])):
        r = abs(l8 - lb) / abs(lb)
        dev = max(dev, r)
        f.write(f"{i},{lb:.6f},{l8:.6f},{r:.3e}\n")
print(f"{a.model} B={a.batch}: loss {curves['bf16'][0]:.4f} -> {curves['bf16'][-1]:.4f} (bf16), "
      f"{curves['fp8'][0]:.4f} -> {curves['fp8'][-1]:.4f} (fp8); max relative deviation {dev:.3e}; wrote {path}")
