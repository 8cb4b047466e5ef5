"""Which torch ops still launch kernels inside one DiT-S/2 training step (bench.py's step)?  torch.profiler over three steps, the
aten ops with device time, and the Python stack of each."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd.dit import DiT  # noqa: E402
from uwudiff_amd.objective import DiffusionLoss  # noqa: E402
from uwudiff_amd.optim import FusedAdamW, cosine_lr  # noqa: E402
from uwudiff_amd.scheduler import EulerDiscreteScheduler  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(1215)
model = DiT.from_config("DiT-S/2", cond_dim=1280, init="random", compute_dtype="bf16").to(dev)
loss_fn = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("stabilityai/stable-diffusion-xl-base-1.0", subfolder="scheduler"))
opt = FusedAdamW(model.parameters(), lr=1e-6, weight_decay=0.01, betas=(0.9, 0.999))
pool, pooled = torch.randn(4 * B, 4, 32, 32, device=dev), torch.randn(4 * B, 1280, device=dev)
model.flat.grad = torch.zeros_like(model.flat.data)


def step(i):
    x, c = pool[i * B:(i + 1) * B], pooled[i * B:(i + 1) * B]
    loss, _ = loss_fn(x, model, added_cond_kwargs={"text_embeds": c})
    loss.backward()
    opt.param_groups[0]["lr"] = cosine_lr(1e-6, i, 100_000, 1e-7)
    opt.step(zero_grad=True)


for i in range(2):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::ones_like",
                                                                      "aten::zeros", "aten::zeros_like", "aten::full", "aten::to", "aten::_to_copy"):
        if ev.device_time_total <= 0 and not any(k.device_time > 0 for k in ev.kernels):
            continue
        st = [f for f in (ev.stack or []) if "site-packages" not in f and "dist-packages" not in f][:5]
        key = (ev.name, tuple(st))
        seen[key] = seen.get(key, 0) + 1
for (name, st), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name}  shapes/stack:")
    for f in st:
        print("       ", f[:160])
