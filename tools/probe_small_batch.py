"""Where a small-batch DiT-S/2 step spends its time: forward only / forward + backward / whole step, ms (wall, many reps).
Usage: [UWU_DIT_FORK=0] python tools/probe_small_batch.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from duwu.loss import DiffusionLoss  # noqa: E402
from uwudiff_amd.dit import DiT  # noqa: E402
from uwudiff_amd.optim import FusedAdamW  # noqa: E402
from uwudiff_amd.scheduler import EulerDiscreteScheduler  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = DiT.from_config("DiT-S/2", cond_dim=1280, init="random", compute_dtype="bf16").to(dev)
loss_fn = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("stabilityai/stable-diffusion-xl-base-1.0", subfolder="scheduler"))
opt = FusedAdamW(model.parameters(), lr=1e-6, weight_decay=0.01)
x = torch.randn(B, 4, 32, 32, device=dev)
c = torch.randn(B, 1280, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
model.flat.grad = torch.zeros_like(model.flat.data)


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fwd():
    with torch.no_grad():
        return model(x, t, added_cond_kwargs={"text_embeds": c})[0]


def fwd_bwd():
    loss, _ = loss_fn(x, model, added_cond_kwargs={"text_embeds": c})
    loss.backward()


def full():
    model.flat.grad.zero_()
    fwd_bwd()
    opt.step()


print(f"batch {B}: forward {timeit(fwd):.3f} ms | forward + loss + backward {timeit(fwd_bwd):.3f} ms | whole step {timeit(full):.3f} ms")
