import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd.dit import DiT
from uwudiff_amd.objective import DiffusionLoss
from uwudiff_amd.scheduler import EulerDiscreteScheduler
from uwudiff_amd.optim import FusedAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = DiT.from_config('DiT-S/2', cond_dim=1280, init='random').cuda()
loss_fn = DiffusionLoss(EulerDiscreteScheduler.from_pretrained('sdxl'))
opt = FusedAdamW(m.parameters(), lr=1e-6)
x = torch.randn(B, 4, 32, 32, device='cuda'); pooled = torch.randn(B, 1280, device='cuda')
def step():
    m.flat.grad.zero_() if m.flat.grad is not None else None
    loss, aux = loss_fn(x, m, added_cond_kwargs={'text_embeds': pooled})
    loss.backward()
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
N = 10
ev[0].record()
for _ in range(N): l = step()
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / N
print(f"B={B} step {ms:.3f} ms  {B/ms*1e3:.0f} img/s  loss {l.item():.4f}  {B*36.3e9/ms/1e9:.1f} TFLOP/s")
