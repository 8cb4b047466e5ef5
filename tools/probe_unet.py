"""SDXL-shape UNet forward+backward+AdamW step timing on the GPU (4x32x32 latents, 77x2048 context)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd.unet import UNet2DConditionModel
from uwudiff_amd.objective import DiffusionLoss
from uwudiff_amd.scheduler import EulerDiscreteScheduler
from uwudiff_amd.optim import FusedAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(0)
t0 = time.time()
m = UNet2DConditionModel.from_config("sdxl").cuda()
print(f"built SDXL UNet: {sum(v.numel() for _, v in m.named_tensors())/1e6:.1f} M params in {time.time()-t0:.1f}s", flush=True)
lf = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl"))
opt = FusedAdamW(m.parameters(), lr=1e-6)
x = torch.randn(B, 4, S, S, device="cuda")
kw = dict(encoder_hidden_states=torch.randn(B, 77, 2048, device="cuda"),
          added_cond_kwargs={"text_embeds": torch.randn(B, 1280, device="cuda"),
                             "time_ids": torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B, device="cuda")})
def step():
    if m.flat.grad is not None: m.flat.grad.zero_()
    loss, _ = lf(x, m, **kw); loss.backward(); opt.step(); return loss
for i in range(2):
    l = step(); torch.cuda.synchronize(); print("warm", i, float(l), flush=True)
t0 = time.time(); N = 3
for _ in range(N): l = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / N
fl = 1283e9 if S == 32 else 20284e9
print(f"B={B} S={S}: {dt*1e3:.1f} ms/step, {B/dt:.2f} img/s, {B*fl/dt/1e12:.1f} TFLOP/s, loss {float(l):.4f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
if os.environ.get("UWU_PROBE_GRAPH", "1") == "1":
    # the same forward+backward replayed from a hipGraph (torch.cuda.CUDAGraph around the ctypes launches): the python
    # composition of ~4000 launches per step is host-bound, the replay is not
    def fwd_bwd():
        m.flat.grad.zero_()
        loss, _ = lf(x, m, **kw); loss.backward(); return loss
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sl = fwd_bwd()
    for _ in range(2):
        g.replay(); opt.step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(N):
        g.replay(); opt.step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / N
    print(f"graph replay B={B} S={S}: {dt*1e3:.1f} ms/step, {B/dt:.2f} img/s, {B*fl/dt/1e12:.1f} TFLOP/s, loss {float(sl.detach()):.4f}")
