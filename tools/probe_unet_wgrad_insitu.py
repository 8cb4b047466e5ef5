"""Every GEMM-shaped launch of ONE SDXL-shape UNet step timed where it runs (no side stream, HIP events + a sync around each
call), grouped by (op, M, N, K): which shapes the weight-gradient family spends its time on inside the step, next to what
`probe_wgrad_unet.py` measures for the same shapes alone.
Usage: python tools/probe_unet_wgrad_insitu.py [batch] [latent]"""
import collections
import os
import sys

os.environ["UWU_UNET_FORK"] = "0"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uwudiff_amd import ops  # noqa: E402
from uwudiff_amd.objective import DiffusionLoss  # noqa: E402
from uwudiff_amd.scheduler import EulerDiscreteScheduler  # noqa: E402
from uwudiff_amd.unet import UNet2DConditionModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
torch.manual_seed(1215)
model = UNet2DConditionModel.from_config("sdxl", compute_dtype="bf16").to(dev)
loss_fn = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("stabilityai/stable-diffusion-xl-base-1.0", subfolder="scheduler"))
x = torch.randn(B, 4, S, S, device=dev)
pooled = torch.randn(B, 1280, device=dev)
ctx = torch.randn(B, 77, 2048, device=dev)
tid = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B, device=dev)
model.flat.grad = torch.zeros_like(model.flat.data)


def run():
    loss, _ = loss_fn(x, model, encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": pooled, "time_ids": tid})
    loss.backward()
    torch.cuda.synchronize()


run()
run()
stats = collections.defaultdict(lambda: [0, 0.0])
on = [False]


def wrap(name, key):
    fn = getattr(ops, name)

    def timed(*a, **kw):
        if not on[0]:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        r = fn(*a, **kw)
        e1.record()
        torch.cuda.synchronize()
        s = stats[(name,) + key(*a, **kw)]
        s[0] += 1
        s[1] += e0.elapsed_time(e1) * 1e3
        return r

    setattr(ops, name, timed)


def k_wgrad(dy, xx, out, **kw):  # dW[N, K] over M tokens
    return (dy.shape[1], xx.shape[1], dy.shape[0], "b" if kw.get("bias_grad") is not None else "")


def k_gemm(a, w, **kw):
    tb = bool(kw.get("trans_b"))
    return (a.shape[0], w.shape[1] if tb else w.shape[0], a.shape[1], "tb" if tb else f"e{kw.get('epilogue', 0)}")


def k_conv(dy, xx, *a, **kw):
    return tuple(int(v) for v in a[-6:] if isinstance(v, int)) + ("",)


wrap("gemm_wgrad", k_wgrad)
wrap("gemm", k_gemm)
for nm in ("conv3x3_wgrad", "conv3x3_dgrad", "conv3x3_fwd"):
    if hasattr(ops, nm):
        wrap(nm, k_conv)
on[0] = True
run()
on[0] = False
tot = sum(v[1] for v in stats.values())
print(f"batch {B}, {S}x{S} latents: {tot / 1e3:.1f} ms in {sum(v[0] for v in stats.values())} timed launches (one step)")
for k, (n, us) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    fl = ""
    if k[0] in ("gemm_wgrad", "gemm"):
        fl = f"{2.0 * k[1] * k[2] * k[3] * n / us / 1e6:7.1f} TFLOP/s"
    print(f"{100 * us / tot:5.1f}%  {k[0]:18s} {str(k[1:]):44s} x{n:4d}  {us / n:8.1f} us  {fl}")
