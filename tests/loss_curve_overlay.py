"""C1 loss-curve overlay: tiny UNet, 3x32x32 pixels, batch 16, eps objective with min-SNR + debias,
AdamW -- the HIP path vs the fp32 CPU oracle trained with torch.optim.AdamW from the same state_dict and the same injected
(noise, t) stream.  Writes profiles/r01_loss_curve_c1.csv and prints the max relative deviation."""
import csv
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loss as OL  # noqa: E402
from oracle.scheduler import EulerDiscreteScheduler as OSched  # noqa: E402
from oracle.unet import UNetOracle  # noqa: E402
from uwudiff_amd.objective import DiffusionLoss  # noqa: E402
from uwudiff_amd.optim import FusedAdamW  # noqa: E402
from uwudiff_amd.scheduler import EulerDiscreteScheduler  # noqa: E402
from uwudiff_amd.unet import TINY_UNET_CONFIG, UNet2DConditionModel  # noqa: E402


def main(steps=100, dtype="fp32", lr=2e-4, out=None):
    torch.manual_seed(1215)
    cfg = {k: v for k, v in TINY_UNET_CONFIG.items() if k != "sample_size"}
    ora = UNetOracle(**cfg)
    ora.init_weight()
    model = UNet2DConditionModel(cfg, compute_dtype=dtype).cuda()
    model.load_state_dict(ora.state_dict())
    oopt = torch.optim.AdamW(ora.parameters(), lr=lr, weight_decay=0.01)
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=0.01)
    kw = dict(use_snr_weight=True, use_debiased_estimation=True)
    lf = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl"), **kw)
    osch = OSched.sdxl()
    g = torch.Generator().manual_seed(7)
    B = 16
    data = torch.randn(50, 3, 32, 32, generator=g)
    ctx, pooled = torch.randn(B, 77, 2048, generator=g) * 0.5, torch.randn(B, 1280, generator=g) * 0.5
    ids = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B)
    cond = dict(encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": pooled, "time_ids": ids})
    cond_d = dict(encoder_hidden_states=ctx.cuda(), added_cond_kwargs={"text_embeds": pooled.cuda(), "time_ids": ids.cuda()})
    rows = []
    for step in range(steps):
        x = data[(step * B) % 34:(step * B) % 34 + B]
        noise = torch.randn(B, 3, 32, 32, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        o = OL.diffusion_loss(osch, x, noise, t, lambda n, tt: ora(n, tt, **cond)[0], **kw)
        oopt.zero_grad()
        o.loss.backward()
        oopt.step()
        lf.inject(noise=noise.cuda(), timesteps=t.cuda())
        if model.flat.grad is not None:
            model.flat.grad.zero_()
        loss, _ = lf(x.cuda(), model, **cond_d)
        loss.backward()
        opt.step()
        rows.append((step, float(o.loss), float(loss.detach())))
    dev = max(abs(a - b) / abs(a) for _, a, b in rows)
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["step", "loss_cpu_oracle_fp32", f"loss_hip_{dtype}"])
            w.writerows(rows)
    print(f"[overlay] {dtype}: {steps} steps, first {rows[0][1]:.5f}/{rows[0][2]:.5f}, last {rows[-1][1]:.5f}/{rows[-1][2]:.5f}, "
          f"max rel dev {dev:.2e}")
    return rows, dev


def main_dit(steps=60, dtype="bf16", lr=1e-4, B=16, out=None):
    """The same overlay on the headline denoiser (C2: DiT-S/2, 4x32x32 latents, pooled-text conditioning, eps
    objective): HIP (bf16 or fp32 compute) vs the fp32 CPU oracle from the same state_dict and injected draws."""
    from oracle.dit import DiTOracle
    from uwudiff_amd.dit import PRESETS, DiT, DiTConfig

    torch.manual_seed(1215)
    cfg = dict(PRESETS["DiT-S/2"], cond_dim=1280)
    ora = DiTOracle(**cfg)
    with torch.no_grad():
        for p in ora.parameters():
            p.copy_(torch.randn_like(p) * 0.02)
    model = DiT(DiTConfig(compute_dtype=dtype, **cfg)).cuda()
    model.load_state_dict(ora.state_dict())
    oopt = torch.optim.AdamW(ora.parameters(), lr=lr, weight_decay=0.01)
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=0.01)
    lf = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl"))
    osch = OSched.sdxl()
    g = torch.Generator().manual_seed(7)
    data = torch.randn(50, 4, 32, 32, generator=g)
    pooled = torch.randn(B, 1280, generator=g) * 0.5
    rows = []
    for step in range(steps):
        x = data[(step * B) % 34:(step * B) % 34 + B]
        noise = torch.randn(B, 4, 32, 32, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        o = OL.diffusion_loss(osch, x, noise, t, lambda n, tt: ora(n, tt, added_cond_kwargs={"text_embeds": pooled})[0])
        oopt.zero_grad()
        o.loss.backward()
        oopt.step()
        lf.inject(noise=noise.cuda(), timesteps=t.cuda())
        if model.flat.grad is not None:
            model.flat.grad.zero_()
        loss, _ = lf(x.cuda(), model, added_cond_kwargs={"text_embeds": pooled.cuda()})
        loss.backward()
        opt.step()
        rows.append((step, float(o.loss), float(loss.detach())))
    dev = max(abs(a - b) / abs(a) for _, a, b in rows)
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["step", "loss_cpu_oracle_fp32", f"loss_hip_{dtype}"])
            w.writerows(rows)
    print(f"[overlay DiT-S/2] {dtype}: {steps} steps, first {rows[0][1]:.5f}/{rows[0][2]:.5f}, last {rows[-1][1]:.5f}/"
          f"{rows[-1][2]:.5f}, max rel dev {dev:.2e}", flush=True)
    return rows, dev


if __name__ == "__main__":
    dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[2] == "dit":
        main_dit(60, dt, out=os.path.join(ROOT, "gpurun_out", f"loss_curve_c2_dit_s2_{dt}.csv"))
    else:
        main(100, dt, out=os.path.join(ROOT, "gpurun_out", f"loss_curve_c1_{dt}.csv"))
