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


def main_dit(steps=60, dtype="bf16", lr=1e-4, B=16, out=None, preset="DiT-S/2", depth=None):
    """The same overlay on the headline denoiser (C2: DiT-S/2, 4x32x32 latents, pooled-text conditioning, eps
    objective): HIP (bf16, fp8 or fp32 compute) vs the fp32 CPU oracle from the same state_dict and injected draws.
    ``preset`` / ``depth``: the other BASELINE DiT configs at their exact width with the depth cut (DiT-B/2, DiT-XL/2 fp8)."""
    from oracle.dit import DiTOracle
    from uwudiff_amd.dit import PRESETS, DiT, DiTConfig

    torch.manual_seed(1215)
    cfg = dict(PRESETS[preset], cond_dim=1280)
    if depth is not None:
        cfg["depth"] = depth
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
    print(f"[overlay {preset}{'' if depth is None else ' depth ' + str(depth)}] {dtype}: {steps} steps, first {rows[0][1]:.5f}/{rows[0][2]:.5f}, last {rows[-1][1]:.5f}/"
          f"{rows[-1][2]:.5f}, max rel dev {dev:.2e}", flush=True)
    return rows, dev


def main_unet_sdxl_width(steps=6, dtype="bf16", lr=1e-4, B=1, out=None, latent=32):
    """The overlay on the SDXL-WIDTH UNet (BASELINE config 4's widths 320 / 640 / 1280, heads 5 / 10 / 20, 77 x 2048 context,
    text_time conditioning; transformer depth cut to 1 / 2 / 2 as in tests/test_configs_gpu.py) at 4x32x32 latents: per-step loss
    of the HIP path against the fp32 CPU oracle (oracle/unet.py) under AdamW from the same weights and injected draws."""
    from tests.test_configs_gpu import SDXL_CUT, unet_models

    ora, model = unet_models(dtype, seed=3)
    oopt = torch.optim.AdamW(ora.parameters(), lr=lr, weight_decay=0.01)
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=0.01)
    lf = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl"))
    osch = OSched.sdxl()
    g = torch.Generator().manual_seed(11)
    data = torch.randn(8, 4, latent, latent, generator=g)
    ctx, pooled = torch.randn(B, 77, 2048, generator=g) * 0.5, torch.randn(B, 1280, generator=g) * 0.5
    ids = torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * B)
    cond = dict(encoder_hidden_states=ctx, added_cond_kwargs={"text_embeds": pooled, "time_ids": ids})
    cond_d = dict(encoder_hidden_states=ctx.cuda(), added_cond_kwargs={"text_embeds": pooled.cuda(), "time_ids": ids.cuda()})
    rows = []
    for step in range(steps):
        x = data[(step * B) % 7:(step * B) % 7 + B]
        noise = torch.randn(B, 4, latent, latent, generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        o = OL.diffusion_loss(osch, x, noise, t, lambda n, tt: ora(n, tt, **cond)[0])
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
    print(f"[overlay SDXL-width UNet] {dtype}: {steps} steps, first {rows[0][1]:.5f}/{rows[0][2]:.5f}, last {rows[-1][1]:.5f}/"
          f"{rows[-1][2]:.5f}, max rel dev {dev:.2e}", flush=True)
    return rows, dev


if __name__ == "__main__":
    dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[2] == "configs345":  # the measured runs behind the tolerances of tests/test_train_gpu.py
        main_dit(12, "bf16", B=8, preset="DiT-B/2", depth=2, out=os.path.join(ROOT, "gpurun_out", "loss_curve_dit_b2_d2_bf16.csv"))
        main_dit(12, "fp8", B=8, preset="DiT-XL/2", depth=2, out=os.path.join(ROOT, "gpurun_out", "loss_curve_dit_xl2_d2_fp8.csv"))
        main_dit(12, "bf16", B=8, preset="DiT-XL/2", depth=2, out=os.path.join(ROOT, "gpurun_out", "loss_curve_dit_xl2_d2_bf16.csv"))
        main_unet_sdxl_width(8, "bf16", out=os.path.join(ROOT, "gpurun_out", "loss_curve_unet_sdxl_width_bf16.csv"))
    elif len(sys.argv) > 2 and sys.argv[2] == "dit":
        main_dit(60, dt, out=os.path.join(ROOT, "gpurun_out", f"loss_curve_c2_dit_s2_{dt}.csv"))
    else:
        main(100, dt, out=os.path.join(ROOT, "gpurun_out", f"loss_curve_c1_{dt}.csv"))
