"""Sampling loop.

CPU: the oracle restatement and the product's host-side schedule logic vs the REFERENCE's own
``sampling/k_diffusion_wrapper.py`` / ``sampling/get_sigmas.py`` outputs (tests/golden/kdiff_schedule.npz).
GPU: the fused CFG + Euler-ancestral step and a whole guided sampling run of a small DiT vs the CPU oracle loop."""
import numpy as np
import pytest
import torch

from tests.golden_util import load


def test_schedule_helpers_match_reference_golden():
    from oracle import sampling as OS
    from oracle.scheduler import EulerDiscreteScheduler
    from uwudiff_amd.sampling import DiscreteEpsDDPMDenoiser, get_sigmas_for_rf

    _, d = load("kdiff_schedule")
    abar = EulerDiscreteScheduler.sdxl().alphas_cumprod
    sig = OS.sigmas_from_alphas_cumprod(abar)
    den = DiscreteEpsDDPMDenoiser(None, abar)
    for s2t, t2s, gs in ((lambda s: OS.sigma_to_t(sig.log(), s), lambda t: OS.t_to_sigma(sig.log(), t),
                          lambda n: OS.get_sigmas(sig, n)),
                         (den.sigma_to_t, den.t_to_sigma, den.get_sigmas)):
        # fp32 log/exp differ in the last bit between host CPUs: 1e-6 relative (t runs up to 999)
        torch.testing.assert_close(s2t(d["probe_sigmas"]), d["sigma_to_t"], rtol=1e-6, atol=1e-4)
        torch.testing.assert_close(t2s(d["t_probe"]), d["t_to_sigma"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(gs(20), d["get_sigmas_20"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(gs(None), d["get_sigmas_all"], rtol=1e-6, atol=1e-7)
    assert abs(float(den.sigma_max) - 14.6146) < 5e-5  # configs/sampling/demo_sampling.yaml:49
    for f in (OS.get_sigmas_for_rf, get_sigmas_for_rf):
        np.testing.assert_allclose(f(16, float(den.sigma_max)), d["rf_sigmas_16"].numpy(), rtol=1e-12)
        np.testing.assert_allclose(f(8, 14.6146, 0.03), d["rf_sigmas_8_min"].numpy(), rtol=1e-12)


@pytest.mark.gpu
def test_sampler_step_kernel():
    from oracle.sampling import get_ancestral_step
    from uwudiff_amd import lib as L

    torch.manual_seed(0)
    n = 4 * 4 * 16 * 16
    x, ec, eu, nz = [torch.randn(n) for _ in range(4)]
    s, sn, cfg, s_noise = 3.7, 2.1, 5.0, 1.0
    sd, su = get_ancestral_step(s, sn, 1.0)
    eps = eu + (ec - eu) * cfg
    den = x - s * eps
    ref = x + ((x - den) / s) * (sd - s) + nz * s_noise * su
    xd, ecd, eud, nzd = x.cuda(), ec.cuda(), eu.cuda(), nz.cuda()
    out, dn = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    L.call("uwu_sampler_step", L.ptr(xd), L.ptr(ecd), L.ptr(eud), L.ptr(nzd), L.ptr(out), L.ptr(dn), n, cfg, s, sd, su,
           s_noise, L.stream())
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dn.cpu(), den, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_guided_sampling_run_matches_cpu_oracle():
    from oracle import sampling as OS
    from oracle.dit import DiTOracle
    from oracle.scheduler import EulerDiscreteScheduler
    from uwudiff_amd.dit import DiT, DiTConfig
    from uwudiff_amd.sampling import DiscreteEpsDDPMDenoiser, sample_euler_ancestral

    torch.manual_seed(0)
    cfg = dict(depth=2, hidden=128, heads=2, patch=2, sample_size=16, in_channels=4, out_channels=4, cond_dim=32)
    ora = DiTOracle(**cfg)
    with torch.no_grad():
        for p in ora.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    model = DiT(DiTConfig(compute_dtype="fp32", **cfg)).cuda()
    model.load_state_dict(ora.state_dict())
    abar = EulerDiscreteScheduler.sdxl().alphas_cumprod
    den = DiscreteEpsDDPMDenoiser(model, abar)
    B, steps = 2, 6
    sigmas = den.get_sigmas(steps)
    x0 = torch.randn(B, 4, 16, 16) * float(sigmas[0])
    noises = [torch.randn(B, 4, 16, 16) for _ in range(steps)]
    pc, pu = torch.randn(B, 32), torch.zeros(B, 32)
    ref = OS.sample_euler_ancestral_cfg(ora, x0, sigmas, den.log_sigmas, {"added_cond_kwargs": {"text_embeds": pc}},
                                        {"added_cond_kwargs": {"text_embeds": pu}}, 4.0, noises)
    out = sample_euler_ancestral(den, x0.cuda(), sigmas, {"added_cond_kwargs": {"text_embeds": pc.cuda()}},
                                 {"added_cond_kwargs": {"text_embeds": pu.cuda()}}, cfg=4.0,
                                 noise_sampler=lambda i: noises[i].cuda())
    err = ((out.cpu() - ref).norm() / ref.norm()).item()
    assert err < 1e-3, err


@pytest.mark.gpu
def test_sampler_combine_kernel():
    from uwudiff_amd import lib as L

    torch.manual_seed(1)
    n = 2 * 4 * 16 * 16
    base, ec, eu, nz = [torch.randn(n) for _ in range(4)]
    cfg, a, b, c = 4.5, -3.2, 1.7, 0.6
    bd, ecd, eud, nzd = base.cuda(), ec.cuda(), eu.cuda(), nz.cuda()
    out = torch.empty(n, device="cuda")
    L.call("uwu_sampler_combine", L.ptr(bd), L.ptr(ecd), L.ptr(eud), L.ptr(nzd), L.ptr(out), n, cfg, a, b, c, L.stream())
    torch.testing.assert_close(out.cpu(), base + a * (eu + (ec - eu) * cfg) + b * eu + c * nz, rtol=1e-5, atol=1e-5)
    L.call("uwu_sampler_combine", L.ptr(bd), L.ptr(ecd), None, None, L.ptr(out), n, cfg, a, b, c, L.stream())
    torch.testing.assert_close(out.cpu(), base + (a + b) * ec, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["euler_cfgpp", "dpm2", "dpm2_single_call", "dpm2_churn", "dpm2_cfgpp"])
def test_other_samplers_match_cpu_oracle(which):
    """sample_euler_ancestral_cfgpp / sample_dpm2 / sample_dpm2_cfgpp (k_diffusion_euler.py:51-106,
    k_diffusion_dpm2.py:8-111) on the HIP DiT vs the line-by-line CPU restatement on the oracle DiT, same weights,
    injected noises.  fp32 mode, 1e-3 relative."""
    from oracle import sampling as OS
    from oracle.dit import DiTOracle
    from oracle.scheduler import EulerDiscreteScheduler
    from uwudiff_amd import sampling as S
    from uwudiff_amd.dit import DiT, DiTConfig

    torch.manual_seed(3)
    cfgm = dict(depth=2, hidden=128, heads=2, patch=2, sample_size=16, in_channels=4, out_channels=4, cond_dim=32)
    ora = DiTOracle(**cfgm)
    with torch.no_grad():
        for p in ora.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    model = DiT(DiTConfig(compute_dtype="fp32", **cfgm)).cuda()
    model.load_state_dict(ora.state_dict())
    abar = EulerDiscreteScheduler.sdxl().alphas_cumprod
    den = S.DiscreteEpsDDPMDenoiser(model, abar)
    B, steps = 2, 5
    sigmas = den.get_sigmas(steps)
    x0 = torch.randn(B, 4, 16, 16) * float(sigmas[0])
    noises = [torch.randn(B, 4, 16, 16) for _ in range(steps)]
    pc, pu = torch.randn(B, 32), torch.zeros(B, 32)
    ck, uk = {"added_cond_kwargs": {"text_embeds": pc}}, {"added_cond_kwargs": {"text_embeds": pu}}
    ckd, ukd = {"added_cond_kwargs": {"text_embeds": pc.cuda()}}, {"added_cond_kwargs": {"text_embeds": pu.cuda()}}
    ns = lambda i: noises[i].cuda()  # noqa: E731
    g = 3.0
    if which == "euler_cfgpp":
        ref = OS.sample_euler_ancestral_cfgpp(ora, x0, sigmas, den.log_sigmas, ck, uk, g, noises)
        out = S.sample_euler_ancestral_cfgpp(den, x0.cuda(), sigmas, ckd, ukd, cfg=g, noise_sampler=ns)
    elif which.startswith("dpm2") and which != "dpm2_cfgpp":
        kw = dict(single_call=which == "dpm2_single_call", s_churn=2.0 if which == "dpm2_churn" else 0.0)
        ref = OS.sample_dpm2(ora, x0, sigmas, den.log_sigmas, ck, uk, g, noises, **kw)
        out = S.sample_dpm2(den, x0.cuda(), sigmas, ckd, ukd, cfg=g, noise_sampler=ns, **kw)
    else:
        ref = OS.sample_dpm2_cfgpp(ora, x0, sigmas, den.log_sigmas, ck, uk, g, noises)
        out = S.sample_dpm2_cfgpp(den, x0.cuda(), sigmas, ckd, ukd, cfg=g, noise_sampler=ns)
    err = ((out.cpu() - ref).norm() / ref.norm()).item()
    assert err < 1e-3, (which, err)
