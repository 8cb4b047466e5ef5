"""Sampling loop on the HIP kernels: k-diffusion style discrete-sigma eps denoiser,
classifier-free guidance and Euler-ancestral steps, mirroring the reference's ``duwu.sampling``
(src/duwu/sampling/k_diffusion_wrapper.py:22-108, cfg.py:54-127, k_diffusion_euler.py:8-48, get_sigmas.py:6-41).

The denoiser forward re-uses the training kernels (batch doubled for CFG); the guidance combine, the eps->denoised
conversion and the ancestral update are one fused kernel (``uwu_sampler_step``).  The sigma grid and sigma<->t mapping
are host-side table logic (a 1000-entry table), as in the reference.
"""
import math

import numpy as np
import torch

from . import lib as L


class DiscreteEpsDDPMDenoiser:
    """k_diffusion_wrapper.py:22-108 (DiscreteSchedule + DiscreteEpsDDPMDenoiser)."""

    def __init__(self, model, alphas_cumprod, quantize=False):
        self.inner_model = model
        self.sigmas = (((1 - alphas_cumprod) / alphas_cumprod) ** 0.5).float().cpu()
        self.log_sigmas = self.sigmas.log()
        self.quantize = quantize
        self.sigma_data = 1.0

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def get_sigmas(self, n=None):
        if n is None:
            return torch.cat([self.sigmas.flip(0), self.sigmas.new_zeros([1])])
        t = torch.linspace(len(self.sigmas) - 1, 0, n)
        return torch.cat([self.t_to_sigma(t), self.sigmas.new_zeros([1])])

    def sigma_to_t(self, sigma, quantize=None):
        quantize = self.quantize if quantize is None else quantize
        sigma = torch.as_tensor(sigma, dtype=torch.float32).cpu()
        log_sigma = sigma.log()
        dists = log_sigma - self.log_sigmas[:, None]
        if quantize:
            return dists.abs().argmin(dim=0).view(sigma.shape)
        low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=self.log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = self.log_sigmas[low_idx], self.log_sigmas[high_idx]
        w = ((low - log_sigma) / (low - high)).clamp(0, 1)
        return ((1 - w) * low_idx + w * high_idx).view(sigma.shape)

    def t_to_sigma(self, t):
        t = torch.as_tensor(t).float()
        low_idx, high_idx, w = t.floor().long(), t.ceil().long(), t.frac()
        return ((1 - w) * self.log_sigmas[low_idx] + w * self.log_sigmas[high_idx]).exp()

    def get_scalings(self, sigma):
        return -sigma, 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5

    def eps(self, x_scaled, sigma_cond, **kwargs):
        B = x_scaled.shape[0]
        t = self.sigma_to_t(torch.full((B,), float(sigma_cond))).to(x_scaled.device)
        return self.inner_model(x_scaled, t, **kwargs)[0]


def get_sigmas_for_rf(num_steps, max_sigma, min_sigma=0, time_disc_func=None):
    """get_sigmas.py:6-18."""
    max_time, min_time = max_sigma / (1 + max_sigma), min_sigma / (1 + min_sigma)
    f = time_disc_func or (lambda a, b, n: np.linspace(a, b, n + 1))
    time = np.flip(f(min_time, max_time, num_steps))
    return time / (1 - time)


def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """k_diffusion.sampling.get_ancestral_step (third-party, published algorithm)."""
    if not eta:
        return sigma_to, 0.0
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    return (sigma_to ** 2 - sigma_up ** 2) ** 0.5, sigma_up


def _cat_kwargs(cond, uncond):
    """Batch-doubled conditioning (cond first, uncond second), cfg.py:95-112."""
    out = {}
    for k in set(cond) | set(uncond):
        a, b = cond.get(k), uncond.get(k)
        if isinstance(a, dict):
            out[k] = {kk: torch.cat([a[kk], b[kk]]) for kk in a}
        elif torch.is_tensor(a):
            out[k] = torch.cat([a, b])
        else:
            out[k] = a
    return out


@torch.no_grad()
def sample_euler_ancestral(denoiser: DiscreteEpsDDPMDenoiser, x, sigmas, cond_kwargs, uncond_kwargs=None, cfg=1.0,
                           eta=1.0, s_noise=1.0, noise_sampler=None, callback=None):
    """k_diffusion_euler.py:8-48 with the CFG wrapper folded in.  x: fp32 [B,C,H,W] on the device, already scaled by
    sigma_max.  ``noise_sampler(i)`` returns the fresh noise of step i (default: torch.randn_like on the device)."""
    x = x.float().contiguous()
    n = x.numel()
    guided = uncond_kwargs is not None
    kw = _cat_kwargs(cond_kwargs, uncond_kwargs) if guided else cond_kwargs
    xin = torch.empty_like(x)
    for i in range(len(sigmas) - 1):
        s, s_next = float(sigmas[i]), float(sigmas[i + 1])
        c_in = 1.0 / math.sqrt(s * s + 1.0)
        L.call("uwu_scale_copy", L.ptr(x), L.ptr(xin), n, c_in, L.stream())
        if guided:
            eps = denoiser.eps(torch.cat([xin, xin]), s, **kw).float().contiguous()
            eps_c, eps_u = eps[: x.shape[0]], eps[x.shape[0]:]
        else:
            eps_c, eps_u = denoiser.eps(xin, s, **kw).float().contiguous(), None
        sd, su = get_ancestral_step(s, s_next, eta)
        noise = None
        if s_next > 0:
            noise = (noise_sampler(i) if noise_sampler is not None else torch.randn_like(x)).float().contiguous()
        out = torch.empty_like(x)
        den = torch.empty_like(x) if callback is not None else None
        L.call("uwu_sampler_step", L.ptr(x), L.ptr(eps_c), L.ptr(eps_u) if eps_u is not None else None,
               L.ptr(noise) if noise is not None else None, L.ptr(out), L.ptr(den) if den is not None else None, n,
               float(cfg), s, float(sd), float(su), float(s_noise), L.stream())
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": den})
        x = out
    return x


# ---- DPM-Solver-2 and the CFG++ variants (section 8f rank 1).  Every update of these samplers is
#   out = base + a * eps_cfg + b * eps_uncond + c * noise      (uwu_sampler_combine)
# because the wrapped model predicts eps: denoised = x - sigma * eps and k-diffusion's to_d(x, sigma, denoised) = eps.
def _eval_eps(denoiser, x, xin, sigma, kw, guided):
    """(eps_cond, eps_uncond | None) of the eps model at (x, sigma): c_in scaling + one batched call for both branches."""
    L.call("uwu_scale_copy", L.ptr(x), L.ptr(xin), x.numel(), 1.0 / math.sqrt(sigma * sigma + 1.0), L.stream())
    if guided:
        eps = denoiser.eps(torch.cat([xin, xin]), sigma, **kw).float().contiguous()
        return eps[: x.shape[0]], eps[x.shape[0]:]
    return denoiser.eps(xin, sigma, **kw).float().contiguous(), None


def _combine(base, eps_c, eps_u, noise, cfg, a, b, c=0.0):
    out = torch.empty_like(base)
    L.call("uwu_sampler_combine", L.ptr(base), L.ptr(eps_c), L.ptr(eps_u) if eps_u is not None else None,
           L.ptr(noise) if noise is not None else None, L.ptr(out), base.numel(), float(cfg), float(a), float(b),
           float(c), L.stream())
    return out


def sample_euler_ancestral_cfgpp(denoiser: DiscreteEpsDDPMDenoiser, x, sigmas, cond_kwargs, uncond_kwargs, cfg=1.0, eta=1.0,
                                 s_noise=1.0, noise_sampler=None):
    """k_diffusion_euler.py:51-106 (image_to_noise=False): x' = cfg_denoised + to_d(x, sigma, uncond_denoised) * sigma_down
    (+ noise) = x - sigma eps_cfg + sigma_down eps_uncond (+ s_noise sigma_up noise)."""
    x = x.float().contiguous()
    kw = _cat_kwargs(cond_kwargs, uncond_kwargs)
    xin = torch.empty_like(x)
    for i in range(len(sigmas) - 1):
        s, s_next = float(sigmas[i]), float(sigmas[i + 1])
        eps_c, eps_u = _eval_eps(denoiser, x, xin, s, kw, True)
        sd, su = get_ancestral_step(s, s_next, eta)
        noise = None
        if s_next > 0:
            noise = (noise_sampler(i) if noise_sampler is not None else torch.randn_like(x)).float().contiguous()
        x = _combine(x, eps_c, eps_u, noise, cfg, -s, sd, s_noise * su)
    return x


def sample_dpm2(denoiser: DiscreteEpsDDPMDenoiser, x, sigmas, cond_kwargs, uncond_kwargs=None, cfg=1.0, s_churn=0.0,
                s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, single_call=False, noise_sampler=None):
    """k_diffusion_dpm2.py:8-57.  ``single_call`` reuses the midpoint derivative of the previous step as this step's
    first derivative (one model call per step), as the reference does."""
    x = x.float().contiguous()
    guided = uncond_kwargs is not None
    kw = _cat_kwargs(cond_kwargs, uncond_kwargs) if guided else cond_kwargs
    xin = torch.empty_like(x)
    cached = None
    nsteps = len(sigmas) - 1
    for i in range(nsteps):
        s, s_next = float(sigmas[i]), float(sigmas[i + 1])
        gamma = min(s_churn / nsteps, 2 ** 0.5 - 1) if s_tmin <= s <= s_tmax else 0.0
        s_hat = s * (gamma + 1)
        if gamma > 0:
            noise = (noise_sampler(i) if noise_sampler is not None else torch.randn_like(x)).float().contiguous()
            zero = torch.zeros_like(x)
            x = _combine(x, zero, None, noise, 1.0, 0.0, 0.0, s_noise * (s_hat ** 2 - s ** 2) ** 0.5)
        if s_next == 0:  # Euler step
            eps_c, eps_u = _eval_eps(denoiser, x, xin, s_hat, kw, guided)
            x = _combine(x, eps_c, eps_u, None, cfg, s_next - s_hat, 0.0)
            continue
        if single_call and cached is not None:
            eps_c, eps_u = cached
        else:
            eps_c, eps_u = _eval_eps(denoiser, x, xin, s_hat, kw, guided)
        s_mid = math.exp(0.5 * (math.log(s_hat) + math.log(s_next)))
        x2 = _combine(x, eps_c, eps_u, None, cfg, s_mid - s_hat, 0.0)
        eps_c2, eps_u2 = _eval_eps(denoiser, x2, xin, s_mid, kw, guided)
        cached = (eps_c2, eps_u2)
        x = _combine(x, eps_c2, eps_u2, None, cfg, s_next - s_hat, 0.0)
    return x


def sample_dpm2_cfgpp(denoiser: DiscreteEpsDDPMDenoiser, x, sigmas, cond_kwargs, uncond_kwargs, cfg=1.0, s_churn=0.0,
                      s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, noise_sampler=None):
    """k_diffusion_dpm2.py:60-111 (single_call=False; the reference marks cfg++ with single_call as not working):
    x_2 = cfg_denoised + uncond_d * sigma_mid;  x' = cfg_denoised_2 + uncond_d_2 * sigma_next;  last step x' = cfg_denoised."""
    x = x.float().contiguous()
    kw = _cat_kwargs(cond_kwargs, uncond_kwargs)
    xin = torch.empty_like(x)
    nsteps = len(sigmas) - 1
    for i in range(nsteps):
        s, s_next = float(sigmas[i]), float(sigmas[i + 1])
        gamma = min(s_churn / nsteps, 2 ** 0.5 - 1) if s_tmin <= s <= s_tmax else 0.0
        s_hat = s * (gamma + 1)
        if gamma > 0:
            noise = (noise_sampler(i) if noise_sampler is not None else torch.randn_like(x)).float().contiguous()
            zero = torch.zeros_like(x)
            x = _combine(x, zero, None, noise, 1.0, 0.0, 0.0, s_noise * (s_hat ** 2 - s ** 2) ** 0.5)
        eps_c, eps_u = _eval_eps(denoiser, x, xin, s_hat, kw, True)
        if s_next == 0:
            x = _combine(x, eps_c, eps_u, None, cfg, -s_hat, 0.0)  # x = cfg_denoised
            continue
        s_mid = math.exp(0.5 * (math.log(s_hat) + math.log(s_next)))
        x2 = _combine(x, eps_c, eps_u, None, cfg, -s_hat, s_mid)
        eps_c2, eps_u2 = _eval_eps(denoiser, x2, xin, s_mid, kw, True)
        x = _combine(x2, eps_c2, eps_u2, None, cfg, -s_mid, s_next)
    return x
