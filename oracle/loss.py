"""Oracle restatement of the reference objective.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain PyTorch fp32 on CPU, explicit
formulas, no autograd needed for the gradient (``dloss_dout`` is analytic and is itself
cross-checked against autograd in tests/test_oracle_loss.py).  Every function cites the
reference lines it follows (paths relative to /root/reference/src/duwu/loss/).

RNG is *injected*: the caller passes ``noise`` and ``timesteps`` (or RF ``time``), because
the CPU and HIP generators differ (SURVEY.md section 8c "RNG note").
"""
from typing import NamedTuple, Optional

import torch

PRED_TYPES = ("epsilon", "v_prediction", "sample", "rectified_flow")


class LossOracleOut(NamedTuple):
    loss: torch.Tensor          # scalar
    losses: torch.Tensor        # [B] (after weighting)
    timesteps: torch.Tensor     # [B]
    sigmas: torch.Tensor        # [B]
    noisy_latent: torch.Tensor  # [B,...]
    pred: torch.Tensor
    target: torch.Tensor
    dloss_dout: torch.Tensor    # d loss / d model_output, [B,...]


def _bcast(v, like):
    while v.dim() < like.dim():
        v = v.unsqueeze(-1)
    return v


def all_snr(scheduler):
    """diffusion.py:42-51: all_snr = (sqrt(abar)/sqrt(1-abar))**2."""
    abar = scheduler.alphas_cumprod
    return (torch.sqrt(abar) / torch.sqrt(1.0 - abar)) ** 2


def sigmas_for_timesteps(scheduler, timesteps):
    """diffusion.py:53-62: index of t in the descending ``scheduler.timesteps`` -> sigmas[idx]."""
    sched_t = scheduler.timesteps
    idx = [(sched_t == t).nonzero().item() for t in timesteps]
    return scheduler.sigmas[idx].flatten()


def q_sample(x, noise, sigmas):
    """diffusion.py:77-82: noisy = (x + noise*sigma) * 1/sqrt(sigma^2+1)."""
    s = _bcast(sigmas, x)
    scales = 1 / (s**2 + 1) ** 0.5
    return (x + noise * s) * scales


def get_target(target_type, scheduler, x0, noise, timesteps):
    """diffusion.py:84-98."""
    if target_type == "epsilon":
        return noise
    if target_type == "v_prediction":
        return scheduler.get_velocity(x0, noise, timesteps)
    if target_type == "sample":
        return x0
    if target_type == "rectified_flow":
        return noise - x0
    raise ValueError(f"Unsupported target type {target_type}")


def x0_eps_from_pred(prediction_type, xt, model_output, sigmas):
    """diffusion.py:100-125 (note: ``xt`` is the *scaled* noisy latent)."""
    s = _bcast(sigmas, xt)
    scales = 1 / (s**2 + 1) ** 0.5
    if prediction_type == "sample":
        x0 = model_output
        eps = (xt / scales - x0) / s
    elif prediction_type == "epsilon":
        eps = model_output
        x0 = xt / scales - s * eps
    elif prediction_type == "v_prediction":
        x0 = scales * (xt - s * model_output)
        eps = (xt / scales - x0) / s
    elif prediction_type == "rectified_flow":
        x0 = (xt / scales - s * model_output) / (1 + s)
        eps = (xt / scales + model_output) / (1 + s)
    else:
        raise ValueError(f"Unsupported prediction type {prediction_type}")
    return x0, eps


def snr_weight(scheduler, prediction_type, timesteps, gamma):
    """diffusion.py:141-153 (min-SNR-gamma)."""
    snr = all_snr(scheduler)[timesteps.long()]
    m = torch.minimum(snr, torch.full_like(snr, gamma))
    if prediction_type == "v_prediction":
        return (m / (snr + 1)).float()
    return (m / snr).float()


def debias_weight(scheduler, timesteps):
    """diffusion.py:155-167."""
    snr = all_snr(scheduler)[timesteps.long()]
    snr = torch.minimum(snr, torch.ones_like(snr) * 1000)
    return 1 / torch.sqrt(snr)


def _pred_coeffs(prediction_type, target_type, scheduler, sigmas, timesteps, force_convert=False):
    """d pred / d model_output as a per-sample scalar (every conversion is affine in out)."""
    s = sigmas
    scales = 1 / (s**2 + 1) ** 0.5
    if prediction_type == target_type and not force_convert:
        return torch.ones_like(s)
    # d x0 / d out, d eps / d out
    if prediction_type == "sample":
        dx0, deps = torch.ones_like(s), -1 / s
    elif prediction_type == "epsilon":
        dx0, deps = -s, torch.ones_like(s)
    elif prediction_type == "v_prediction":
        dx0 = -scales * s
        deps = -dx0 / s
    else:  # rectified_flow
        dx0, deps = -s / (1 + s), 1 / (1 + s)
    if target_type == "epsilon":
        return deps
    if target_type == "sample":
        return dx0
    if target_type == "rectified_flow":
        return deps - dx0
    if target_type == "v_prediction":
        abar = scheduler.alphas_cumprod[timesteps.long()]
        return abar**0.5 * deps - (1 - abar) ** 0.5 * dx0
    raise ValueError(target_type)


def diffusion_loss(
    scheduler,
    x,
    noise,
    timesteps,
    model_fn,
    *,
    prediction_type: Optional[str] = None,
    target_type: Optional[str] = None,
    use_snr_weight=False,
    min_snr_gamma=5.0,
    use_debiased_estimation=False,
) -> LossOracleOut:
    """``DiffusionLoss.forward`` diffusion.py:169-193 with injected (noise, timesteps).

    ``model_fn(noisy, timesteps) -> model_output``.  NB the reference passes the *clean*
    ``x`` as ``xt`` to ``get_prediction_for_training`` (diffusion.py:177); reproduced.
    """
    prediction_type = prediction_type or scheduler.config.prediction_type
    target_type = target_type or scheduler.config.prediction_type
    sigmas = sigmas_for_timesteps(scheduler, timesteps).to(x)
    noisy = q_sample(x, noise, sigmas)
    out = model_fn(noisy, timesteps)
    if prediction_type == target_type:
        pred = out
    else:
        x0h, epsh = x0_eps_from_pred(prediction_type, x, out, sigmas)
        pred = get_target(target_type, scheduler, x0h, epsh, timesteps)
    target = get_target(target_type, scheduler, x, noise, timesteps)
    losses = ((pred - target) ** 2).flatten(1).mean(1)
    w = torch.ones_like(losses)
    if use_snr_weight:
        assert prediction_type == target_type and prediction_type in ("epsilon", "v_prediction")
        w = w * snr_weight(scheduler, prediction_type, timesteps, min_snr_gamma)
    if use_debiased_estimation:
        assert prediction_type == target_type == "epsilon"
        w = w * debias_weight(scheduler, timesteps)
    losses = losses * w
    B = x.shape[0]
    n = x[0].numel()
    c = _pred_coeffs(prediction_type, target_type, scheduler, sigmas, timesteps)
    g = _bcast(2.0 * w * c / (n * B), x) * (pred - target)
    return LossOracleOut(losses.mean(), losses, timesteps, sigmas, noisy, pred, target, g)


def sigma_to_timestep(scheduler, sigmas):
    """rectified_flow.py:98-129 (log-sigma piecewise-linear inverse)."""
    log_s = torch.log(sigmas.clamp(min=1e-10))
    tbl = torch.log(scheduler.sigmas[:-1]).flip(0).to(log_s)
    dists = log_s - tbl[:, None]
    low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=tbl.shape[0] - 2)
    high_idx = low_idx + 1
    low, high = tbl[low_idx], tbl[high_idx]
    w = torch.clamp((low - log_s) / (low - high), 0, 1)
    t = (1 - w) * low_idx + w * high_idx
    return t.view(sigmas.shape)


def rf_time_to_sigma(scheduler, u01):
    """rectified_flow.py:29-38: time = u * sigma_max/(1+sigma_max); sigma = time/(1-time)."""
    smax = scheduler.sigmas[0]
    max_time = smax / (1 + smax)
    time = u01 * max_time
    return time / (1 - time)


def rectified_flow_loss(
    scheduler,
    x,
    noise,
    sigmas,
    model_fn,
    *,
    prediction_type: Optional[str] = None,
    rescale_image=False,
    rescale_noise=False,
    timesteps=None,
) -> LossOracleOut:
    """``RectifiedFlowLoss.forward`` rectified_flow.py:63-96 with injected (noise, sigmas).  ``timesteps`` given =
    ``time_sampling_type="uniform_timestep"`` (rectified_flow.py:32-33: integer timesteps, table sigmas)."""
    prediction_type = prediction_type or scheduler.config.prediction_type
    if rescale_image:  # rectified_flow.py:56-58
        x = x / x.std([1, 2, 3], keepdim=True) * 0.937
    if rescale_noise:  # :59-60
        noise = noise / noise.std([1, 2, 3], keepdim=True)
    if timesteps is None:
        timesteps = sigma_to_timestep(scheduler, sigmas)
    noisy = q_sample(x, noise, sigmas)  # :71
    out = model_fn(noisy, timesteps)
    target = noise - x  # :79
    x0h, epsh = x0_eps_from_pred(prediction_type, noisy, out, sigmas)  # :80-82
    pred = epsh - x0h
    losses = ((pred - target) ** 2).flatten(1).mean(1)
    B = x.shape[0]
    n = x[0].numel()
    # the RF forward always converts through (x0_hat, eps_hat) (rectified_flow.py:80-83)
    c = _pred_coeffs(prediction_type, "rectified_flow", scheduler, sigmas, None, force_convert=True)
    g = _bcast(2.0 * c / (n * B), x) * (pred - target)
    return LossOracleOut(losses.mean(), losses, timesteps, sigmas, noisy, pred, target, g)
