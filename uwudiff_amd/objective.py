"""Diffusion / rectified-flow objectives on the HIP kernels of csrc/objective.hip.

Mirrors the reference operator interface (same names, arguments and error behaviour):
  * ``DiffusionLoss``      -- reference src/duwu/loss/diffusion.py:18-193
  * ``RectifiedFlowLoss``  -- reference src/duwu/loss/rectified_flow.py:9-129
  * aux tuple              -- diffusion.py:9-15
``forward(x, unet, **unet_kwargs) -> (scalar_loss, DiffusionLossAuxOutput)``.

What changed underneath: the B host syncs per step of ``get_sigmas_for_timesteps`` (diffusion.py:58) and the
Python ``torch.stack`` loops (:146,:159) are one ``uwu_schedule_gather`` launch; q-sample is one fused kernel;
prediction conversion + target + per-sample MSE + SNR weights + d loss/d model_output are one fused kernel.
"""
from typing import Any, NamedTuple, Optional

import torch
import torch.nn as nn

from . import lib as L
from .scheduler import EulerDiscreteScheduler


class _PendingNoise:
    """noise that the q-sample kernel will draw: the output tensor and the Philox counters reserved for it"""

    def __init__(self, out, seed, offset):
        self.out, self.seed, self.offset = out, seed, offset


class DiffusionLossAuxOutput(NamedTuple):
    losses: torch.Tensor
    timesteps: torch.Tensor
    pred: torch.Tensor
    target: torch.Tensor
    noisy_latent: torch.Tensor


class _FusedLoss(torch.autograd.Function):
    """loss = mean_b w_b * mean((pred-target)^2); the gradient wrt model_output is produced in the forward pass."""

    @staticmethod
    def forward(ctx, model_output, x, noise, xt, coef, pred_type, target_type, force_convert):
        B = x.shape[0]
        n = x[0].numel()
        mo = model_output.contiguous()
        if mo.dtype not in (torch.float32, torch.bfloat16):
            mo = mo.float()
        losses = torch.empty(B, device=x.device, dtype=torch.float32)
        loss = torch.empty((), device=x.device, dtype=torch.float32)
        grad = torch.empty_like(mo)
        pred = torch.empty_like(x)
        target = torch.empty_like(x)
        L.call("uwu_loss_fwd_bwd", L.ptr(x), L.ptr(noise), L.ptr(xt), L.ptr(mo), L.dt(mo), L.ptr(coef),
               pred_type, target_type, int(force_convert), B, n, L.ptr(losses), L.ptr(loss), L.ptr(grad),
               L.ptr(pred), L.ptr(target), L.stream())
        ctx.save_for_backward(grad)
        ctx.out_dtype = model_output.dtype
        ctx.mark_non_differentiable(losses, pred, target)
        ctx.set_materialize_grads(False)  # (autograd otherwise fills a zero gradient for each of the three outputs nobody differentiates)
        return loss, losses, pred, target

    @staticmethod
    def backward(ctx, g_loss, *_):
        (saved,) = ctx.saved_tensors
        if g_loss is None:  # (materialize_grads is off: nobody asked for d loss)
            return (None,) * 8
        g = g_loss.to(device=saved.device, dtype=torch.float32).contiguous()
        # scaled INTO a new tensor: a second backward through the same graph (retain_graph / accumulation helpers) must see the
        # unscaled gradient again, and the returned tensor must not alias the saved one
        grad = torch.empty_like(saved)
        L.call("uwu_scale_into", L.ptr(saved), L.ptr(grad), L.dt(grad), grad.numel(), L.ptr(g), L.stream())
        if grad.dtype != ctx.out_dtype:
            grad = grad.to(ctx.out_dtype)
        return grad, None, None, None, None, None, None, None


class DiffusionLoss(nn.Module):
    def __init__(
        self,
        scheduler: EulerDiscreteScheduler,
        use_snr_weight: bool = False,
        min_snr_gamma: float = 5.0,
        use_debiased_estimation: bool = False,
        prediction_type: Optional[str] = None,
        target_type: Optional[str] = None,
        loss: Optional[nn.Module] = None,
    ):
        super().__init__()
        if loss is not None and not (isinstance(loss, nn.MSELoss) and loss.reduction == "none"):
            raise NotImplementedError("only nn.MSELoss(reduction='none') is implemented on the HIP path")
        self.scheduler = scheduler
        self.prepare_scheduler_for_custom_training()
        self.use_snr_weight = use_snr_weight
        self.min_snr_gamma = min_snr_gamma
        self.use_debiased_estimation = use_debiased_estimation
        self.prediction_type = prediction_type or self.scheduler.config.prediction_type
        self.target_type = target_type or self.scheduler.config.prediction_type
        self.n_diffusion_time_steps = self.scheduler.config.num_train_timesteps
        self._tables = {}
        self._inject = None
        self._latent_norm = None  # (vae_mean, vae_std): trainer.py:241-244, folded into the q-sample kernel

    # diffusion.py:42-51
    def prepare_scheduler_for_custom_training(self):
        if hasattr(self.scheduler, "all_snr"):
            return
        abar = self.scheduler.alphas_cumprod
        self.scheduler.all_snr = (torch.sqrt(abar) / torch.sqrt(1.0 - abar)) ** 2

    def set_latent_normalisation(self, mean=0.0, std=None):
        """``x <- (x - vae_mean) / vae_std`` (reference trainer.py:241-244, applied there between the VAE and the loss) as a
        fused pre-step of the forward process: one kernel writes the normalised latent and the noisy latent."""
        self._latent_norm = None if std is None else (float(mean or 0.0), float(std))

    def inject(self, noise=None, timesteps=None, u01=None):
        """One-shot RNG injection for parity runs (CPU and HIP generators differ)."""
        self._inject = dict(noise=noise, timesteps=timesteps, u01=u01)

    def _take_injected(self, key):
        if self._inject is None:
            return None
        return self._inject.get(key)

    def _dev_tables(self, device):
        t = self._tables.get(device)
        if t is None:
            s = self.scheduler
            t = dict(
                sigmas=s.sigmas.to(device=device, dtype=torch.float32).contiguous(),
                all_snr=s.all_snr.to(device=device, dtype=torch.float32).contiguous(),
                abar=s.alphas_cumprod.to(device=device, dtype=torch.float32).contiguous(),
                log_sigmas_asc=torch.log(s.sigmas[:-1]).flip(0).to(device=device, dtype=torch.float32).contiguous(),
            )
            self._tables[device] = t
        return t

    @staticmethod
    def _type_id(name, what):
        if name not in L.PT:
            raise ValueError(f"Unsupported {what} type {name}")
        return L.PT[name]

    def _snr_mode(self):
        if not self.use_snr_weight:
            return 0
        assert self.prediction_type == self.target_type
        assert self.prediction_type in ["epsilon", "v_prediction"]
        return 2 if self.prediction_type == "v_prediction" else 1

    @staticmethod
    def _reserve(device, n_counters):
        """(seed, offset) of ``n_counters`` Philox counters taken from torch's CUDA generator of ``device`` (whose offset moves on
        by them, rounded up to its granule of 4): the in-kernel draws follow ``torch.manual_seed`` like torch's own, and the ORDER
        of reservations is the reference's order of draws (noise before timesteps, diffusion.py:75-76 / :68-70)."""
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        seed, off = gen.initial_seed() & ((1 << 64) - 1), gen.get_offset()
        gen.set_offset(off + (int(n_counters) + 3) // 4 * 4)
        return seed, off

    # diffusion.py:53-72
    def sample_timesteps_and_sigmas(self, ref_params: torch.Tensor):
        B = ref_params.size(0)
        t = self._take_injected("timesteps")
        if t is None:  # t ~ U{0 .. N-1}, drawn by the library's Philox kernel (diffusion.py:68-70)
            t = torch.empty(B, device=ref_params.device, dtype=torch.int64)
            seed, off = self._reserve(ref_params.device, (B + 3) // 4)
            L.call("uwu_draw_timesteps", L.ptr(t), int(self.scheduler.config.num_train_timesteps), B, seed, off, L.stream())
        t = t.to(device=ref_params.device, dtype=torch.int64).contiguous()
        debias = 0
        if self.use_debiased_estimation:
            assert self.prediction_type == self.target_type == "epsilon"
            debias = 1
        tb = self._dev_tables(ref_params.device)
        coef = torch.empty(B, 4, device=ref_params.device, dtype=torch.float32)
        L.call("uwu_schedule_gather", L.ptr(t), L.ptr(tb["sigmas"]), L.ptr(tb["all_snr"]), L.ptr(tb["abar"]),
               self.n_diffusion_time_steps, B, self._snr_mode(), float(self.min_snr_gamma), debias, L.ptr(coef),
               L.stream())
        return t, coef

    def get_sigmas_for_timesteps(self, timesteps):
        return self.scheduler.sigmas.to(timesteps.device)[self.n_diffusion_time_steps - 1 - timesteps.long()]

    @staticmethod
    def _qsample(x, noise, coef):
        noisy = torch.empty_like(x)
        L.call("uwu_qsample", L.ptr(x), L.ptr(noise), L.ptr(coef), x.shape[0], x[0].numel(), L.ptr(noisy), None,
               L.stream())
        return noisy

    def _qsample_normed(self, x, noise, coef):
        """(clean latent the loss sees, noisy latent); with a latent normalisation set both come out of one kernel."""
        if isinstance(noise, _PendingNoise):  # not injected: drawn inside the q-sample kernel (one pass writes noise + noisy)
            use_norm = self._latent_norm is not None
            mean, std = self._latent_norm if use_norm else (0.0, 1.0)
            xn = torch.empty_like(x) if use_norm else None
            noisy = torch.empty_like(x)
            L.call("uwu_qsample_draw", L.ptr(x), L.ptr(coef), x.shape[0], x[0].numel(), int(use_norm), float(mean), float(std),
                   L.ptr(xn), L.ptr(noise.out), L.ptr(noisy), noise.seed, noise.offset, L.stream())
            return (xn if use_norm else x), noisy
        if self._latent_norm is None:
            return x, self._qsample(x, noise, coef)
        mean, std = self._latent_norm
        xn, noisy = torch.empty_like(x), torch.empty_like(x)
        L.call("uwu_qsample_norm", L.ptr(x), L.ptr(noise), L.ptr(coef), x.shape[0], x[0].numel(), mean, std, L.ptr(xn),
               L.ptr(noisy), L.stream())
        return xn, noisy

    def _noise_like(self, x, now=False):
        """Injected noise, or -- diffusion.py:75 ``randn_like(x)`` -- N(0, 1) from the library's Philox kernels: the counters are
        reserved HERE (before the timesteps', the reference's order); the values are written by the q-sample kernel itself
        unless ``now`` asks for the tensor at once (rescale_noise needs its statistics first)."""
        n = self._take_injected("noise")
        if n is not None:
            return n.to(device=x.device, dtype=torch.float32).contiguous()
        assert x.numel() % 4 == 0
        seed, off = self._reserve(x.device, x.numel() // 4)
        out = torch.empty_like(x)
        if now:
            L.call("uwu_philox_normal", L.ptr(out), out.numel(), seed, off, L.stream())
            return out
        return _PendingNoise(out, seed, off)

    # diffusion.py:169-193
    def forward(self, x: torch.Tensor, unet: nn.Module, **unet_kwargs):
        pt = self._type_id(self.prediction_type, "prediction")
        tt = self._type_id(self.target_type, "target")
        x = x.float().contiguous()
        noise = self._noise_like(x)  # drawn before the timesteps, as diffusion.py:75-76
        timesteps, coef = self.sample_timesteps_and_sigmas(x)
        self._inject = None
        x, noisy = self._qsample_normed(x, noise, coef)
        if isinstance(noise, _PendingNoise):
            noise = noise.out
        model_output = unet(noisy, timesteps, **unet_kwargs)[0]
        # NB the reference hands the *clean* x to get_prediction_for_training as `xt` (diffusion.py:177)
        loss, losses, pred, target = _FusedLoss.apply(model_output, x, noise, x, coef, pt, tt, False)
        aux = DiffusionLossAuxOutput(losses=losses, timesteps=timesteps, pred=pred, target=target,
                                     noisy_latent=noisy)
        return loss, aux


class RectifiedFlowLoss(DiffusionLoss):
    def __init__(
        self,
        time_sampling_type: str = "uniform_time",
        time_sampling_kwargs: dict[str, Any] = {},
        rescale_image: bool = False,
        rescale_noise: bool = False,
        **kwargs,
    ):
        super().__init__(**kwargs)
        self.target_type = "rectified_flow"
        self.time_sampling_type = time_sampling_type
        self.time_sampling_kwargs = time_sampling_kwargs
        self.rescale_image = rescale_image
        self.rescale_noise = rescale_noise

    # rectified_flow.py:26-47
    def sample_timesteps_and_sigmas(self, ref_params: torch.Tensor):
        if self.time_sampling_type == "uniform_timestep":
            return super().sample_timesteps_and_sigmas(ref_params)
        if self.time_sampling_type != "uniform_time":
            raise ValueError(f"Unsupported time sampling type: {self.time_sampling_type}")
        B = ref_params.size(0)
        u = self._take_injected("u01")
        if u is None:  # rectified_flow.py:37 ``torch.rand(B)``: the library's Philox kernel
            u = torch.empty(B, device=ref_params.device, dtype=torch.float32)
            seed, off = self._reserve(ref_params.device, (B + 3) // 4)
            L.call("uwu_draw_u01", L.ptr(u), B, seed, off, L.stream())
        u = u.to(device=ref_params.device, dtype=torch.float32).contiguous()
        tb = self._dev_tables(ref_params.device)
        coef = torch.empty(B, 4, device=ref_params.device, dtype=torch.float32)
        timesteps = torch.empty(B, device=ref_params.device, dtype=torch.float32)
        L.call("uwu_rf_time_to_sigma", L.ptr(u), float(self.scheduler.sigmas[0]), L.ptr(tb["log_sigmas_asc"]),
               self.n_diffusion_time_steps, B, L.ptr(coef), L.ptr(timesteps), L.stream())
        return timesteps, coef

    # rectified_flow.py:49-61
    def get_x0_and_noises(self, x: torch.Tensor):
        if len(x.shape) == 5:
            noises = x[:, 1, ...].float().contiguous()
            x = x[:, 0, ...].float().contiguous()
        else:
            x = x.float().contiguous()
            noises = self._noise_like(x, now=self.rescale_noise)
        if self.rescale_image:
            x = (x / x.std([1, 2, 3], keepdim=True) * 0.937).contiguous()
        if self.rescale_noise:
            noises = (noises / noises.std([1, 2, 3], keepdim=True)).contiguous()
        return x, noises

    def _prenormalise(self, x):
        """The reference normalises the VAE latent BEFORE the loss sees it (trainer.py:241-244), so ``rescale_image`` takes the
        per-sample std of the normalised latent: with that flag (or a 5-D sample+noise input) the normalisation cannot ride
        in the q-sample kernel and is applied here, first.  Returns (x, done)."""
        if self._latent_norm is not None and (self.rescale_image or x.dim() == 5):
            mean, std = self._latent_norm
            if x.dim() == 5:  # [B, 2, C, H, W]: plane 0 is the latent, plane 1 the injected noise
                x = x.float().clone()
                x[:, 0] = (x[:, 0] - mean) / std
                return x, True
            return (x.float() - mean) / std, True
        return x, False

    def _forward_process(self, x):
        """(clean latent, noises, timesteps, coef, noisy latent) in the reference's order of operations"""
        x, normed = self._prenormalise(x)
        x, noises = self.get_x0_and_noises(x)
        timesteps, coef = self.sample_timesteps_and_sigmas(x)
        self._inject = None
        if isinstance(noises, _PendingNoise):  # drawn inside the q-sample kernel (a prenormalised x takes no second normalisation)
            norm, self._latent_norm = self._latent_norm, (None if normed else self._latent_norm)
            try:
                x, noisy = self._qsample_normed(x, noises, coef)
            finally:
                self._latent_norm = norm
            noises = noises.out
        elif normed:
            noisy = self._qsample(x, noises, coef)
        else:
            x, noisy = self._qsample_normed(x, noises, coef)
        return x, noises, timesteps, coef, noisy

    # rectified_flow.py:63-96
    def forward(self, x: torch.Tensor, unet: nn.Module, **unet_kwargs):
        pt = self._type_id(self.prediction_type, "prediction")
        x, noises, timesteps, coef, noisy = self._forward_process(x)
        model_output = unet(noisy, timesteps, **unet_kwargs)[0]
        loss, losses, pred, target = _FusedLoss.apply(model_output, x, noises, noisy, coef, pt, L.PT["rectified_flow"],
                                                      True)
        aux = DiffusionLossAuxOutput(losses=losses, timesteps=timesteps, pred=pred, target=target,
                                     noisy_latent=noisy)
        return loss, aux


class NNWeightedRFLossAuxOutput(NamedTuple):
    losses: torch.Tensor
    rescaled_losses: torch.Tensor
    pred_losses: torch.Tensor
    loss_pred_losses: torch.Tensor
    timesteps: torch.Tensor
    pred: torch.Tensor
    target: torch.Tensor
    noisy_latent: torch.Tensor


class NNWeightedRFLoss(RectifiedFlowLoss):
    """reference src/duwu/loss/rectified_flow.py:144-203: a second network predicts the log of the per-sample RF loss;
    ``losses = rf / exp(s_hat).clamp(1e-4) + (log rf - s_hat)^2``.

    The denoiser-side term ``rf_b / pred_loss_b`` is the fused loss kernel with ``1/pred_loss_b`` in its per-sample
    weight slot (so the gradient wrt the model output comes out of the same single pass); the ``[B]``-sized
    log/square tail and the gradient into ``loss_pred_module`` stay in autograd."""

    def __init__(self, loss_pred_module: nn.Module, **kwargs):
        super().__init__(**kwargs)
        self.loss_pred_module = loss_pred_module

    def forward(self, x: torch.Tensor, unet: nn.Module, **unet_kwargs):
        pt = self._type_id(self.prediction_type, "prediction")
        x, noises, timesteps, coef, noisy = self._forward_process(x)  # (incl. the VAE latent normalisation, as every loss)
        model_output = unet(noisy, timesteps, **unet_kwargs)[0]
        sigmas = coef[:, 0]
        log_ls_pred = self.loss_pred_module(noisy, sigmas, **unet_kwargs).flatten()  # takes sigmas (:180-183)
        pred_loss = log_ls_pred.detach().exp().clamp(min=1e-4)
        coef_w = coef.clone()
        coef_w[:, 1] = 1.0 / pred_loss
        rescaled_mean, rescaled, pred, target = _FusedLoss.apply(model_output, x, noises, noisy, coef_w,
                                                                 pt, L.PT["rectified_flow"], True)
        rf_losses = rescaled * pred_loss
        ls_pred_loss = (rf_losses.log() - log_ls_pred).square()
        loss = rescaled_mean + ls_pred_loss.mean()
        aux = NNWeightedRFLossAuxOutput(losses=rf_losses, rescaled_losses=rescaled, pred_losses=pred_loss,
                                        loss_pred_losses=ls_pred_loss, timesteps=timesteps, pred=pred, target=target,
                                        noisy_latent=noisy)
        return loss, aux
