"""Oracle restatement of ``diffusers.EulerDiscreteScheduler`` constructor-time state.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference never calls
``set_timesteps`` on the training scheduler; it only reads the attributes listed in
SURVEY.md section 8(b):

  * ``.alphas_cumprod [N]``                (reference src/duwu/loss/diffusion.py:45)
  * ``.timesteps [N]`` descending fp32     (diffusion.py:57)
  * ``.sigmas [N+1]`` descending + 0       (diffusion.py:61, rectified_flow.py:29,108)
  * ``.config.prediction_type`` / ``.config.num_train_timesteps`` (diffusion.py:37-40,67)
  * ``.get_velocity(x0, noise, t)``        (diffusion.py:90)

Third-party dependency: ``diffusers`` (unpinned in reference pyproject.toml:23, cited as
v0.30.2 at rectified_flow.py:101).  Published algorithm restated here:

    betas  = linspace(sqrt(beta_start), sqrt(beta_end), N, fp32) ** 2   ("scaled_linear")
    abar   = cumprod(1 - betas)
    sigmas = concat(flip(sqrt((1 - abar) / abar)), [0])
    timesteps = linspace(0, N-1, N)[::-1]  (fp32)

SDXL ``scheduler_config.json`` values (not fetchable offline) are the defaults below.
Pinned by the reference's own constant sigma_max = 14.6146
(configs/sampling/demo_sampling.yaml:49).  ``get_velocity`` follows the DDPM definition
v = sqrt(abar_t) * eps - sqrt(1 - abar_t) * x0; no reference fixture
covers it, so the v-target rows are "parity unpinned" on the third-party side.
"""
from types import SimpleNamespace

import numpy as np
import torch

SDXL_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000,
    beta_start=0.00085,
    beta_end=0.012,
    beta_schedule="scaled_linear",
    prediction_type="epsilon",
    steps_offset=1,
    timestep_spacing="leading",
    interpolation_type="linear",
    use_karras_sigmas=False,
)


class EulerDiscreteScheduler:
    def __init__(
        self,
        num_train_timesteps=1000,
        beta_start=0.0001,
        beta_end=0.02,
        beta_schedule="linear",
        trained_betas=None,
        prediction_type="epsilon",
        **extra,
    ):
        if trained_betas is not None:
            betas = torch.as_tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = (
                torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32) ** 2
            )
        else:
            raise NotImplementedError(beta_schedule)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).flip(0)
        timesteps = np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy()
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.float32)
        self.sigmas = torch.cat([sigmas, torch.zeros(1, dtype=sigmas.dtype)])
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps,
            beta_start=beta_start,
            beta_end=beta_end,
            beta_schedule=beta_schedule,
            prediction_type=prediction_type,
            **extra,
        )

    @classmethod
    def sdxl(cls, **overrides):
        cfg = dict(SDXL_SCHEDULER_CONFIG)
        cfg.update(overrides)
        return cls(**cfg)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, **kw):
        # offline: every hub name the reference configs use is the SDXL scheduler
        return cls.sdxl(**kw)

    def get_velocity(self, sample, noise, timesteps):
        abar = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)
        t = timesteps.to(sample.device).long()
        sa = abar[t] ** 0.5
        sb = (1 - abar[t]) ** 0.5
        while sa.dim() < sample.dim():
            sa = sa.unsqueeze(-1)
            sb = sb.unsqueeze(-1)
        return sa * noise - sb * sample


def laplace_logsnr(t, mu=0.0, b=1.0, eps=float(np.finfo(np.float64).eps)):
    """test_scripts/test_diffusion_scheduler.ipynb cell 1 ``t_to_logsnr_laplace`` (numpy, f64 -> f32)."""
    t = np.float64(t)
    logsnr = mu - b * np.sign(0.5 - t) * np.log(1 - 2 * np.abs(t - 0.5) + eps)
    return np.float32(logsnr)


def cosine_logsnr(t, mu=0.0, s=1.0, eps=float(np.finfo(np.float32).eps)):
    """Notebook cell 1 ``t_to_logsnr_cosine``."""
    t = np.float64(t)
    logsnr = mu + 2 / s * np.log(1 / (np.tan(np.pi * (t + eps * np.sign(0.5 - t)) / 2)))
    return np.float32(logsnr)


def logsnr_to_sigmas(logsnr):
    """Notebook cells 1-2: snr = exp(logsnr); abar = snr/(1+snr); sigma = sqrt((1-abar)/abar)."""
    snr = np.exp(logsnr)
    abar = snr / (1 + snr)
    return ((1 - abar) / abar) ** 0.5, abar


def abar_to_betas(abar):
    """Notebook cell 1 ``alpha_cumprod_to_all``: alphas[0]=abar[0], alphas[i]=abar[i]/abar[i-1]."""
    alphas = abar.copy()
    alphas[1:] = abar[1:] / abar[:-1]
    return 1 - alphas
