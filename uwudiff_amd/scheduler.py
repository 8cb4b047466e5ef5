"""Host-side noise schedule used by the objective: the constructor-time state of
``diffusers.EulerDiscreteScheduler`` that the reference reads (duck-typed):

    .alphas_cumprod  .timesteps  .sigmas  .config.{prediction_type,num_train_timesteps}  .get_velocity

``diffusers`` is absent offline, so hub names used by the reference configs
(``stabilityai/stable-diffusion-xl-base-1.0`` + ``subfolder: scheduler``) resolve to the built-in SDXL
``scheduler_config.json`` values below -- nothing is ever fetched.
"""
from types import SimpleNamespace

import torch

_SDXL = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    prediction_type="epsilon", steps_offset=1, timestep_spacing="leading", interpolation_type="linear",
    use_karras_sigmas=False,
)
_SD15 = dict(_SDXL)  # same beta schedule family; SD1.x uses identical betas
KNOWN_SCHEDULERS = {
    "stabilityai/stable-diffusion-xl-base-1.0": _SDXL,
    "runwayml/stable-diffusion-v1-5": _SD15,
    "sdxl": _SDXL,
}


class EulerDiscreteScheduler:
    """Training-time view of the Euler discrete scheduler (tables only; sampling lives elsewhere)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, prediction_type="epsilon", **extra):
        n = int(num_train_timesteps)
        if trained_betas is not None:
            betas = torch.as_tensor(trained_betas, dtype=torch.float32)
            n = betas.numel()
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        sig = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.sigmas = torch.cat([sig.flip(0), torch.zeros(1)])
        self.timesteps = torch.arange(n - 1, -1, -1, dtype=torch.float32)
        self.config = SimpleNamespace(num_train_timesteps=n, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type, **extra)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, **overrides):
        name = pretrained_model_name_or_path
        if isinstance(name, dict):
            cfg = dict(name)
        elif name in KNOWN_SCHEDULERS:
            cfg = dict(KNOWN_SCHEDULERS[name])
        else:
            raise ValueError(
                f"scheduler {name!r} is not in the offline registry {sorted(KNOWN_SCHEDULERS)}; "
                "pass a config dict instead (no hub access)")
        cfg.update(overrides)
        return cls(**cfg)

    from_config = from_pretrained

    def get_velocity(self, sample, noise, timesteps):
        abar = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)[timesteps.long()]
        sa, sb = abar ** 0.5, (1 - abar) ** 0.5
        while sa.dim() < sample.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * noise - sb * sample
