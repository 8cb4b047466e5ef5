"""reference src/duwu/trainer/trainer.py: ``DMTrainer`` (constructor signature, ``training_step`` contract,
``configure_optimizers``) without Lightning.  LyCORIS adapters and Lightning checkpoint fix-ups are out of scope.
"""
from typing import Any

import torch
import torch.nn as nn
import torch.optim as optim
import torch.optim.lr_scheduler as lr_sch

from duwu.loader import load_any
from duwu.utils import instantiate_any
from uwudiff_amd.engine import GradualWarmupScheduler
from uwudiff_amd.optim import FusedAdamW


class BaseTrainer(nn.Module):
    def __init__(self, *args, name="", lr=1e-5, optimizer=optim.AdamW,
                 opt_config={"weight_decay": 0.01, "betas": (0.9, 0.999)},
                 lr_scheduler=lr_sch.CosineAnnealingLR, lr_scheduler_config={"T_max": 100_000, "eta_min": 1e-7},
                 use_warm_up=True, warm_up_period=1000, **kwargs):
        super().__init__()
        self.name = name
        self.train_params = None
        self.optimizer = instantiate_any(optimizer)
        self.opt_config = dict(opt_config)
        if "betas" in self.opt_config:
            self.opt_config["betas"] = tuple(self.opt_config["betas"])
        self.lr = lr
        self.lr_sch = instantiate_any(lr_scheduler)
        self.lr_sch_config = dict(lr_scheduler_config)
        self.use_warm_up = use_warm_up
        self.warm_up_period = warm_up_period
        self.global_step = 0
        self._trainer = None

    # trainer.py:52-74
    def configure_optimizers(self):
        assert self.train_params is not None
        params = list(self.train_params)
        opt_cls = self.optimizer
        if opt_cls is optim.AdamW and all(p.is_cuda for p in params):
            opt_cls = FusedAdamW  # same update rule, one HIP launch over the flat buffer
        optimizer = opt_cls(params, lr=self.lr, **self.opt_config)
        sched = self.lr_sch(optimizer, **self.lr_sch_config) if self.lr_sch is not None else None
        if self.use_warm_up:
            sched = GradualWarmupScheduler(optimizer, 1, self.warm_up_period, sched)
        if sched is None:
            return optimizer
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sched, "interval": "step"}}


class DMTrainer(BaseTrainer):
    def __init__(self, model_config, te_use_normed_ctx=False, vae_std=None, vae_mean=None, lycoris_config=None, *args,
                 name="", lr=1e-5, optimizer=optim.AdamW, opt_config={"weight_decay": 0.01, "betas": (0.9, 0.999)},
                 lr_scheduler=lr_sch.CosineAnnealingLR, lr_scheduler_config={"T_max": 100_000, "eta_min": 1e-7},
                 use_warm_up=True, warm_up_period=1000, loss_config=None):
        super().__init__(*args, name=name, lr=lr, optimizer=optimizer, opt_config=opt_config,
                         lr_scheduler=lr_scheduler, lr_scheduler_config=lr_scheduler_config, use_warm_up=use_warm_up,
                         warm_up_period=warm_up_period)
        if lycoris_config is not None:
            raise NotImplementedError("LyCORIS adapters are out of scope of the MI355X hot path")
        self.unet = load_any(model_config["unet"])
        self.te = load_any(model_config["te"]) if model_config.get("te") is not None else None
        # trainer.py:136,241-244: any frozen module with `.encode(x).latent_dist.sample()` (diffusers.AutoencoderKL needs
        # hub weights; `uwudiff_amd.conditioning.SyntheticVAE` is the offline stand-in).  The (x - mean) / std step that
        # follows the encoder is folded into the loss's q-sample kernel.
        self.vae = load_any(model_config["vae"]) if model_config.get("vae") is not None else None
        if self.vae is not None:
            self.vae.requires_grad_(False).eval()
        self.te_use_normed_ctx = te_use_normed_ctx
        self.vae_std, self.vae_mean = vae_std, vae_mean or 0
        self.register_buffer("ema_loss", torch.tensor(0.0))
        self.ema_decay = 0.99
        self.unet.requires_grad_(True).train()
        self.train_params = self.unet.parameters()
        if loss_config is None:  # trainer.py:171-178: default = SDXL scheduler, epsilon objective
            from duwu.loss import DiffusionLoss
            from uwudiff_amd.scheduler import EulerDiscreteScheduler

            self.loss = DiffusionLoss(EulerDiscreteScheduler.from_pretrained(
                "stabilityai/stable-diffusion-xl-base-1.0", subfolder="scheduler"))
        else:
            self.loss = instantiate_any(loss_config)
        self.n_diffusion_time_steps = self.loss.n_diffusion_time_steps
        if self.vae is not None and self.vae_std is not None:
            if not hasattr(self.loss, "set_latent_normalisation"):
                raise NotImplementedError("vae_mean / vae_std need a loss with set_latent_normalisation (duwu.loss.*)")
            self.loss.set_latent_normalisation(self.vae_mean, self.vae_std)

    # trainer.py:233-261
    def get_latent_and_conditioning(self, batch):
        x, captions, tokenizer_outputs, added_cond, cross_attn_kwargs = batch
        ctx = attn_mask = pooled = None
        with torch.no_grad():
            if self.vae is not None:  # trainer.py:241-243 (the normalisation of :244 happens inside the loss kernel)
                x = self.vae.encode(x).latent_dist.sample()
            if self.te is not None:
                embedding, normed, pooled, attn_mask = self.te(tokenizer_outputs)
                ctx = normed if self.te_use_normed_ctx else embedding
        added_cond = dict(added_cond)
        added_cond["text_embeds"] = pooled
        return x, ctx, attn_mask, added_cond, cross_attn_kwargs

    # trainer.py:263-294 (the two .item() host syncs per step are left to the driver's periodic logging)
    def training_step(self, batch, idx):
        x, ctx, attn_mask, added_cond, cross_attn_kwargs = self.get_latent_and_conditioning(batch)
        loss, aux_output = self.loss(x, self.unet, encoder_hidden_states=ctx, encoder_attention_mask=attn_mask,
                                     added_cond_kwargs=added_cond, cross_attention_kwargs=cross_attn_kwargs)
        ema_decay = min(self.global_step / (10 + self.global_step), self.ema_decay)
        self.ema_loss = ema_decay * self.ema_loss + (1 - ema_decay) * loss.detach()
        return {"loss": loss, "aux_output": aux_output}

    def validation_step(self, batch, idx):
        x, ctx, attn_mask, added_cond, cross_attn_kwargs = self.get_latent_and_conditioning(batch)
        with torch.no_grad():
            return self.loss(x, self.unet, encoder_hidden_states=ctx, encoder_attention_mask=attn_mask,
                             added_cond_kwargs=added_cond, cross_attention_kwargs=cross_attn_kwargs)
