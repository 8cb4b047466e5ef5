"""reference src/duwu/loader.py: ``load_any`` / ``load_all`` / ``_load_config_`` post-processing."""
from dataclasses import dataclass
from typing import Any

import torch
import torch.nn as nn

from duwu.utils import instantiate_any

_PRECISIONS = {"torch.float32": torch.float32, "torch.float16": torch.float16, "torch.bfloat16": torch.bfloat16,
               "torch.float": torch.float32, "torch.half": torch.float16}


@dataclass
class ModelLoadingConfig:
    ckpt_path: str | None = None
    state_dict_key: str | None = None
    state_dict_prefix: str | None = None
    precision: str | None = None
    device: str | None = None
    to_compile: bool = False
    to_freeze: bool = False


def extract_state_dict(state_dict: dict[str, Any], key: str | None, prefix: str | None):
    if key is not None:
        state_dict = state_dict[key]
    if prefix is None:
        return state_dict
    return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}


def prepare_model(model: nn.Module, cfg: ModelLoadingConfig):
    if cfg.ckpt_path is not None:
        sd = torch.load(cfg.ckpt_path, map_location="cpu", weights_only=True)
        model.load_state_dict(extract_state_dict(sd, cfg.state_dict_key, cfg.state_dict_prefix))
    if cfg.precision is not None:
        if cfg.precision not in _PRECISIONS:  # the reference eval()s this string (loader.py:48); we whitelist
            raise ValueError(f"unsupported precision {cfg.precision!r}")
        if not getattr(model, "_uwu_keep_fp32_master", False):
            model = model.to(_PRECISIONS[cfg.precision])
    if cfg.device is not None:
        model = model.to(cfg.device)
    # to_compile: torch.compile is a tracing compiler; this build has none (explicit HIP kernels) -> ignored
    if cfg.to_freeze:
        model.requires_grad_(False).eval()
    return model


def load_any(obj):
    load_config = None
    if isinstance(obj, dict) and "_load_config_" in obj:
        obj = dict(obj)
        load_config = ModelLoadingConfig(**obj.pop("_load_config_"))
    obj = instantiate_any(obj)
    if load_config is not None:
        obj = prepare_model(obj, load_config)
    return obj


def load_all(conf, trainer=None, data_module=None):
    trainer = trainer or instantiate_any(conf.pop("trainer"))
    data_module = data_module or instantiate_any(conf.pop("data"))
    data_module.set_tokenizers(trainer.te.tokenizers if trainer.te is not None else [])
    return data_module, trainer
