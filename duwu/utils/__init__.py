"""reference src/duwu/utils/__init__.py: config -> object helpers and small utilities (same names)."""
import logging
import sys
from inspect import isfunction

import torch.nn as nn

from uwudiff_amd.config import get_obj_from_str, instantiate, instantiate_any, instantiate_class  # noqa: F401


def exists(val):
    return val is not None


def default(val, d):
    if val is not None:
        return val
    return d() if isfunction(d) else d


def zero_module(module: nn.Module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def count_params(model, verbose=False):
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total


def remove_none(list_x):
    return [i for i in list_x if i is not None]


def get_duwu_logger() -> logging.Logger:
    return logging.getLogger("duwu")


def setup_duwu_logger(level: int = logging.DEBUG) -> logging.Logger:
    logger = get_duwu_logger()
    logger.setLevel(level)
    handler = logging.StreamHandler(sys.stdout)
    handler.setLevel(level)
    handler.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    logger.addHandler(handler)
    return logger
