"""Import surface of the reference's ``duwu.utils`` (reference src/duwu/utils/__init__.py): the config -> object helpers
(``instantiate_any`` & co. live in uwudiff_amd/config.py), the package logger and the small list / module helpers that reference
code and user configs import from here (``text_encoders.py:8`` imports ``remove_none``, ``sampling.py:8``
``truncate_or_pad_to_length``, ``data/text_image_local.py:9`` ``get_images_recursively``).  Semantics follow the reference
function by function (cited below); the bodies are written for this package."""
import itertools
import logging
import os
import random
import sys
from inspect import isfunction
from pathlib import Path

import torch

from uwudiff_amd.config import get_obj_from_str, instantiate, instantiate_any, instantiate_class  # noqa: F401

_IMAGE_SUFFIXES = (".png", ".jpg", ".jpeg", ".webp", ".gif")


def exists(val):
    """reference utils/__init__.py:52-53"""
    return val is not None


def uniq(arr):
    """First occurrences, in order, as a dict-keys view (reference :56-57)."""
    return dict.fromkeys(arr).keys()


def default(val, d):
    """``val`` unless it is None; a plain function ``d`` is called for the fallback (reference :60-63)."""
    if val is not None:
        return val
    return d() if isfunction(d) else d


def zero_module(module):
    """Zero every parameter in place and hand the module back (reference :66-72)."""
    with torch.no_grad():
        for p in module.parameters():
            p.zero_()
    return module


def random_choice(x, num):
    """``num`` rows of ``x`` in a shuffled order, stacked (reference :75-83; Python's ``random`` as there)."""
    rows = list(x)
    random.shuffle(rows)
    return torch.stack(rows[:num])


def count_params(model, verbose=False):
    """reference :86-90"""
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total


def remove_none(list_x):
    """reference :93-94"""
    return [v for v in list_x if v is not None]


def balance_sharding_index(total, shards):
    """(start, length) of ``shards`` consecutive pieces of ``total`` items whose lengths differ by at most one, the SHORTER
    pieces first (reference :97-104: each piece takes ``remaining // pieces_left``)."""
    start = 0
    for left in range(shards, 0, -1):
        size = (total - start) // left
        yield start, size
        start += size


def balance_sharding(datas, shards):
    """reference :107-110"""
    for start, size in balance_sharding_index(len(datas), shards):
        yield datas[start:start + size]


def balance_sharding_max_size(datas, max_size):
    """The fewest balanced pieces of at most ``max_size`` items (reference :113-116)."""
    return balance_sharding(datas, -(-len(datas) // max_size))


def repeat_last(list_x, target_length):
    """reference :136-137"""
    return list_x + [list_x[-1]] * (target_length - len(list_x))


def cycling(list_x, target_length):
    """reference :140-143"""
    return list(itertools.islice(itertools.cycle(list_x), target_length))


def uniform_expansion(list_x, target_length):
    """Every element repeated in place, the repeat counts balanced as :func:`balance_sharding_index` (reference :146-152)."""
    out = []
    for item, (_, size) in zip(list_x, balance_sharding_index(target_length, len(list_x))):
        out.extend([item] * size)
    return out


_PADDERS = {"repeat_last": repeat_last, "cycling": cycling, "uniform_expansion": uniform_expansion}


def truncate_or_pad_to_length(list_x, target_length, padding_mode):
    """Cut to ``target_length`` or pad by ``padding_mode``; an unknown mode yields None as in the reference (:119-133)."""
    if len(list_x) >= target_length:
        return list_x[:target_length]
    pad = _PADDERS.get(padding_mode)
    return pad(list_x, target_length) if pad else None


def get_duwu_logger() -> logging.Logger:
    """reference :155-164"""
    return logging.getLogger("duwu")


def setup_duwu_logger(level: int = logging.DEBUG):
    """Stream handler on stdout for the package logger -- on rank 0 only, like the reference's ``rank_zero_only`` (:167-187)."""
    if int(os.environ.get("RANK", os.environ.get("LOCAL_RANK", "0"))) != 0:
        return None
    logger = get_duwu_logger()
    logger.setLevel(level)
    handler = logging.StreamHandler(sys.stdout)
    handler.setLevel(level)
    handler.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    logger.addHandler(handler)
    return logger


def get_images_recursively(folder_path: str) -> list:
    """Every png / jpg / jpeg / webp / gif below ``folder_path`` (any letter case), grouped by type in that order;
    ``ValueError`` for a missing folder (reference :190-225)."""
    if not os.path.exists(folder_path):
        raise ValueError(f"The path {folder_path} does not exist.")
    found = [p for p in Path(folder_path).rglob("*") if p.suffix.lower() in _IMAGE_SUFFIXES]
    return [str(p) for suffix in _IMAGE_SUFFIXES for p in found if p.suffix.lower() == suffix]
