"""reference src/duwu/utils/__init__.py: the config -> object helpers the launcher, loader and trainer import
(``instantiate_any`` & co. live in uwudiff_amd/config.py) and the package logger."""
import logging

from uwudiff_amd.config import get_obj_from_str, instantiate, instantiate_any, instantiate_class  # noqa: F401


def get_duwu_logger() -> logging.Logger:
    return logging.getLogger("duwu")
