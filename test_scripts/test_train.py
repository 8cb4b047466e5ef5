"""Launcher with the reference's CLI (reference test_scripts/test_train.py):

    python test_scripts/test_train.py --configs configs/demo_training_latent.yaml [more.yaml ...]

Multi-GPU: ``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 test_scripts/test_train.py ...``
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from duwu.loader import load_all  # noqa: E402
from duwu.utils import get_duwu_logger, instantiate_any  # noqa: E402
from uwudiff_amd.config import load_yaml, merge  # noqa: E402
from uwudiff_amd.engine import Fitter, LearningRateMonitor, seed_everything  # noqa: E402

if __name__ == "__main__":
    print(f"PyTorch version: {torch.__version__}")
    parser = argparse.ArgumentParser()
    parser.add_argument("--configs", type=str, nargs="+", default=None)
    args = parser.parse_args()
    config = merge(*[load_yaml(c) for c in args.configs])

    logger = get_duwu_logger()
    lightning_config = {"accelerator": "gpu", "precision": "16-true", "devices": 1, "fast_dev_run": True,
                        "deterministic": True, "use_distributed_sampler": False, "callbacks": [], "logger": [],
                        "plugins": []}
    if "lightning_config" in config:
        lc = dict(config["lightning_config"])
        lc["callbacks"] = [instantiate_any(c) for c in lc.get("callbacks", [])]
        lightning_config.update(lc)
    lightning_config["callbacks"].append(LearningRateMonitor(logging_interval="step"))
    trainer = Fitter(**lightning_config)
    if torch.cuda.is_available():
        torch.cuda.set_device(trainer.local_rank)

    data_module, trainer_wrapper = load_all(config)
    if config.get("unet_gradient_checkpointing", False):
        trainer_wrapper.unet.enable_gradient_checkpointing()
    if "seed" in config:  # after model construction, as the reference does (weight init is not seeded)
        seed_everything(config.seed + trainer.global_rank)
    ckpt_path = config.get("resume_from_checkpoint", None)
    if isinstance(ckpt_path, dict):  # the reference accepts an instantiable here (test_train.py:71-75)
        ckpt_path = instantiate_any(ckpt_path)
    if ckpt_path is not None:
        logger.info(f"Resume from {ckpt_path}...")
    trainer.fit(trainer_wrapper, data_module, ckpt_path=ckpt_path)
