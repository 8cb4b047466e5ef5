"""reference src/duwu/modules/unet_patch.py: the denoiser slot ``UNet2DFromScratch.from_config``.

``config`` may be a preset name or a dict; hub names are resolved to built-in config dicts, nothing is fetched:
  * "stabilityai/stable-diffusion-xl-base-1.0" (+ ``subfolder: unet``) -> the SDXL-shape UNet2DConditionModel
    (uwudiff_amd/unet.py; what both of the reference's own training YAMLs instantiate);
  * "tiny-unet" -> the small UNet of BASELINE.json configs[0];
  * "DiT-S/2" | "DiT-B/2" | "DiT-L/2" | "DiT-XL/2" -> the MI355X-native DiT (uwudiff_amd/dit.py).
The near-zero init of residual-branch output layers (unet_patch.py:34-45) is applied by the model constructors.
"""
from uwudiff_amd.dit import PRESETS, DiT
from uwudiff_amd.unet import UNet2DConditionModel

_UNET_NAMES = {"stabilityai/stable-diffusion-xl-base-1.0", "sdxl", "tiny-unet"}


class UNet2DFromScratch:
    @classmethod
    def from_config(cls, config, subfolder=None, **kwargs):
        if isinstance(config, str):
            if config in PRESETS:
                return DiT.from_config(config, **kwargs)
            if config in _UNET_NAMES:
                return UNet2DConditionModel.from_config(config, **kwargs)
            raise ValueError(f"unknown denoiser config {config!r} (offline registry: "
                             f"{sorted(PRESETS) + sorted(_UNET_NAMES)})")
        cfg = dict(config)
        kind = cfg.pop("architecture", "unet" if "block_out_channels" in cfg else "dit")
        if kind == "dit":
            return DiT.from_config(cfg, **kwargs)
        return UNet2DConditionModel.from_config(cfg, **kwargs)
