"""reference src/duwu/modules/unet_patch.py: the denoiser slot ``UNet2DFromScratch.from_config``.

``config`` may be a preset name or a dict.  DiT presets ("DiT-S/2", "DiT-B/2", "DiT-L/2", "DiT-XL/2") build the
MI355X-native DiT (uwudiff_amd/dit.py).  Hub names are never fetched: the SDXL UNet name used by the reference's
own YAMLs is recognised and reported as not yet available on the HIP path (SURVEY.md section 8f ranks it "next").
"""
from uwudiff_amd.dit import PRESETS, DiT

_HUB_UNETS = {"stabilityai/stable-diffusion-xl-base-1.0", "runwayml/stable-diffusion-v1-5"}


class UNet2DFromScratch:
    @classmethod
    def from_config(cls, config, subfolder=None, **kwargs):
        if isinstance(config, str):
            if config in PRESETS:
                return DiT.from_config(config, **kwargs)
            if config in _HUB_UNETS:
                raise NotImplementedError(
                    f"{config!r} (SDXL-shape UNet2DConditionModel) has no HIP implementation yet in this build; "
                    "use a DiT preset (e.g. config: DiT-S/2).  No hub access is attempted.")
            raise ValueError(f"unknown denoiser config {config!r}")
        cfg = dict(config)
        kind = cfg.pop("architecture", "dit")
        if kind != "dit":
            raise NotImplementedError(f"architecture {kind!r}")
        return DiT.from_config(cfg, **kwargs)
