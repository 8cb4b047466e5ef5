"""reference src/duwu/modules/rope_unet.py:562-608: the two UNet variants a config can name in the denoiser slot.

  * ``HDUNet2DConditionModel.from_config(arch)``  -- UNet2DConditionModel whose residual-branch output layers (attention
    ``to_out``, feed-forward output, resnet ``conv2``, ``conv_out``) start at exactly zero;
  * ``RoPEUNet2DConditionModel.from_config(arch)`` -- the same plus axial RoPE in every attention of every transformer block
    (q always, k in self-attention; learnable per-head log-frequencies ``<attn>.axial_rope.freqs_h / freqs_w``).
``arch``: a config dict (diffusers ``UNet2DConditionModel`` keys) or a preset name of uwudiff_amd/unet.py; a JSON path is read
like the reference does.  Both run on the HIP kernels of uwudiff_amd/unet.py.
"""
import json
import os

from uwudiff_amd.unet import UNet2DConditionModel


def _arch(arch):
    if isinstance(arch, str) and os.path.isfile(arch):
        with open(arch) as f:
            arch = json.load(f)
    if isinstance(arch, dict) and "_target_" in arch:  # the reference's arch files wrap the keyword arguments
        arch = {k: v for k, v in arch.items() if not k.startswith("_")}
    return arch


class HDUNet2DConditionModel:
    @classmethod
    def from_config(cls, arch, **kw):
        return UNet2DConditionModel.from_config(_arch(arch), zero_init=True, **kw)


class RoPEUNet2DConditionModel:
    @classmethod
    def from_config(cls, arch, **kw):
        return UNet2DConditionModel.from_config(_arch(arch), zero_init=True, rope=True, **kw)
