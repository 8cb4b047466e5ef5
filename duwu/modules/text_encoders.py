"""reference src/duwu/modules/text_encoders.py: ``ConcatTextEncoders`` (assembly logic restated in
uwudiff_amd/conditioning.py; the text models themselves are synthetic offline stand-ins), ``TextModelExtraConfig``."""
from dataclasses import dataclass

from uwudiff_amd.conditioning import ConcatTextEncoders, SyntheticTextModel, SyntheticTokenizer  # noqa: F401

BaseTextEncoder = ConcatTextEncoders


@dataclass
class TextModelExtraConfig:  # text_encoders.py:29-36
    concat_bucket: int = 0
    use_pooled: bool = False
    layer_idx: int = -1
    need_mask: bool = False
    disable_autocast: bool = False

    def keys(self):
        return self.__dataclass_fields__.keys()

    def __getitem__(self, k):
        return getattr(self, k)
