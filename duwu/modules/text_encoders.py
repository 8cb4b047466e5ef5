"""reference src/duwu/modules/text_encoders.py (interface only; synthetic offline provider)."""
from uwudiff_amd.conditioning import ConcatTextEncoders, SyntheticTextModel, SyntheticTokenizer  # noqa: F401

BaseTextEncoder = ConcatTextEncoders
