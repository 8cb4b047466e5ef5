"""Synthetic conditioning front-end (the frozen, ``no_grad`` text path is OUT OF SCOPE).

The reference encodes captions with hub-hosted CLIP weights (src/duwu/modules/text_encoders.py:139-264), which
cannot exist offline.  This module keeps the *interface* -- tokenizers list, ``forward(tokenizer_outputs) ->
(embedding [B,77,sum(hidden)], normed_embedding, pooled [B,hidden_last], attn_mask|None)`` -- and produces
deterministic pseudo-embeddings from the token ids, so the denoiser receives tensors of the reference's shapes.
"""
import hashlib

import torch
import torch.nn as nn

HIDDEN = {"text_encoder": 768, "text_encoder_2": 1280}


class SyntheticTokenizer:
    """Whitespace tokenizer hashed into a 49408-entry vocabulary; CLIP-style bos/eos/pad to 77."""

    model_max_length = 77
    vocab_size = 49408
    pad_token = "<|endoftext|>"
    eos_token = "<|endoftext|>"

    def __init__(self, name="synthetic"):
        self.name = name

    def _ids(self, text):
        ids = [49406]
        for w in text.lower().split():
            ids.append(int(hashlib.md5(w.encode()).hexdigest(), 16) % 49000 + 300)
        ids = ids[: self.model_max_length - 1] + [49407]
        n = len(ids)
        ids = ids + [49407] * (self.model_max_length - n)
        mask = [1] * n + [0] * (self.model_max_length - n)
        return ids, mask

    def __call__(self, text, padding="max_length", truncation=True, return_tensors="pt", **kw):
        texts = [text] if isinstance(text, str) else list(text)
        pairs = [self._ids(t) for t in texts]
        return {"input_ids": torch.tensor([p[0] for p in pairs]), "attention_mask": torch.tensor([p[1] for p in pairs])}


class SyntheticTextModel(nn.Module):
    """Stand-in for ``transformers.CLIPTextModel``: hashed embedding table + position table, frozen."""

    def __init__(self, hidden=768, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(1000 + seed)
        self.hidden = hidden
        self.register_buffer("table", torch.randn(4096, hidden, generator=g) * 0.5)
        self.register_buffer("pos", torch.randn(77, hidden, generator=g) * 0.1)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, **kw):
        return cls(hidden=HIDDEN.get(subfolder, 768), seed=len(str(subfolder)))

    def forward(self, input_ids, attention_mask=None, **kw):
        emb = self.table[input_ids % 4096] + self.pos[None, : input_ids.shape[1]]
        pooled = emb.mean(dim=1)
        return emb, pooled


class ConcatTextEncoders(nn.Module):
    def __init__(self, tokenizers=(), text_model_and_configs=(), zero_for_padding=True, max_length=256,
                 use_normed_ctx=False):
        super().__init__()
        from .config import instantiate_any

        self.tokenizers = [SyntheticTokenizer(t) for t in tokenizers]
        models, self.configs = [], []
        for model, extra in text_model_and_configs:
            models.append(model if isinstance(model, nn.Module) else instantiate_any(model))
            self.configs.append(dict(extra) if extra is not None else {})
        self.text_models = nn.ModuleList(models)
        self.zero_for_padding = zero_for_padding
        self.use_normed_ctx = use_normed_ctx

    @torch.no_grad()
    def forward(self, tokenizer_outputs):
        embs, pooled = [], None
        for tok, model, cfg in zip(tokenizer_outputs, self.text_models, self.configs):
            dev = model.table.device
            e, p = model(tok["input_ids"].to(dev))
            if self.zero_for_padding:
                e = e * tok["attention_mask"].to(dev)[..., None]
            embs.append(e.float())
            if cfg.get("use_pooled", False):
                pooled = p.float()
        emb = torch.cat(embs, dim=-1)  # SDXL: CLIP-L 768 (+) bigG 1280 on the feature axis (text_encoders.py:212)
        normed = torch.nn.functional.layer_norm(emb, emb.shape[-1:])
        return emb, normed, pooled, None

    def tokenize(self, text, **kw):
        return [t(text, **kw) for t in self.tokenizers]

    def encode(self, text, nested: bool = False, pad_to_n_elements=None, **kw):
        """reference text_encoders.py:102-137.  ``nested``: ``text`` is a list of caption lists (one list per image);
        the per-caption contexts are concatenated per image on the sequence axis and padded to the longest image
        (``duwu.utils.aggregation``, HIP kernels), the pooled vector is the first caption's."""
        if not nested:
            return self.forward(self.tokenize(text, **kw))
        from duwu.utils.aggregation import aggregate_embeddings

        n_per_image = [len(t) for t in text]
        flat = [c for t in text for c in t]
        embs, normed, pools, masks = self.forward(self.tokenize(flat, **kw))
        embs = aggregate_embeddings(embs, n_per_image, mode="concat", pad_to_n_elements=pad_to_n_elements)
        normed = aggregate_embeddings(normed, n_per_image, mode="concat", pad_to_n_elements=pad_to_n_elements)
        if pools is not None:  # only the first provided caption is used for pooling
            pools = aggregate_embeddings(pools, n_per_image, mode="first")
        if masks is not None:
            masks = aggregate_embeddings(masks, n_per_image, mode="concat", pad_to_n_elements=pad_to_n_elements)
        return embs, normed, pools, masks
