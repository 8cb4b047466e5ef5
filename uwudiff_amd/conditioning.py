"""Conditioning front-end.

The reference encodes captions with hub-hosted CLIP / T5 weights (src/duwu/modules/text_encoders.py), which cannot
exist offline: the text MODELS here are deterministic synthetic stand-ins with the transformers call convention.
What the reference itself owns -- ``ConcatTextEncoders.forward``'s bucket / feature-concat / pad / mask / pooled logic
(text_encoders.py:139-264) and ``encode(nested=True)`` (:101-137) -- is restated in full and assembled on the device.
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
    """Stand-in for the ``transformers`` text models the reference wraps (hub weights, unavailable offline): a hashed
    embedding table + position table and ``n_layers`` fixed random mixing layers, frozen.  It keeps the CALL CONVENTION
    ``ConcatTextEncoders.forward`` relies on (text_encoders.py:169-187), ``text_model(input_ids, attention_mask=...,
    output_hidden_states=True, return_dict=False)``:

      kind "clip"      -> (last_hidden_state, pooled, hidden_states tuple)        (CLIPTextModel / ...WithProjection)
      kind "t5"        -> (last_hidden_state, hidden_states tuple)                (T5EncoderModel: no pooled output)
    """

    def __init__(self, hidden=768, seed=0, kind="clip", n_layers=3, vocab=4096, max_pos=512):
        super().__init__()
        g = torch.Generator().manual_seed(1000 + seed)
        self.hidden, self.kind = hidden, kind
        self.register_buffer("table", torch.randn(vocab, hidden, generator=g) * 0.5)
        self.register_buffer("pos", torch.randn(max_pos, hidden, generator=g) * 0.1)
        self.register_buffer("mix", torch.randn(n_layers, hidden, generator=g) * 0.2 + 1.0)  # per-layer channel gains
        self.final_layer_norm = nn.LayerNorm(hidden)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, **kw):
        return cls(hidden=HIDDEN.get(subfolder, 768), seed=len(str(subfolder)), **kw)

    def forward(self, input_ids, attention_mask=None, output_hidden_states=False, return_dict=False, **kw):
        h = self.table[input_ids % self.table.shape[0]] + self.pos[None, : input_ids.shape[1]]
        hidden = [h]
        for l in range(self.mix.shape[0]):
            h = torch.tanh(h * self.mix[l]) + 0.5 * h
            hidden.append(h)
        last = self.final_layer_norm(h)
        if self.kind == "t5":
            return (last, tuple(hidden)) if output_hidden_states else (last,)
        eos = input_ids.argmax(dim=-1)  # CLIP pools at the eos token (the highest id)
        pooled = last[torch.arange(last.shape[0], device=last.device), eos]
        return (last, pooled, tuple(hidden)) if output_hidden_states else (last, pooled)


class SyntheticCLIPTextModel(SyntheticTextModel):
    """What ``transformers.CLIPTextModel`` resolves to (uwudiff_amd/config.py ALIASES): the reference recomputes
    ``normed_embedding = final_layer_norm(hidden_states[layer_idx])`` for every plain ``CLIPTextModel`` -- an ``isinstance``
    test (text_encoders.py:190-192) that ``CLIPTextModelWithProjection`` does not pass -- so the class a YAML names decides
    it.  Both shipped YAMLs name ``CLIPTextModel`` for both SDXL encoders."""

    def __init__(self, hidden=768, seed=0, kind="clip_sd1", **kw):
        super().__init__(hidden=hidden, seed=seed, kind=kind, **kw)


class _LatentDist:
    def __init__(self, mean, logvar):
        self.mean, self.logvar = mean, logvar

    def sample(self, generator=None):
        return self.mean + torch.exp(0.5 * self.logvar) * torch.randn(self.mean.shape, generator=generator,
                                                                      device=self.mean.device, dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class _EncoderOutput:
    def __init__(self, dist):
        self.latent_dist = dist


class SyntheticVAE(nn.Module):
    """Offline stand-in for ``diffusers.AutoencoderKL`` in the trainer's VAE slot (reference trainer.py:136,241-244; hub
    weights cannot exist here): ``encode(x).latent_dist.sample()`` with the 8x spatial reduction and 4 latent channels of
    the SDXL VAE -- a frozen average pool + fixed channel mix, small fixed variance."""

    def __init__(self, in_channels=3, latent_channels=4, factor=8, seed=7):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.factor = factor
        self.register_buffer("mix", torch.randn(latent_channels, in_channels, generator=g))
        self.register_buffer("bias", torch.randn(latent_channels, generator=g) * 0.1)

    @classmethod
    def from_pretrained(cls, *a, **kw):
        return cls()

    @torch.no_grad()
    def encode(self, x):
        p = torch.nn.functional.avg_pool2d(x.float(), self.factor)
        mean = torch.einsum("oc,bchw->bohw", self.mix, p) + self.bias[None, :, None, None]
        return _EncoderOutput(_LatentDist(mean, torch.full_like(mean, -6.0)))


def _remove_none(xs):
    return [x for x in xs if x is not None]


class ConcatTextEncoders(nn.Module):
    """reference src/duwu/modules/text_encoders.py:41-264 (``ConcatTextEncoders``): several tokenizers / text models, each
    with a ``TextModelExtraConfig`` (``concat_bucket, use_pooled, layer_idx, need_mask, disable_autocast``).  Encoders of
    one bucket are concatenated on the FEATURE axis, buckets on the SEQUENCE axis after zero-padding the narrower ones to
    the widest (``:196-262``); pooled vectors of the ``use_pooled`` encoders are concatenated on the feature axis; the
    mask of a bucket is its first ``need_mask`` encoder's attention mask (ones for buckets without one, once any exists).

        SDXL: [CLIP-L, openCLIP-G], buckets [0, 0], layer_idx [-1, -2], pooled from the second  -> ctx [B, 77, 2048]
        SD3:  [CLIP-L, openCLIP-G, T5], buckets [0, 0, 1]                                         -> ctx [B, 77 + S5, 4096]

    The text models are synthetic (frozen, ``no_grad``, hub weights do not exist offline); the assembly above -- the part
    the reference owns -- runs on the device through ``uwu_ctx_place``."""

    def __init__(self, tokenizers=(), text_model_and_configs=(), zero_for_padding=True, max_length=256,
                 use_normed_ctx=False):
        super().__init__()
        from .config import instantiate_any

        self.tokenizers = [SyntheticTokenizer(t) for t in tokenizers]
        for t in self.tokenizers:
            if t.model_max_length > max_length:
                t.model_max_length = max_length
        models, self.configs = [], []
        self.max_bucket = 0
        for model, extra in text_model_and_configs:
            models.append(model if isinstance(model, nn.Module) else instantiate_any(model))
            cfg = dict(concat_bucket=0, use_pooled=False, layer_idx=-1, need_mask=False, disable_autocast=False)
            cfg.update(dict(extra) if extra is not None else {})
            self.configs.append(cfg)
            self.max_bucket = max(self.max_bucket, cfg["concat_bucket"])
        self.text_models = nn.ModuleList(models)
        self.zero_for_padding = zero_for_padding
        self.use_normed_ctx = use_normed_ctx

    @property
    def device(self):
        return next(self.text_models[0].buffers()).device

    @torch.no_grad()
    def forward(self, tokenizer_outputs):
        from . import lib as L

        dev = self.device
        nb = self.max_bucket + 1
        masks = [None] * nb
        parts = [[] for _ in range(nb)]        # per bucket: (embedding, normed, attention mask) of each encoder
        pooled = [[] for _ in range(nb)]
        for tokens, model, cfg in zip(tokenizer_outputs, self.text_models, self.configs):
            bucket = cfg["concat_bucket"]
            ids = tokens["input_ids"].to(dev)
            am = tokens["attention_mask"].to(dev).long().contiguous()
            if masks[bucket] is None and cfg["need_mask"]:
                masks[bucket] = am
            normed, pool, *rest = model(ids, attention_mask=am, output_hidden_states=True, return_dict=False)
            if len(rest):            # CLIP: (last_hidden_state, pooled, hidden_states)
                emb = rest[-1][cfg["layer_idx"]]
            else:                    # T5: (last_hidden_state, hidden_states) -- no pooled output
                emb, pool = pool[-1], None
            if getattr(model, "kind", "") == "clip_sd1":  # text_encoders.py:190-192 (plain CLIPTextModel)
                normed = model.final_layer_norm(emb)
            parts[bucket].append((emb.float().contiguous(), normed.float().contiguous(), am))
            if cfg["use_pooled"] and pool is not None:
                pooled[bucket].append(pool.float())
        used = [b for b in range(nb) if parts[b]]
        widths = {b: sum(e.shape[-1] for e, _, _ in parts[b]) for b in used}
        seqs = {b: parts[b][0][0].shape[1] for b in used}
        B = parts[used[0]][0][0].shape[0]
        F_total, S_total = max(widths.values()), sum(seqs.values())
        emb_out = torch.zeros(B, S_total, F_total, device=dev)   # zero fill == F.pad of the narrower buckets (:216-231)
        nrm_out = torch.zeros(B, S_total, F_total, device=dev)
        s_off = 0
        for b in used:
            f_off = 0
            for emb, normed, am in parts[b]:
                S, F = emb.shape[1], emb.shape[2]
                for src, dst in ((emb, emb_out), (normed, nrm_out)):
                    if src.is_cuda:  # training: conditioning is assembled on the device
                        L.call("uwu_ctx_place", L.ptr(src), L.dt(src), L.ptr(am) if self.zero_for_padding else None,
                               L.ptr(dst), B, S, F, S_total, F_total, s_off, f_off, L.stream())
                    else:  # host-side caption preparation (dataset caching, config checks): plain indexing, no kernels
                        dst[:, s_off:s_off + S, f_off:f_off + F] = src * am[..., None] if self.zero_for_padding else src
                f_off += F
            s_off += seqs[b]
        if any(m is not None for m in masks):
            attn = torch.cat([masks[b] if masks[b] is not None else torch.ones(B, seqs[b], device=dev, dtype=torch.long)
                              for b in used], dim=1)
        else:
            attn = None
        pools = [torch.cat(p, dim=-1) for p in pooled if p]
        pooled_out = torch.cat(pools, dim=-1) if pools else None
        return emb_out, nrm_out, pooled_out, attn

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
