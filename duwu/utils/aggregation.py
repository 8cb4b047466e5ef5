"""reference src/duwu/utils/aggregation.py (same names and argument meaning), on the HIP device.

Ragged per-caption embeddings ``[sum(n_elements), seq, ...]`` <-> padded per-image ``[B, max_n * seq, ...]``.  The
reference scatters through Python loops / advanced indexing; here each direction is one byte-moving kernel
(``uwu_aggregate_concat / _split / _first``).  No CPU fallback: tensors must live on the device.
"""
import struct

import torch

from uwudiff_amd import lib as L


def _starts(n_elements, device):
    n = n_elements.tolist() if torch.is_tensor(n_elements) else list(n_elements)
    acc, out = 0, [0]
    for v in n:
        acc += int(v)
        out.append(acc)
    return n, torch.tensor(out, dtype=torch.int32, device=device)


def _pad_bits(dtype, value):
    if dtype == torch.float32:
        return struct.unpack("<I", struct.pack("<f", float(value)))[0]
    if dtype == torch.bfloat16:
        return struct.unpack("<I", struct.pack("<f", float(torch.tensor(float(value)).bfloat16())))[0] >> 16
    if dtype == torch.float16:
        return struct.unpack("<H", struct.pack("<e", float(value)))[0]
    if dtype == torch.float64:
        return struct.unpack("<Q", struct.pack("<d", float(value)))[0]
    if dtype in (torch.int64, torch.int32, torch.int16, torch.int8, torch.uint8, torch.bool):
        return int(value) & ((1 << (8 * torch.empty((), dtype=dtype).element_size())) - 1)
    raise ValueError(f"unsupported dtype {dtype}")


def aggregate_embeddings(embeddings: torch.Tensor, n_elements, mode: str, **kwargs):
    """aggregation.py:6-13."""
    if mode == "concat":
        return concat_aggregate_embeddings_vectorize(embeddings, n_elements, **kwargs)
    if mode == "first":
        return first_aggregate_embeddings(embeddings, n_elements, **kwargs)
    raise ValueError(f'Invalid aggregation mode "{mode}"')


def concat_aggregate_embeddings(embeddings, n_elements, pad_value: float = 0, pad_to_n_elements=None):
    """aggregation.py:15-39 (same result as the vectorised form)."""
    return concat_aggregate_embeddings_vectorize(embeddings, n_elements, pad_value, pad_to_n_elements)


def concat_aggregate_embeddings_vectorize(embeddings, n_elements, pad_value: float = 0, pad_to_n_elements=None,
                                          batch_indices_flat=None, positions_flat=None, cat_embeddings=None):
    """aggregation.py:64-108.  ``batch_indices_flat`` / ``positions_flat`` (the reference's precomputed scatter indices)
    are accepted and ignored; a preallocated ``cat_embeddings`` is overwritten completely (padding included)."""
    emb = embeddings.contiguous()
    n, starts = _starts(n_elements, emb.device)
    assert sum(n) == emb.shape[0], "sum(n_elements) must equal len(embeddings)"
    max_n = int(pad_to_n_elements) if pad_to_n_elements else max(n)
    assert max_n >= max(n)
    seq = emb.shape[1]
    out_shape = (len(n), max_n * seq, *emb.shape[2:])
    out = cat_embeddings if cat_embeddings is not None else torch.empty(out_shape, dtype=emb.dtype, device=emb.device)
    assert tuple(out.shape) == out_shape and out.is_contiguous() and out.dtype == emb.dtype
    unit = emb[0].numel() * emb.element_size()
    L.call("uwu_aggregate_concat", L.ptr(emb), L.ptr(starts), L.ptr(out), len(n), max_n, unit, emb.element_size(),
           _pad_bits(emb.dtype, pad_value), L.stream())
    return out


def split_aggregate_embeddings(cat_embeddings, n_elements, sequence_length: int):
    """aggregation.py:111-171: back to ``[sum(n_elements), sequence_length, ...]``."""
    cat = cat_embeddings.contiguous()
    n, starts = _starts(n_elements, cat.device)
    B, max_total = cat.shape[0], cat.shape[1]
    assert B == len(n) and max_total % sequence_length == 0
    max_n = max_total // sequence_length
    out = torch.empty((sum(n), sequence_length, *cat.shape[2:]), dtype=cat.dtype, device=cat.device)
    unit = out[0].numel() * out.element_size() if sum(n) else sequence_length * cat[0, 0].numel() * cat.element_size()
    L.call("uwu_aggregate_split", L.ptr(cat), L.ptr(starts), L.ptr(out), B, max_n, unit, L.stream())
    return out


def first_aggregate_embeddings(embeddings, n_elements):
    """aggregation.py:174-185."""
    emb = embeddings.contiguous()
    n, starts = _starts(n_elements, emb.device)
    assert sum(n) == emb.shape[0]
    out = torch.empty((len(n), *emb.shape[1:]), dtype=emb.dtype, device=emb.device)
    L.call("uwu_aggregate_first", L.ptr(emb), L.ptr(starts), L.ptr(out), len(n), emb[0].numel() * emb.element_size(),
           L.stream())
    return out
