"""TEST INFRASTRUCTURE: CPU restatement of reference src/duwu/utils/aggregation.py (plain loops), pinned by
tests/golden/aggregation_*.npz which oracle/make_golden.py generated from the reference file itself."""
import torch


def concat(embeddings, n_elements, pad_value=0, pad_to_n_elements=None):
    """aggregation.py:15-39."""
    max_n = pad_to_n_elements or max(n_elements)
    seq = embeddings.shape[1]
    out = embeddings.new_full((len(n_elements), max_n * seq, *embeddings.shape[2:]), pad_value)
    s = 0
    for b, n in enumerate(n_elements):
        out[b, : n * seq] = embeddings[s:s + n].flatten(end_dim=1)
        s += n
    return out


def split(cat, n_elements, sequence_length):
    """aggregation.py:111-171."""
    parts = [cat[b, : n * sequence_length].reshape(n, sequence_length, *cat.shape[2:]) for b, n in enumerate(n_elements)]
    return torch.cat(parts, 0)


def first(embeddings, n_elements):
    """aggregation.py:174-185."""
    out, s = [], 0
    for n in n_elements:
        out.append(embeddings[s])
        s += n
    return torch.stack(out)
