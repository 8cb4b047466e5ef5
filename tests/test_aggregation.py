"""Conditioning front-end: ragged -> padded aggregation, reference
src/duwu/utils/aggregation.py.  Fixtures tests/golden/aggregation_*.npz were produced by the reference file itself
(oracle/make_golden.py loads it by path).  Byte movement: every comparison is bit-exact."""
import pytest
import torch

from tests.golden_util import load

CASES = ["aggregation_a", "aggregation_b", "aggregation_c", "aggregation_d"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    from oracle import aggregation as OA

    meta, d = load(name)
    n, seq = meta["n_elements"], meta["seq"]
    cat = OA.concat(d["emb"], n, meta["pad_value"], meta["pad_to_n_elements"])
    assert torch.equal(cat, d["cat"])
    assert torch.equal(OA.split(d["cat"], n, seq), d["emb"])
    assert torch.equal(OA.first(d["emb"], n), d["first"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_reference_golden(name):
    from duwu.utils import aggregation as A

    meta, d = load(name)
    n, seq = meta["n_elements"], meta["seq"]
    emb = d["emb"].cuda()
    kw = dict(pad_value=meta["pad_value"], pad_to_n_elements=meta["pad_to_n_elements"])
    assert torch.equal(A.aggregate_embeddings(emb, n, "concat", **kw).cpu(), d["cat"])
    assert torch.equal(A.concat_aggregate_embeddings(emb, torch.tensor(n), **kw).cpu(), d["cat"])
    assert torch.equal(A.split_aggregate_embeddings(d["cat"].cuda(), n, seq).cpu(), d["emb"])
    assert torch.equal(A.aggregate_embeddings(emb, n, "first").cpu(), d["first"])
    with pytest.raises(ValueError):
        A.aggregate_embeddings(emb, n, "mean")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.int64, torch.uint8])
def test_hip_round_trip_sdxl_shapes(dtype):
    """[captions, 77, 2048]-shaped contexts (and odd byte widths), against the loop oracle; split(concat(x)) == x."""
    from duwu.utils import aggregation as A
    from oracle import aggregation as OA

    g = torch.Generator().manual_seed(9)
    n = [3, 1, 4, 2, 1, 5]
    shape = (77, 2048) if dtype in (torch.bfloat16, torch.float32) else (77, 3)
    if dtype.is_floating_point:
        emb = torch.randn(sum(n), *shape, generator=g).to(dtype)
    else:
        emb = torch.randint(0, 100, (sum(n), *shape), generator=g).to(dtype)
    pad = 1.25 if dtype.is_floating_point else 7
    cat = A.concat_aggregate_embeddings_vectorize(emb.cuda(), n, pad_value=pad, pad_to_n_elements=6)
    assert torch.equal(cat.cpu(), OA.concat(emb, n, pad, 6))
    assert torch.equal(A.split_aggregate_embeddings(cat, n, 77).cpu(), emb)
    assert torch.equal(A.first_aggregate_embeddings(emb.cuda(), n).cpu(), OA.first(emb, n))


def test_no_cpu_fallback():
    from duwu.utils import aggregation as A
    from uwudiff_amd import lib as L

    with pytest.raises(L.UwuError):
        A.aggregate_embeddings(torch.zeros(3, 2, 4), [1, 2], "concat")


@pytest.mark.gpu
def test_nested_caption_encode():
    """ConcatTextEncoders.encode(nested=True) (reference text_encoders.py:102-137): per-image concatenation of caption
    contexts, first caption's pooled vector."""
    from duwu.modules.text_encoders import ConcatTextEncoders
    from uwudiff_amd.conditioning import SyntheticTextModel

    te = ConcatTextEncoders(tokenizers=["a", "b"], text_model_and_configs=[(SyntheticTextModel(768, 0), {}),
                                                                           (SyntheticTextModel(1280, 1), {"use_pooled": True})]).cuda()
    caps = [["a cat", "on a mat"], ["a dog"], ["x", "y", "z"]]
    emb, normed, pooled, mask = te.encode(caps, nested=True, padding="max_length", truncation=True)
    flat_emb, _, flat_pool, _ = te.encode([c for t in caps for c in t], padding="max_length", truncation=True)
    S = flat_emb.shape[1]
    assert emb.shape == (3, 3 * S, 2048) and pooled.shape == (3, 1280) and mask is None
    assert torch.equal(emb[0, : 2 * S], flat_emb[0:2].flatten(0, 1)) and bool((emb[0, 2 * S:] == 0).all())
    assert torch.equal(emb[1, :S], flat_emb[2]) and torch.equal(emb[2], flat_emb[3:6].flatten(0, 1))
    assert torch.equal(pooled, flat_pool[[0, 2, 3]])
