#!/usr/bin/env python3
"""Golden vectors for the conditioning front-end from the REFERENCE's own ``ConcatTextEncoders.forward``
(/root/reference/src/duwu/modules/text_encoders.py:139-264), run in the build container.

TEST INFRASTRUCTURE.  The reference file is loaded by path; the third-party packages it imports that are not installed
here (lightning, omegaconf, hydra) are replaced by empty stubs -- none of them takes part in ``forward`` -- while
``transformers`` (installed) is imported for real.  The text MODELS are this build's deterministic synthetic stand-ins
(hub weights do not exist offline); what is pinned is the reference's bucket / concat / pad / mask / pooled ASSEMBLY.
Only inputs and outputs (data) are written: tests/golden/te_*.npz.

    python oracle/make_golden_te.py
"""
import importlib.util
import json
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

REF = "/root/reference/src/duwu"
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference_text_encoders():
    class LightningModule(nn.Module):  # only .device / .dtype are used by forward()
        @property
        def device(self):
            return next(self.buffers()).device

        @property
        def dtype(self):
            return torch.float32

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("lightning", LightningModule=LightningModule, LightningDataModule=object)
    stub("lightning.pytorch", LightningModule=LightningModule)
    stub("lightning.pytorch.utilities", rank_zero_only=lambda f: f)
    stub("omegaconf", DictConfig=dict, ListConfig=list, OmegaConf=object)
    stub("hydra")
    stub("hydra.utils", instantiate=lambda *a, **k: None)
    for name in ("duwu", "duwu.modules"):
        stub(name)
    mods = {}
    for name, fn in (("duwu.utils", "utils/__init__.py"), ("duwu.utils.aggregation", "utils/aggregation.py"),
                     ("duwu.loader", "loader.py"), ("duwu.modules.text_encoders", "modules/text_encoders.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fn),
                                                      submodule_search_locations=[] if fn.endswith("__init__.py") else None)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["duwu.modules.text_encoders"]


CASES = {
    # name: (zero_for_padding, [(hidden, seed, extra config), ...]); widths scaled down 16x from CLIP-L / bigG (768 / 1280)
    # to keep the fixtures small -- the assembly logic does not depend on them
    "te_sdxl": (False, [(48, 12, dict(concat_bucket=0, use_pooled=False, need_mask=False, layer_idx=-2)),
                        (80, 14, dict(concat_bucket=0, use_pooled=True, need_mask=False, layer_idx=-2))]),
    "te_sdxl_zero_pad_mask": (True, [(48, 12, dict(concat_bucket=0, use_pooled=True, need_mask=True, layer_idx=-1)),
                                     (80, 14, dict(concat_bucket=0, use_pooled=True, need_mask=False, layer_idx=-2))]),
    "te_two_buckets": (True, [(48, 1, dict(concat_bucket=0, use_pooled=True, need_mask=False, layer_idx=-1)),
                              (80, 2, dict(concat_bucket=0, use_pooled=False, need_mask=False, layer_idx=-3)),
                              (96, 3, dict(concat_bucket=1, use_pooled=True, need_mask=True, layer_idx=-2))]),
}
CAPTIONS = ["a cat on a mat", "", "an oil painting of a ship in a storm with dramatic lighting and seagulls", "dog"]


def main():
    from uwudiff_amd.conditioning import SyntheticTextModel, SyntheticTokenizer

    te = load_reference_text_encoders()
    os.makedirs(OUT, exist_ok=True)
    tok = SyntheticTokenizer()(CAPTIONS)
    for name, (zero_pad, specs) in CASES.items():
        enc = te.ConcatTextEncoders.__new__(te.ConcatTextEncoders)
        nn.Module.__init__(enc)  # (the reference constructor downloads tokenizers / weights from the hub)
        enc.tokenizers = []
        enc.text_models = nn.ModuleList([SyntheticTextModel(h, seed) for h, seed, _ in specs])
        enc.configs = [te.TextModelExtraConfig(**c) for _, _, c in specs]
        enc.max_bucket = max(c["concat_bucket"] for _, _, c in specs)
        enc.zero_for_padding = zero_pad
        enc.use_normed_ctx = False
        with torch.no_grad():
            emb, normed, pooled, mask = enc.forward([tok] * len(specs))
        meta = dict(kind="text_encoders", zero_for_padding=zero_pad, captions=CAPTIONS,
                    models=[dict(hidden=h, seed=s, config=c) for h, s, c in specs], has_mask=mask is not None)
        arrays = dict(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"], embedding=emb, normed=normed,
                      pooled=pooled)
        if mask is not None:
            arrays["mask"] = mask
        arrays = {k: v.detach().cpu().numpy() for k, v in arrays.items()}
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        print("wrote", name, {k: v.shape for k, v in arrays.items() if k != "meta"})


if __name__ == "__main__":
    main()
