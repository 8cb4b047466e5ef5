"""reference src/duwu/data/base.py: synthetic dataset + collate 5-tuple + data module (no Lightning)."""
import torch
import torch.utils.data as Data

from duwu.loader import load_any


class UwUBaseDataset(Data.Dataset):
    @staticmethod
    def collate(batch):
        # (samples, captions, tokenizer_outputs, added_cond, cross_attention_kwargs)  -- base.py:11-31
        samples = torch.stack([x["sample"] for x in batch])
        caption = [x["caption"] for x in batch]
        tokenizer_outs = [x["tokenizer_out"] for x in batch]
        add_time_ids = torch.stack([x["add_time_ids"] for x in batch]).float()
        tokenizer_outputs = []
        for per_tok in zip(*tokenizer_outs):
            tokenizer_outputs.append({
                "input_ids": torch.concat([x["input_ids"] for x in per_tok]),
                "attention_mask": torch.concat([x["attention_mask"] for x in per_tok]),
            })
        return samples, caption, tokenizer_outputs, {"time_ids": add_time_ids}, {}


class DummyDataset(UwUBaseDataset):
    def __init__(self, sample_size=(3, 1024, 1024), n_samples=100, tokenizers=(), **kwargs):
        sample_size = tuple(sample_size)
        self.samples = [torch.randn(sample_size) for _ in range(n_samples)]
        self.tokenizers = list(tokenizers) if isinstance(tokenizers, (list, tuple)) else [tokenizers]

    def set_tokenizers(self, tokenizers):
        self.tokenizers = tokenizers

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        caption = "DUMMY TEST"
        return {
            "sample": self.samples[index],
            "caption": caption,
            "tokenizer_out": [t(caption, padding="max_length", truncation=True, return_tensors="pt")
                              for t in self.tokenizers],
            "add_time_ids": torch.tensor([1024, 1024, 0, 0, 1024, 1024]),  # base.py:73
        }


class TrainDataModule:
    def __init__(self, dataset_config, dataloader_config):
        self.dataset_config = dataset_config
        self.dataloader_config = dict(dataloader_config)

    def setup(self, stage: str = "fit"):
        self.dataset = load_any(self.dataset_config)
        if hasattr(self, "tokenizers"):
            self.dataset.set_tokenizers(self.tokenizers)

    def train_dataloader(self):
        cfg = dict(self.dataloader_config)
        # the synthetic samples are in-memory tensors: worker processes would only add IPC copies
        cfg["num_workers"] = 0
        return Data.DataLoader(self.dataset, collate_fn=self.dataset.collate, **cfg)

    def set_tokenizers(self, tokenizers):
        self.tokenizers = tokenizers
