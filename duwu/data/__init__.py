from .base import DummyDataset, TrainDataModule, UwUBaseDataset  # noqa: F401
