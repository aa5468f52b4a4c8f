"""`TextDataset` of this tree (code/multi-mnist/miscc/datasets.py:25-86): see ...datasets.MnistTextDataset."""
from ...datasets import MnistTextDataset as TextDataset, image_transform  # noqa: F401
