"""`TextDataset` of this tree (code/coco/stackgan/miscc/datasets.py:25-217): see ...datasets.CocoTextDataset."""
from ...datasets import CocoTextDataset as TextDataset, crop_imgs, image_transform  # noqa: F401
