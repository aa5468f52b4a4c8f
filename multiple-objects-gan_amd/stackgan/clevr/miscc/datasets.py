"""`TextDataset` of this tree (code/clevr/miscc/datasets.py:44-145): see ...datasets.ClevrTextDataset."""
from ...datasets import CLEVR_COLORS as color_dict, CLEVR_SHAPES as shape_dict  # noqa: F401
from ...datasets import ClevrTextDataset as TextDataset, image_transform  # noqa: F401
