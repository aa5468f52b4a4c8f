"""Synthetic stand-ins for the three trees' `TextDataset`s: items have the structure each reference
`__getitem__` returns (S/miscc/datasets.py:185-213, C/miscc/datasets.py:114-148, M/miscc/datasets.py:73-88),
so the default-collated minibatch is what the reference train loop unpacks."""
import torch
import torch.utils.data as data

from . import synthetic


class SyntheticDataset(data.Dataset):
    def __init__(self, tree, stage=1, length=64, seed=0, text_dim=1024):
        self.tree, self.stage, self.length, self.seed, self.text_dim = tree, stage, length, seed, text_dim

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        b = synthetic.make_batch(self.tree, 1, stage=self.stage, seed=self.seed + index, text_dim=self.text_dim)
        img = b["real_imgs"][0]
        if self.tree == "clevr":
            return img, [b["tm"][0], b["tmi"][0]], b["label_one_hot"][0], index
        if self.tree == "mnist":
            return img, b["bbox"][0], b["label_one_hot"][0]
        label = b["label_one_hot"][0].argmax(-1, keepdim=True).float()
        label[label == 80] = -1                                   # absent objects are stored as -1
        bbox = [b["bbox"][0], b["bbox_s2"][0]] if self.stage == 2 else b["bbox"][0]
        return img, bbox, label, b["txt_embedding"][0]
