#!/usr/bin/env python
"""Golden fixture tests/golden/stackgan_data.npz: the REFERENCE's StackGAN-family data code run on CPU through
ref_shim.load_tree on the deterministic inputs of tests/stackgan_data_cases.py:

  * coco  code/coco/stackgan/miscc/datasets.py: TextDataset.crop_imgs (100-183) on seeded (stage, image, boxes) cases -- the
    scaled box sets (stage 2: both) and corner samples + sum of the crop; and TextDataset.__getitem__ (185-213) of a tiny tree
    for stage 1 and 2 (python `random` and numpy seeded per item): boxes, label, the drawn embedding, image probe;
  * clevr code/clevr/miscc/datasets.py: TextDataset.__getitem__ (114-142) of a tiny tree: image probe (the flip), both affine
    matrices, the 4 + 9 one-hot label, the boxes;
  * mnist code/multi-mnist/miscc/datasets.py: TextDataset.__getitem__ (73-88): boxes, label, image probe.

torchvision.transforms is not importable here: the reference datasets get THIS package's image_transform as their `transform`
argument, so the fixture pins file formats, the random draws and their order, crop / flip and the box arithmetic -- not the
bilinear resize.   Usage: python tests/golden/make_golden_stackgan_data.py"""
import importlib
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import ref_shim                                   # noqa: E402
import stackgan_data_cases as C                   # noqa: E402
import mogan_loader                               # noqa: E402
mogan_loader.load()
from mogan_amd.stackgan.datasets import image_transform  # noqa: E402


def seed(s):
    random.seed(s)
    np.random.seed(s)


def main():
    out = {}
    tmp = tempfile.mkdtemp()
    # ---------------------------------------------------------------- coco
    ref_shim.load_tree("coco")
    ref = importlib.import_module("miscc.datasets")
    boxes1, boxes2, probes = [], [], []
    for case in range(C.N_CROP):
        stage, img, b = C.crop_case(case)
        stand_in = types.SimpleNamespace(imsize=64 if stage == 1 else 256, max_objects=3, stage=stage)
        seed(700 + case)
        crop, scaled = ref.TextDataset.crop_imgs(stand_in, img, b)
        sets = [scaled, scaled] if stage == 1 else scaled
        boxes1.append(sets[0]); boxes2.append(sets[1]); probes.append(C.item_probe(crop))
    out["crop_b1"], out["crop_b2"], out["crop_probe"] = np.stack(boxes1), np.stack(boxes2), np.asarray(probes)
    data_dir, img_dir, _ = C.build_coco_tree(tmp)
    for stage in (1, 2):
        resize, imsize = (76, 64) if stage == 1 else (268, 256)
        ds = ref.TextDataset(data_dir, img_dir, imsize, split="train", transform=image_transform(resize), crop=True, stage=stage)
        bb, lab, emb, pr = [], [], [], []
        for i in range(len(ds)):
            seed(800 + 10 * stage + i)
            img, bbox, label, e = ds[i]
            bb.append(np.stack(bbox) if stage == 2 else bbox); lab.append(label); emb.append(e); pr.append(C.item_probe(img))
        out["coco%d_bbox" % stage], out["coco%d_label" % stage] = np.stack(bb), np.stack(lab)
        out["coco%d_emb" % stage], out["coco%d_probe" % stage] = np.stack(emb), np.asarray(pr)
    # ---------------------------------------------------------------- clevr
    ref_shim.load_tree("clevr")
    ref = importlib.import_module("miscc.datasets")
    data_dir = C.build_clevr_tree(tmp)
    ds = ref.TextDataset(data_dir, 64, split="train", transform=image_transform())
    order = sorted(range(len(ds)), key=lambda i: ds.filenames[i])
    pr, tms, labs, bbs = [], [], [], []
    for n, i in enumerate(order):
        seed(900 + n)
        img, (tm, tmi), label, bbox = ds[i]
        pr.append(C.item_probe(img)); tms.append(np.stack([tm.numpy(), tmi.numpy()])); labs.append(label.numpy()); bbs.append(bbox)
    out["clevr_probe"], out["clevr_mats"], out["clevr_label"], out["clevr_bbox"] = np.asarray(pr), np.stack(tms), np.stack(labs), \
        np.stack(bbs)
    # ---------------------------------------------------------------- mnist
    ref_shim.load_tree("mnist")
    ref = importlib.import_module("miscc.datasets")
    data_dir = C.build_mnist_tree(tmp)
    ds = ref.TextDataset(data_dir, 64, split="train", transform=image_transform(), crop=True)
    items = [ds[i] for i in range(len(ds))]
    out["mnist_probe"] = np.asarray([C.item_probe(it[0]) for it in items])
    out["mnist_bbox"], out["mnist_label"] = np.stack([it[1] for it in items]), np.stack([it[2] for it in items])
    path = os.path.join(HERE, "stackgan_data.npz")
    np.savez_compressed(path, **out)
    print("wrote stackgan_data.npz %.1f KB, %d arrays" % (os.path.getsize(path) / 1024, len(out)))
    for k, v in out.items():
        print("  %-14s %s %s" % (k, v.dtype, v.shape))


if __name__ == "__main__":
    main()
