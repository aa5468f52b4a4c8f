"""Deterministic tiny data trees of the three StackGAN-family TextDatasets (file formats of code/coco/stackgan/miscc/datasets.py,
code/clevr/miscc/datasets.py, code/multi-mnist/miscc/datasets.py) and the seeded cases of the crop fixture: shared by the
generator (tests/golden/make_golden_stackgan_data.py, imports the reference) and the tests (no reference import)."""
import json
import os
import pickle

import numpy as np
import torch

N_CROP = 40
N_ITEMS = 6


def crop_case(case):
    """(stage, image, boxes): stage 1 = (3, 76, 76) -> 64, stage 2 = (3, 268, 268) -> 256; element value = flat index, so the
    crop origin and the flip can be read off corner samples.  Boxes as the loader yields them (float64, -1 = absent)."""
    stage = 1 + (case % 2)
    ori = 76 if stage == 1 else 268
    img = torch.arange(3 * ori * ori, dtype=torch.float32).view(3, ori, ori)
    rng = np.random.RandomState(3000 + case)
    b = np.full((3, 4), -1.0, dtype=np.float64)
    n = 3 - (1 if case % 3 == 0 else 0) - (1 if case % 7 == 0 else 0)
    for k in range(n):
        x, y = rng.uniform(0.0, 0.75), rng.uniform(0.0, 0.75)
        w, h = rng.uniform(0.05, 0.6), rng.uniform(0.05, 0.6)
        if case % 5 == 0 and k == 0:
            x, w = 0.7, 0.45                     # width clamp
        if case % 4 == 0 and k == 1:
            y, h = 0.8, 0.9                      # height clamp
        b[k] = [x, y, w, h]
    return stage, img, b


def _png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path, format="PNG")         # lossless, whatever the file is called


def build_coco_tree(root, n=N_ITEMS, text_dim=16, n_emb=4):
    rng = np.random.RandomState(41)
    data_dir, img_dir = os.path.join(root, "coco"), os.path.join(root, "coco_images")
    for split in ("train", "test"):
        os.makedirs(os.path.join(data_dir, split), exist_ok=True)
    os.makedirs(img_dir, exist_ok=True)
    names = ["COCO_train2014_%012d" % i for i in range(n)]
    for nm in names:
        _png(os.path.join(img_dir, nm + ".jpg"), rng.randint(0, 255, (90, 120, 3), dtype=np.uint8))
    bbox = np.full((n, 3, 4), -1.0, np.float32)
    labels = np.full((n, 3, 1), -1.0, np.float32)
    for i in range(n):
        for k in range(2 + (i % 2)):
            bbox[i, k] = [rng.uniform(0, 0.6), rng.uniform(0, 0.6), rng.uniform(0.2, 0.5), rng.uniform(0.2, 0.5)]
            labels[i, k] = rng.randint(0, 80)
    emb = rng.standard_normal((n, n_emb, text_dim)).astype(np.float32)
    for split in ("train", "test"):
        d = os.path.join(data_dir, split)
        pickle.dump(names, open(os.path.join(d, "filenames.pickle"), "wb"))
        pickle.dump(bbox, open(os.path.join(d, "bboxes.pickle"), "wb"))
        pickle.dump(labels, open(os.path.join(d, "labels.pickle"), "wb"))
        pickle.dump(emb, open(os.path.join(d, "char-CNN-RNN-embeddings.pickle"), "wb"))
    return data_dir, img_dir, dict(names=names, bbox=bbox, labels=labels, emb=emb)


def build_clevr_tree(root, n=N_ITEMS):
    rng = np.random.RandomState(42)
    data_dir = os.path.join(root, "clevr")
    shapes, colors = ["cube", "cylinder", "sphere"], ["gray", "red", "blue", "green", "brown", "purple", "cyan", "yellow"]
    for split in ("train", "test"):
        os.makedirs(os.path.join(data_dir, split, "images"), exist_ok=True)
        os.makedirs(os.path.join(data_dir, split, "scenes"), exist_ok=True)
        for i in range(n):
            fn = "CLEVR_%s_%06d.png" % (split, i)
            _png(os.path.join(data_dir, split, "images", fn), rng.randint(0, 255, (64, 64, 3), dtype=np.uint8))
            objs = []
            for k in range(1 + i % 4):
                x, y = int(rng.randint(0, 30)), int(rng.randint(0, 30))
                objs.append({"bbox": [x, y, int(rng.randint(8, 30)), int(rng.randint(8, 30))],
                             "shape": shapes[int(rng.randint(0, 3))], "color": colors[int(rng.randint(0, 8))]})
            json.dump({"image_filename": fn, "objects": objs},
                      open(os.path.join(data_dir, split, "scenes", "CLEVR_%s_%06d.json" % (split, i)), "w"))
    return data_dir


def build_mnist_tree(root, n=N_ITEMS):
    rng = np.random.RandomState(43)
    data_dir = os.path.join(root, "mnist")
    for split in ("train", "test"):
        d = os.path.join(data_dir, split, "normal")
        os.makedirs(os.path.join(d, "imgs"), exist_ok=True)
        names = ["some/where/%05d.png" % i for i in range(n)]
        for nm in names:
            _png(os.path.join(d, "imgs", nm.split("/")[-1]), rng.randint(0, 255, (64, 64), dtype=np.uint8))
        bbox = np.stack([np.concatenate([rng.uniform(0, 0.6, (3, 2)), rng.uniform(0.15, 0.3, (3, 2))], 1) for _ in range(n)])
        lab = np.eye(10)[rng.randint(0, 10, (n, 3))]
        pickle.dump(names, open(os.path.join(d, "filenames.pickle"), "wb"))
        pickle.dump(bbox.tolist(), open(os.path.join(d, "bboxes.pickle"), "wb"))
        pickle.dump(lab.tolist(), open(os.path.join(d, "labels.pickle"), "wb"))
    return data_dir


def item_probe(img):
    """corner samples + sum of an image tensor (enough to pin origin, flip and content)"""
    img = torch.as_tensor(img).double()
    return [float(img[0, 0, 0]), float(img[0, 0, -1]), float(img[-1, -1, 0]), float(img[-1, -1, -1]), float(img.sum())]
