"""Real-data side of the three StackGAN-family trees on CPU (VERDICT r4 "missing" item 5): the
TextDatasets of mogan_amd.stackgan.datasets against tests/golden/stackgan_data.npz -- outputs of the REFERENCE's own dataset
classes on the tiny trees / seeded cases of tests/stackgan_data_cases.py (tests/golden/make_golden_stackgan_data.py) -- plus the
Torch7 reader of the sampling path."""
import os
import random

import numpy as np
import pytest
import torch

from helpers import load_pkg

load_pkg()
import stackgan_data_cases as C  # noqa: E402
from mogan_amd.stackgan import datasets as D, t7  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stackgan_data.npz"))


def seed(s):
    random.seed(s)
    np.random.seed(s)


def test_crop_imgs_equals_the_reference():
    """S/miscc/datasets.py:100-183 on 40 seeded cases (both stages, absent objects, both clamps, flips): the box sets bit for bit
    (float64 arithmetic in the reference's order), the crop by its corners and sum."""
    flips = 0
    for case in range(C.N_CROP):
        stage, img, b = C.crop_case(case)
        seed(700 + case)
        crop, scaled = D.crop_imgs(img, b, 64 if stage == 1 else 256, stage, 3)
        sets = [scaled, scaled] if stage == 1 else scaled
        assert tuple(crop.shape) == ((3, 64, 64) if stage == 1 else (3, 256, 256)) and sets[0].dtype == np.float64
        np.testing.assert_array_equal(sets[0], GOLD["crop_b1"][case], err_msg="case %d" % case)
        np.testing.assert_array_equal(sets[1], GOLD["crop_b2"][case], err_msg="case %d" % case)
        np.testing.assert_array_equal(np.asarray(C.item_probe(crop)), GOLD["crop_probe"][case])
        flips += int(crop[0, 0, 0] > crop[0, 0, -1])
    assert 5 < flips < C.N_CROP - 5                                   # both branches were exercised


@pytest.mark.parametrize("stage", [1, 2])
def test_coco_text_dataset_items_equal_the_reference(tmp_path, stage):
    data_dir, img_dir, raw = C.build_coco_tree(str(tmp_path))
    resize, imsize = (76, 64) if stage == 1 else (268, 256)
    ds = D.CocoTextDataset(data_dir, img_dir, imsize, split="train", transform=D.image_transform(resize), crop=True, stage=stage)
    assert len(ds) == C.N_ITEMS and ds.max_objects == 3
    for i in range(len(ds)):
        seed(800 + 10 * stage + i)
        img, bbox, label, emb = ds[i]
        assert tuple(img.shape) == (3, imsize, imsize) and img.dtype == torch.float32 and float(img.abs().max()) <= 1.0
        got = np.stack(bbox) if stage == 2 else bbox
        np.testing.assert_array_equal(got, GOLD["coco%d_bbox" % stage][i])
        np.testing.assert_array_equal(label, GOLD["coco%d_label" % stage][i])
        np.testing.assert_array_equal(emb, GOLD["coco%d_emb" % stage][i])
        np.testing.assert_array_equal(np.asarray(C.item_probe(img)), GOLD["coco%d_probe" % stage][i])
    # the default-collated minibatch is what the trainers' loop prologue unpacks (S/trainer.py:153)
    seed(5)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=4, drop_last=True, shuffle=False)))
    assert tuple(batch[0].shape) == (4, 3, imsize, imsize) and tuple(batch[2].shape) == (4, 3, 1) and tuple(batch[3].shape) == (4, 16)
    if stage == 2:
        assert isinstance(batch[1], list) and [tuple(x.shape) for x in batch[1]] == [(4, 3, 4)] * 2
    else:
        assert tuple(batch[1].shape) == (4, 3, 4)
    # crop=False hands the loaded image and the stored boxes through
    ds0 = D.CocoTextDataset(data_dir, img_dir, imsize, split="test", transform=D.image_transform(resize), crop=False, stage=stage)
    img, bbox, _, _ = ds0[1]
    assert tuple(img.shape) == (3, resize, resize)
    np.testing.assert_array_equal(bbox, raw["bbox"][1])


def test_clevr_text_dataset_items_equal_the_reference(tmp_path):
    data_dir = C.build_clevr_tree(str(tmp_path))
    ds = D.ClevrTextDataset(data_dir, 64, split="train", transform=D.image_transform())
    assert len(ds) == C.N_ITEMS and ds.max_objects == 4
    order = sorted(range(len(ds)), key=lambda i: ds.filenames[i])
    for n, i in enumerate(order):
        seed(900 + n)
        img, (tm, tmi), label, bbox = ds[i]
        np.testing.assert_array_equal(np.asarray(C.item_probe(img)), GOLD["clevr_probe"][n])
        np.testing.assert_array_equal(tm.numpy(), GOLD["clevr_mats"][n, 0])
        np.testing.assert_array_equal(tmi.numpy(), GOLD["clevr_mats"][n, 1])
        np.testing.assert_array_equal(label.numpy(), GOLD["clevr_label"][n])
        np.testing.assert_array_equal(bbox, GOLD["clevr_bbox"][n])
        assert label.shape == (4, 13) and float(label[:, :4].sum()) == 4.0 and float(label[:, 4:].sum()) == 4.0
        nobj = 1 + n % 4
        assert torch.all(label[nobj:, 3] == 1) and torch.all(label[nobj:, 12] == 1)          # absent slots: the "empty" classes
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=3, drop_last=True, shuffle=False)))
    assert tuple(batch[0].shape) == (3, 3, 64, 64) and [tuple(x.shape) for x in batch[1]] == [(3, 4, 2, 3)] * 2
    assert tuple(batch[2].shape) == (3, 4, 13) and tuple(batch[3].shape) == (3, 4, 4)


def test_mnist_text_dataset_items_equal_the_reference(tmp_path):
    data_dir = C.build_mnist_tree(str(tmp_path))
    ds = D.MnistTextDataset(data_dir, 64, split="train", transform=D.image_transform(), crop=True)
    assert len(ds) == C.N_ITEMS
    for i in range(len(ds)):
        img, bbox, label = ds[i]
        assert tuple(img.shape) == (1, 64, 64) and bbox.dtype == np.float64
        np.testing.assert_array_equal(np.asarray(C.item_probe(img)), GOLD["mnist_probe"][i])
        np.testing.assert_array_equal(bbox, GOLD["mnist_bbox"][i])
        np.testing.assert_array_equal(label, GOLD["mnist_label"][i])
    lab, bb = D.load_validation_data(os.path.join(data_dir, "test"), tree="mnist")
    assert tuple(lab.shape) == (C.N_ITEMS, 3, 10) and tuple(bb.shape) == (C.N_ITEMS, 3, 4) and bb.dtype == torch.float64


def test_image_transform_is_to_tensor_and_normalize():
    from PIL import Image
    a = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    t = D.image_transform()(Image.fromarray(a))
    np.testing.assert_allclose(t.numpy(), (a.transpose(2, 0, 1).astype(np.float32) / 255.0 - 0.5) / 0.5, rtol=0, atol=1e-7)
    t2 = D.image_transform(8)(Image.fromarray(a))
    assert tuple(t2.shape) == (3, 8, 8)
    g = D.image_transform()(Image.fromarray(a[:, :, 0]))
    assert tuple(g.shape) == (1, 4, 6)


def test_torch7_reader_round_trip(tmp_path):
    """the structure of val_captions.t7 (S/trainer.py:300-302): {raw_txt = {strings}, fea_txt = {FloatTensor (n_i, 1024)}}"""
    rng = np.random.RandomState(0)
    obj = {"raw_txt": ["a man rides a horse", "two dogs / one cat", "ünïcode"],
           "fea_txt": [rng.standard_normal((1, 12)).astype(np.float32) for _ in range(3)],
           "n": 3, "flag": True, "nothing": None, "ids": np.arange(6, dtype=np.int64).reshape(2, 3),
           "d": rng.standard_normal((2, 2))}
    p = str(tmp_path / "val_captions.t7")
    t7.save(p, obj)
    back = t7.load(p)
    assert back.raw_txt == obj["raw_txt"] and back["n"] == 3 and back["flag"] is True and back["nothing"] is None
    assert isinstance(back["fea_txt"], list) and len(back["fea_txt"]) == 3
    for a, b in zip(back["fea_txt"], obj["fea_txt"]):
        assert a.dtype == np.float32
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(back["ids"], obj["ids"])
    np.testing.assert_array_equal(back["d"], obj["d"])
    # a hand-assembled file: a strided (transposed) view over a shared storage, referenced twice
    import struct
    raw = struct.pack("<iii", 3, 1, 2)                                             # table #1, two entries
    raw += struct.pack("<id", 1, 1.0)                                              # key 1
    tens = struct.pack("<ii", 4, 2) + struct.pack("<i", 3) + b"V 1" + struct.pack("<i", 17) + b"torch.FloatTensor"
    tens += struct.pack("<i", 2) + struct.pack("<qq", 3, 2) + struct.pack("<qq", 1, 3) + struct.pack("<q", 1)
    tens += struct.pack("<ii", 4, 3) + struct.pack("<i", 3) + b"V 1" + struct.pack("<i", 18) + b"torch.FloatStorage"
    tens += struct.pack("<q", 6) + np.arange(6, dtype=np.float32).tobytes()
    raw += tens
    raw += struct.pack("<id", 1, 2.0) + struct.pack("<ii", 4, 2)                   # key 2 -> the memoised tensor #2
    open(p, "wb").write(raw)
    back = t7.load(p)
    np.testing.assert_array_equal(back[0], np.arange(6, dtype=np.float32).reshape(2, 3).T)
    assert back[1] is back[0]
    with pytest.raises(EOFError):
        open(p, "wb").write(raw[:40])
        t7.load(p)
