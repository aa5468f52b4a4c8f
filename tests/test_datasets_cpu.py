"""Data side of the coco-attngan train step on CPU: TextDataset reads the
reference's file formats (captions.pickle, <split>/{filenames,bboxes,labels}.pickle, JPEGs) from a tiny fake COCO
tree built here, and prepare_data emits the tuple the reference train loop unpacks (datasets.py:28-68)."""
import os
import pickle

import numpy as np
import pytest
import torch

from helpers import load_pkg

load_pkg()
from mogan_amd.attngan import datasets, synthetic  # noqa: E402
from mogan_amd.attngan.miscc.config import cfg  # noqa: E402


@pytest.fixture
def fake_coco(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(0)
    n, cpi = 6, 5
    data_dir, img_dir = tmp_path / "coco", tmp_path / "coco" / "images"
    (data_dir / "train").mkdir(parents=True)
    img_dir.mkdir()
    names = ["COCO_train2014_%012d" % i for i in range(n)]
    for nm in names:
        Image.fromarray(rng.randint(0, 255, (90, 120, 3), dtype=np.uint8)).save(str(img_dir / (nm + ".jpg")))
    bbox = np.full((n, 3, 4), -1.0, np.float32)
    labels = np.full((n, 3, 1), -1.0, np.float32)
    for i in range(n):
        for k in range(2 + (i % 2)):
            bbox[i, k] = [rng.uniform(0, 0.6), rng.uniform(0, 0.6), rng.uniform(0.2, 0.5), rng.uniform(0.2, 0.5)]
            labels[i, k] = rng.randint(0, 80)
    caps = [list(rng.randint(1, 50, rng.randint(4, 20))) for _ in range(n * cpi)]
    ixtoword = {i: "w%d" % i for i in range(50)}
    wordtoix = {v: k for k, v in ixtoword.items()}
    pickle.dump([caps, caps, ixtoword, wordtoix], open(str(data_dir / "captions.pickle"), "wb"))
    pickle.dump(names, open(str(data_dir / "train" / "filenames.pickle"), "wb"))
    pickle.dump(bbox.tolist(), open(str(data_dir / "train" / "bboxes.pickle"), "wb"))
    pickle.dump(labels.tolist(), open(str(data_dir / "train" / "labels.pickle"), "wb"))
    return str(data_dir), str(img_dir), n


def test_text_dataset_and_prepare_data(fake_coco):
    data_dir, img_dir, n = fake_coco
    cfg.TREE.BRANCH_NUM, cfg.TEXT.WORDS_NUM, cfg.TEXT.CAPTIONS_PER_IMAGE = 3, 12, 5
    ds = datasets.TextDataset(data_dir, img_dir, split="train", base_size=64)
    assert len(ds) == n and ds.n_words == 50
    np.random.seed(1)
    dl = torch.utils.data.DataLoader(ds, batch_size=4, drop_last=True, shuffle=False)
    batch = next(iter(dl))
    imgs, captions, cap_lens, class_ids, keys, (tm, tmi), onehot = datasets.prepare_data(batch, torch.device("cpu"))
    assert [tuple(im.shape) for im in imgs] == [(4, 3, 64, 64), (4, 3, 128, 128), (4, 3, 256, 256)]
    assert all(im.dtype == torch.float32 and im.min() >= -1 and im.max() <= 1 for im in imgs)
    assert captions.shape == (4, 12) and captions.dtype == torch.int64
    assert torch.all(cap_lens[:-1] >= cap_lens[1:])                     # sorted by caption length, descending
    for b in range(4):                                                   # zero padding beyond the length
        assert torch.all(captions[b, int(cap_lens[b]):] == 0) and torch.all(captions[b, :int(cap_lens[b])] > 0)
    assert tm.shape == (4, 3, 2, 3) and tmi.shape == (4, 3, 2, 3) and tm.dtype == torch.float32
    assert onehot.shape == (4, 3, 81) and torch.all(onehot.sum(-1) == 1)
    assert len(keys) == 4 and isinstance(class_ids, np.ndarray)
    # absent third object -> label class 80 and theta built from bbox = -1 (miscc/utils.py:16-49)
    absent = onehot[:, 2, 80] == 1
    assert absent.any()
    np.testing.assert_array_equal(tmi[absent][:, 2].numpy(), np.tile(np.array([[-1, 0, -4], [0, -1, -4]], np.float32),
                                                                    (int(absent.sum()), 1, 1)))


def test_crop_imgs_rules():
    """datasets.py:95-137: 268 -> 256 crop, flip mirrors x, the clamp keeps x+w <= 0.999."""
    class R:
        def __init__(self, vals): self.vals = list(vals)
        def random(self): return self.vals.pop(0)
    img = torch.arange(3 * 268 * 268, dtype=torch.float32).view(3, 268, 268)
    bbox = np.array([[0.5, 0.25, 0.6, 0.5], [0.1, 0.1, 0.2, 0.2], [-1, -1, -1, -1]], np.float32)
    out_img, out = datasets.crop_imgs(img, bbox, rng=R([0.9, 0.5, 0.25]))     # no flip, h1 = 6, w1 = 3
    assert out_img.shape == (3, 256, 256) and out_img[0, 0, 0] == img[0, 3, 6]
    x = (0.5 * 268 - 6) / 256
    np.testing.assert_allclose(out[0], [x, (0.25 * 268 - 3) / 256, 1.0 - x - 0.001, 0.5 * 268 / 256], rtol=1e-6)
    assert np.all(out[2] == -1)
    _, outf = datasets.crop_imgs(img, bbox, rng=R([0.1, 0.5, 0.25]))          # flip
    np.testing.assert_allclose(outf[1][0], 1.0 - out[1][0] - out[1][2], rtol=1e-6)


def test_synthetic_dataset_matches_the_same_contract():
    cfg.TREE.BRANCH_NUM, cfg.TEXT.WORDS_NUM = 3, 12
    dl = torch.utils.data.DataLoader(datasets.SyntheticTextDataset(length=8), batch_size=4)
    out = datasets.prepare_data(next(iter(dl)), torch.device("cpu"))
    assert len(out) == 7 and out[1].shape == (4, 12) and out[6].shape == (4, 3, 81)
    bt = synthetic.make_batch(4)
    assert [tuple(t.shape) for t in out[0]] == [tuple(t.shape) for t in bt["imgs"]]


def test_text_dataset_raw_mode_structure(fake_coco):
    """TextDataset(raw=True): what the device feeder consumes -- the decoded 268x268 u8 image, the UNSCALED boxes, the same
    caption / label fields as the host-augmenting mode."""
    data_dir, img_dir, n = fake_coco
    cfg.TREE.BRANCH_NUM, cfg.TEXT.WORDS_NUM, cfg.TEXT.CAPTIONS_PER_IMAGE = 3, 12, 5
    ds = datasets.TextDataset(data_dir, img_dir, split="train", base_size=64, raw=True)
    dl = torch.utils.data.DataLoader(ds, batch_size=3, drop_last=True, shuffle=False)
    u8, caps, lens, cls, keys, bbox, onehot = next(iter(dl))
    assert u8.dtype == torch.uint8 and tuple(u8.shape) == (3, 268, 268, 3)
    assert tuple(caps.shape) == (3, 12, 1) and tuple(bbox.shape) == (3, 3, 4) and tuple(onehot.shape) == (3, 3, 81)
    np.testing.assert_allclose(bbox.numpy(), np.asarray(ds.bbox[:3], dtype=np.float32))
