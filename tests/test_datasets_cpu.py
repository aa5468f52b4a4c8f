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


# ---------------------------------------------------------------------------- pinned to the reference (tests/golden/datasets.npz)
def _golden_datasets():
    from helpers import GOLDEN
    return np.load(os.path.join(GOLDEN, "datasets.npz"))


def test_crop_imgs_and_draw_crop_against_the_reference_fixture():
    """datasets.py:95-137 run by the REFERENCE (tests/golden/make_golden_datasets.py) on 48 seeded cases -- crop origin, flip,
    box rescale, both clamps, absent objects: datasets.crop_imgs must reproduce the scaled boxes bit for bit (float64, same
    draws in the same order) and the same crop; feeder.draw_crop (the device feeder's host half, float32 boxes as the raw
    loader carries them) the same crop parameters and the boxes to float32 rounding; the theta matrices built from them
    (datasets.py:331-339) to 1e-6."""
    import datasets_cases as C
    from mogan_amd.attngan import feeder
    g = _golden_datasets()
    img = C.crop_image()
    ds = datasets.SyntheticTextDataset(length=1)
    for case in range(C.N_CROP):
        np.random.seed(500 + case)
        crop, scaled = datasets.crop_imgs(img, C.crop_boxes(case))
        assert scaled.dtype == np.float64
        np.testing.assert_array_equal(scaled, g["crop_boxes"][case], err_msg="case %d" % case)
        probe = [float(crop[0, 0, 0]), float(crop[0, 0, 255]), float(crop[2, 255, 0]), float(crop[1, 255, 255]),
                 float(crop.double().sum())]
        np.testing.assert_array_equal(np.asarray(probe), g["crop_probe"][case], err_msg="crop of case %d" % case)
        tm, tmi = ds._matrices(scaled)
        np.testing.assert_allclose(tm.numpy(), g["crop_mats"][case][0], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(tmi.numpy(), g["crop_mats"][case][1], rtol=1e-5, atol=1e-5)
        # the device feeder's host half: same draws -> same (h1, w1, flip); the crop it asks the kernel for is the reference's
        np.random.seed(500 + case)
        (h1, w1, flip), sb = feeder.draw_crop(C.crop_boxes(case).astype(np.float32), np.random)
        want00 = g["crop_probe"][case][0]                     # = flat index of the crop's first element in channel 0
        col = h1 + 255 if flip else h1
        assert float(img[0, w1, col]) == want00, (case, h1, w1, flip)
        np.testing.assert_allclose(sb, g["crop_boxes"][case], rtol=0, atol=2e-7)
    # the fixture covers what it claims: both clamps, flips, absent objects
    b = g["crop_boxes"]
    assert np.any(np.isclose(b[:, :, 0] + b[:, :, 2], 0.999)) and np.any(np.isclose(b[:, :, 1] + b[:, :, 3], 0.999))
    assert np.any(b[:, 2, 0] == -1) and np.any(b[:, 1, 0] == -1) and np.any(b[:, 0, 0] != -1)


def test_one_hot_labels_and_captions_against_the_reference_fixture():
    """datasets.py:341-349 (get_one_hot_labels) and 311-329 (get_caption, incl. the seeded random subset of a caption longer
    than WORDS_NUM) as the reference computes them."""
    import datasets_cases as C
    g = _golden_datasets()
    for i, lab in enumerate(C.label_cases()):
        got = datasets._Base._one_hot(lab).numpy()
        np.testing.assert_array_equal(got, g["onehot"][i])
        np.testing.assert_array_equal(synthetic.one_hot_labels(lab.reshape(-1)).numpy(), g["onehot"][i])
    cfg.TEXT.WORDS_NUM = C.T_WORDS
    holder = datasets.TextDataset.__new__(datasets.TextDataset)
    for i, cap in enumerate(C.caption_cases()):
        holder.captions = [cap]
        np.random.seed(900 + i)
        x, n = holder.get_caption(0)
        np.testing.assert_array_equal(x, g["cap_x"][i])
        assert int(n) == int(g["cap_len"][i])


@pytest.mark.parametrize("ev", [False, True])
def test_prepare_data_against_the_reference_fixture(ev):
    """datasets.py:28-68: every field of the minibatch re-ordered by descending caption length exactly as the reference does
    (ties included), train and eval form."""
    import datasets_cases as C
    g = _golden_datasets()
    tag = "pde" if ev else "pd"
    imgs, caps, lens, cls, keys, tms, label, bbox = C.batch_case()
    data = [list(imgs), caps, lens, cls, keys, list(tms), label] + ([bbox] if ev else [])
    res = datasets.prepare_data(data, torch.device("cpu"), eval=ev)
    for i, im in enumerate(res[0]):
        np.testing.assert_array_equal(im.numpy(), g["%s_img%d" % (tag, i)])
    np.testing.assert_array_equal(res[1].numpy(), g[tag + "_captions"])
    np.testing.assert_array_equal(res[2].numpy(), g[tag + "_lens"])
    np.testing.assert_array_equal(np.asarray(res[3]), g[tag + "_class_ids"])
    assert list(res[4]) == list(g[tag + "_keys"])
    np.testing.assert_array_equal(res[5][0].numpy(), g[tag + "_tm"])
    np.testing.assert_array_equal(res[5][1].numpy(), g[tag + "_tmi"])
    np.testing.assert_array_equal(res[6].numpy(), g[tag + "_label"])
    if ev:
        np.testing.assert_array_equal(res[7].numpy(), g[tag + "_bbox"])
