"""Deterministic inputs of the data-contract fixture (tests/golden/datasets.npz): shared by the generator
(tests/golden/make_golden_datasets.py, imports the reference) and the tests (no reference import)."""
import numpy as np
import torch

N_CROP = 48        # (seed, box set) cases of crop_imgs
T_WORDS = 12


def crop_image():
    """(3, 268, 268) float32 image whose every element is distinct and exactly representable: element value = its flat index,
    so a crop's origin / flip can be read off four corner samples."""
    return torch.arange(3 * 268 * 268, dtype=torch.float32).view(3, 268, 268)


def crop_boxes(case):
    """(3, 4) float64 relative boxes (x, y, w, h) as the reference's loader yields them (np.array of the pickled lists,
    datasets.py:171-177); every third case has an absent last object, every seventh an absent second one as well, and a few
    boxes reach past the right / bottom edge so that both clamps fire."""
    rng = np.random.RandomState(1000 + case)
    b = np.full((3, 4), -1.0, dtype=np.float64)
    n = 3 - (1 if case % 3 == 0 else 0) - (1 if case % 7 == 0 else 0)
    for k in range(n):
        x, y = rng.uniform(0.0, 0.75), rng.uniform(0.0, 0.75)
        w, h = rng.uniform(0.05, 0.6), rng.uniform(0.05, 0.6)
        if case % 5 == 0 and k == 0:
            x, w = 0.7, 0.45                     # x + w > 1 after the 268/256 rescale: width clamp
        if case % 4 == 0 and k == 1:
            y, h = 0.8, 0.9                      # height clamp
        b[k] = [x, y, w, h]
    return b


def label_cases():
    """(n, 3, 1) float64 COCO class ids with -1 for absent objects (labels.pickle layout, datasets.py:179-186)."""
    rng = np.random.RandomState(7)
    lab = rng.randint(0, 80, size=(10, 3, 1)).astype(np.float64)
    lab[1, 2] = -1
    lab[4, 1:] = -1
    lab[7] = -1
    lab[9, 0] = 79
    return lab


def caption_cases():
    """caption index lists: shorter than, equal to and longer than WORDS_NUM (the long ones draw a random subset)"""
    rng = np.random.RandomState(11)
    return [list(rng.randint(1, 500, n)) for n in (3, 12, 13, 20, 37, 1, 12, 25)]


def batch_case():
    """One collated minibatch as torch's DataLoader hands it to prepare_data (datasets.py:28-68): B = 6, caption lengths with
    ties, small images (the function only indexes them)."""
    rng = np.random.RandomState(21)
    B = 6
    imgs = [torch.from_numpy(rng.uniform(-1, 1, (B, 3, s, s)).astype(np.float32)) for s in (4, 8, 16)]
    lens = torch.tensor([5, 12, 7, 12, 3, 7], dtype=torch.int64)
    caps = torch.zeros(B, T_WORDS, 1, dtype=torch.int64)
    for b in range(B):
        caps[b, :int(lens[b]), 0] = torch.from_numpy(rng.randint(1, 500, int(lens[b])))
    class_ids = torch.arange(100, 100 + B)
    keys = ["COCO_%03d" % i for i in range(B)]
    tm = torch.from_numpy(rng.standard_normal((B, 3, 2, 3)).astype(np.float32))
    tmi = torch.from_numpy(rng.standard_normal((B, 3, 2, 3)).astype(np.float32))
    label = torch.from_numpy((rng.uniform(size=(B, 3, 81)) > 0.9).astype(np.float32))
    bbox = torch.from_numpy(rng.uniform(0, 1, (B, 3, 4)))
    return imgs, caps, lens, class_ids, keys, [tm, tmi], label, bbox
