"""Synthetic minibatches for the StackGAN-family trees, shaped like what each reference data loader +
trainer prologue hands to the step:

  coco stage I   S/miscc/datasets.py:100-131,185-216 + S/trainer.py:156-186: img (B,3,64,64), one bbox set
                 scaled 76->64 with the clamp x+w<=0.999, labels -1 -> class 80 one-hot(81), (B,1024) char-CNN-RNN
                 embedding
  coco stage II  S/miscc/datasets.py:132-170: img (B,3,256,256), TWO bbox sets from the same crop offsets
                 (76->64 for the frozen stage-I generator, 268->256 for stage II)
  clevr          C/miscc/datasets.py:114-145: img (B,3,64,64), 4 slots, absent slot = bbox -1/imsize and the
                 "absent" classes (shape 3, colour 8) in the 4+9 one-hot
  multi-mnist    M/miscc/datasets.py:73-88 + M/trainer.py:122-129: img (B,1,64,64), 3 digits, float64 bboxes ->
                 matrices in float64 -> .float()

numpy-seeded so the CPU oracle and the GPU path see bit-identical inputs.
"""
import numpy as np
import torch

from ..attngan.synthetic import bbox_to_theta, one_hot_labels

TREES = ("coco", "clevr", "mnist")


def _theta64(bbox):
    """multi-mnist computes the matrices in float64 (M/miscc/utils.py:28,46) and casts to float32."""
    b = torch.as_tensor(bbox, dtype=torch.float64).view(-1, 4)
    x, y, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    z = torch.zeros_like(x)
    tm = torch.stack([w, z, 2 * ((x + 0.5 * w) - 0.5), z, h, 2 * ((y + 0.5 * h) - 0.5)], 1)
    sx, sy = 1.0 / w, 1.0 / h
    tmi = torch.stack([sx, z, 2 * sx * (0.5 - (x + 0.5 * w)), z, sy, 2 * sy * (0.5 - (y + 0.5 * h))], 1)
    return tm.view(-1, 2, 3).float(), tmi.view(-1, 2, 3).float()


def _rescale(b, ori, size, h1, w1):
    """one bbox through the crop of S/miscc/datasets.py:116-128 (no flip)."""
    x = max(b[0] * float(ori) - h1, 0) / float(size)
    y = max(b[1] * float(ori) - w1, 0) / float(size)
    w = min((float(ori) / size) * b[2], 1.0)
    if x + w > 0.999:
        w = 1.0 - x - 0.001
    h = min((float(ori) / size) * b[3], 1.0)
    if y + h > 0.999:
        h = 1.0 - y - 0.001
    return (x, y, w, h)


def make_batch(tree, batch, stage=1, seed=0, z_dim=100, cond_dim=128, text_dim=1024, p_absent=0.5):
    assert tree in TREES
    rng = np.random.RandomState(seed)
    K = 4 if tree == "clevr" else 3
    size = 256 if stage == 2 else 64
    ch = 1 if tree == "mnist" else 3
    out = {"real_imgs": torch.from_numpy(rng.uniform(-1, 1, (batch, ch, size, size)).astype(np.float32))}
    out["z"] = torch.from_numpy(rng.standard_normal((batch, z_dim)).astype(np.float32))
    raw = np.zeros((batch, K, 4))
    present = np.ones((batch, K), dtype=bool)
    for b in range(batch):
        if tree != "mnist" and rng.random_sample() < p_absent:
            present[b, K - 1] = False
        for k in range(K):
            x, y = rng.uniform(0.0, 0.5, 2)
            w, h = rng.uniform(0.1, 0.5, 2)
            raw[b, k] = (x, y, w, h)
    if tree == "coco":
        labels = np.where(present, rng.randint(0, 80, (batch, K)), -1)
        out["label_one_hot"] = one_hot_labels(labels)
        sets = [np.full((batch, K, 4), -1.0, np.float32) for _ in range(2)]
        for b in range(batch):
            crop = 268 - 256 if stage == 2 else 76 - 64
            h1, w1 = int(np.floor(crop * rng.random_sample())), int(np.floor(crop * rng.random_sample()))
            for k in range(K):
                if not present[b, k]:
                    break
                sets[0][b, k] = _rescale(raw[b, k], 76, 64, h1, w1)
                sets[1][b, k] = _rescale(raw[b, k], 268, 256, h1, w1)
        out["bbox"] = torch.from_numpy(sets[0])
        tm, tmi = bbox_to_theta(sets[0].reshape(-1, 4))
        out["tm"], out["tmi"] = tm.view(batch, K, 2, 3), tmi.view(batch, K, 2, 3)
        if stage == 2:
            out["bbox_s2"] = torch.from_numpy(sets[1])
            tm2, tmi2 = bbox_to_theta(sets[1].reshape(-1, 4))
            out["tm_s2"], out["tmi_s2"] = tm2.view(batch, K, 2, 3), tmi2.view(batch, K, 2, 3)
            out["eps_s1"] = torch.from_numpy(rng.standard_normal((batch, cond_dim)).astype(np.float32))
        out["txt_embedding"] = torch.from_numpy(rng.standard_normal((batch, text_dim)).astype(np.float32))
        out["eps"] = torch.from_numpy(rng.standard_normal((batch, cond_dim)).astype(np.float32))
    elif tree == "clevr":
        bbox = np.where(present[..., None], raw, -1.0 / 64.0).astype(np.float32)
        shape = np.where(present, rng.randint(0, 3, (batch, K)), 3)
        colour = np.where(present, rng.randint(0, 8, (batch, K)), 8)
        lab = np.zeros((batch, K, 13), np.float32)
        for b in range(batch):
            for k in range(K):
                lab[b, k, shape[b, k]] = 1.0
                lab[b, k, 4 + colour[b, k]] = 1.0
        out["label_one_hot"] = torch.from_numpy(lab)
        out["bbox"] = torch.from_numpy(bbox)
        tm, tmi = bbox_to_theta(bbox.reshape(-1, 4))
        out["tm"], out["tmi"] = tm.view(batch, K, 2, 3), tmi.view(batch, K, 2, 3)
    else:
        digits = rng.randint(0, 10, (batch, K))
        lab = np.zeros((batch, K, 10), np.float32)
        for b in range(batch):
            for k in range(K):
                lab[b, k, digits[b, k]] = 1.0
        out["label_one_hot"] = torch.from_numpy(lab)
        out["bbox"] = torch.from_numpy(raw)                       # float64, like the pickled boxes
        tm, tmi = _theta64(raw.reshape(-1, 4))
        out["tm"], out["tmi"] = tm.view(batch, K, 2, 3), tmi.view(batch, K, 2, 3)
    return out
