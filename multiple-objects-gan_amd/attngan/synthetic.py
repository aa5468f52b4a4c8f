"""Synthetic coco-attngan minibatches.

Emits what the reference's `prepare_data` hands to the train loop
(code/coco/attngan/datasets.py:28-68): imgs[3] f32 in [-1,1], captions int64 zero-padded,
cap_lens int64 sorted descending, class_ids numpy int, [tm, tmi] f32 (B,3,2,3),
one-hot labels f32 (B,3,81) -- with the reference's bbox clamp rule
(datasets.py:115-121: x+w > 0.999 -> w = 1-x-0.001) and label -1 -> class 80
(datasets.py:341-349).  bbox -> theta follows miscc/utils.py:16-49.

numpy-seeded so that the CPU oracle and the GPU path see bit-identical inputs.
"""
import numpy as np
import torch

MAX_OBJECTS = 3
N_CLASSES = 81
VOCAB = 27297


def bbox_to_theta(bbox):
    """bbox (N,4) = (x,y,w,h) in [0,1] (or -1 = absent) -> (theta, theta_inv), each (N,2,3).
    theta     = [[w,0,2(x+w/2)-1],[0,h,2(y+h/2)-1]]            (miscc/utils.py:34-49)
    theta_inv = [[1/w,0,(2/w)(0.5-(x+w/2))],[0,1/h,(2/h)(0.5-(y+h/2))]]  (miscc/utils.py:16-31)
    Host-side float32 arithmetic in the same operation order as the reference."""
    bbox = torch.as_tensor(bbox, dtype=torch.float32).view(-1, 4)
    x, y, w, h = bbox[:, 0], bbox[:, 1], bbox[:, 2], bbox[:, 3]
    zeros = torch.zeros_like(x)
    tx = 2 * ((x + 0.5 * w) - 0.5)
    ty = 2 * ((y + 0.5 * h) - 0.5)
    theta = torch.stack([w, zeros, tx, zeros, h, ty], 1).view(-1, 2, 3)
    sx, sy = 1.0 / w, 1.0 / h
    itx = 2 * sx * (0.5 - (x + 0.5 * w))
    ity = 2 * sy * (0.5 - (y + 0.5 * h))
    theta_inv = torch.stack([sx, zeros, itx, zeros, sy, ity], 1).view(-1, 2, 3)
    return theta, theta_inv


def make_bboxes(rng, batch, p_absent=0.5):
    bbox = np.full((batch, MAX_OBJECTS, 4), -1.0, dtype=np.float32)
    labels = np.full((batch, MAX_OBJECTS), -1, dtype=np.int64)
    for b in range(batch):
        nobj = MAX_OBJECTS - (1 if rng.random_sample() < p_absent else 0)
        for k in range(nobj):
            x, y = rng.uniform(0.0, 0.5, 2)
            w, h = rng.uniform(0.1, 0.5, 2)
            if x + w > 0.999:
                w = 1.0 - x - 0.001
            if y + h > 0.999:
                h = 1.0 - y - 0.001
            bbox[b, k] = (x, y, w, h)
            labels[b, k] = rng.randint(0, N_CLASSES - 1)
    return bbox, labels


def one_hot_labels(labels):
    lab = torch.as_tensor(labels).long().clone()
    lab[lab < 0] = N_CLASSES - 1
    out = torch.zeros(lab.shape + (N_CLASSES,), dtype=torch.float32)
    out.scatter_(-1, lab.unsqueeze(-1), 1.0)
    return out


def make_batch(batch, words_num=12, nef=256, z_dim=100, cond_dim=100, seed=0,
               branch_num=3, base_size=64, text="gauss"):
    """One synthetic minibatch (CPU tensors). text='gauss': words_embs/sent_emb ~ N(0,1)
    (kernel-only runs); text='tokens': only captions are produced and the caller runs
    RNN_ENCODER on them."""
    rng = np.random.RandomState(seed)
    out = {}
    out["imgs"] = [torch.from_numpy(rng.uniform(-1, 1, (batch, 3, base_size << i, base_size << i))
                                    .astype(np.float32)) for i in range(branch_num)]
    low = min(5, max(1, words_num // 2))
    cap_lens = np.sort(rng.randint(low, words_num + 1, batch))[::-1].copy()
    cap_lens[0] = words_num
    captions = rng.randint(1, VOCAB, (batch, words_num)).astype(np.int64)
    for b in range(batch):
        captions[b, cap_lens[b]:] = 0
    out["captions"] = torch.from_numpy(captions)
    out["cap_lens"] = torch.from_numpy(cap_lens.astype(np.int64))
    out["mask"] = out["captions"] == 0
    out["class_ids"] = np.arange(batch)
    bbox, labels = make_bboxes(rng, batch)
    tm, tmi = bbox_to_theta(bbox.reshape(-1, 4))
    out["bbox"] = torch.from_numpy(bbox)
    out["tm"] = tm.view(batch, MAX_OBJECTS, 2, 3)
    out["tmi"] = tmi.view(batch, MAX_OBJECTS, 2, 3)
    out["label_one_hot"] = one_hot_labels(labels)
    out["z"] = torch.from_numpy(rng.standard_normal((batch, z_dim)).astype(np.float32))
    out["eps"] = torch.from_numpy(rng.standard_normal((batch, cond_dim)).astype(np.float32))
    if text == "gauss":
        out["words_embs"] = torch.from_numpy(
            rng.standard_normal((batch, nef, words_num)).astype(np.float32))
        out["sent_emb"] = torch.from_numpy(rng.standard_normal((batch, nef)).astype(np.float32))
    return out


def to_device(batch, device):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v.to(device)
        elif isinstance(v, list):
            out[k] = [t.to(device) for t in v]
        else:
            out[k] = v
    return out
