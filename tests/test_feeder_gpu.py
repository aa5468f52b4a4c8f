"""-m gpu: the device feeder (csrc/mogan_feed.hip through attngan/feeder.DeviceFeeder) against the host pipeline of
datasets.py (ToTensor -> crop/flip -> ToPILImage + PIL bilinear resize -> Normalize; code/coco/attngan/datasets.py:70-137):
byte work, so every image of every scale must be BIT-identical; plus prepare_data_raw's output contract."""
import numpy as np
import pytest
import torch

from helpers import load_pkg

load_pkg()
from mogan_amd.attngan import datasets, feeder  # noqa: E402

pytestmark = pytest.mark.gpu


def _host_reference(u8, p):
    h1, w1, flip = p
    t = torch.from_numpy(u8.astype(np.float32) / 255.0).permute(2, 0, 1)
    img = t[:, w1:w1 + 256, h1:h1 + 256]
    if flip:
        img = torch.flip(img, dims=[2])
    return datasets._multi_scale(img, [64, 128, 256])


def test_feeder_bit_identical_to_host_pipeline():
    rng = np.random.RandomState(3)
    B = 6
    u8 = rng.randint(0, 256, (B, 268, 268, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:268, 0:268]
    u8[1] = np.stack([(xx + yy) // 3, 255 - xx * 255 // 267, (yy * 5) % 256], -1).astype(np.uint8)      # smooth ramps
    params = [(0, 0, 0), (12, 12, 1), (5, 11, 1), (11, 0, 0), (3, 7, 0), (12, 0, 1)]
    fd = feeder.DeviceFeeder("cuda", batch=8)
    for rep in range(3):                                    # the staging slots are reused
        outs = fd(torch.from_numpy(u8), np.asarray(params, dtype=np.int32))
        torch.cuda.synchronize()
        for b in range(B):
            want = _host_reference(u8[b], params[b])
            for got, w, s in zip(outs, want, (64, 128, 256)):
                assert tuple(got.shape) == (B, 3, s, s)
                assert torch.equal(got[b].cpu(), w), "sample %d, %dx%d: max diff %.3g" % (
                    b, s, s, float((got[b].cpu() - w).abs().max()))


def test_prepare_data_raw_contract():
    rng = np.random.RandomState(5)
    B = 4
    u8 = torch.from_numpy(rng.randint(0, 256, (B, 268, 268, 3)).astype(np.uint8))
    caps = torch.from_numpy(rng.randint(1, 50, (B, 12, 1)).astype(np.int64))
    lens = torch.tensor([7, 12, 9, 12])
    bbox = torch.tensor([[[0.1, 0.1, 0.3, 0.4], [0.5, 0.5, 0.45, 0.49], [-1, -1, -1, -1]]] * B)
    label = torch.zeros(B, 3, 81)
    fd = feeder.DeviceFeeder("cuda", batch=B)
    out = datasets.prepare_data_raw((u8, caps, lens, torch.arange(B), ["k%d" % i for i in range(B)], bbox, label), fd,
                                    rng=np.random.RandomState(1))
    imgs, captions, cap_lens, class_ids, keys, (tm, tmi), lab = out
    assert [tuple(i.shape) for i in imgs] == [(B, 3, 64, 64), (B, 3, 128, 128), (B, 3, 256, 256)]
    assert cap_lens.tolist() == sorted(lens.tolist(), reverse=True) and tuple(captions.shape) == (B, 12)
    assert tuple(tm.shape) == (B, 3, 2, 3) and tuple(tmi.shape) == (B, 3, 2, 3) and all(i.is_cuda for i in imgs)
    assert float(imgs[2].min()) >= -1.0 and float(imgs[2].max()) <= 1.0
    assert sorted(keys) == ["k0", "k1", "k2", "k3"] and list(class_ids) != []
