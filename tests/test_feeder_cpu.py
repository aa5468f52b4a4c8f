"""CPU: the device feeder's host side.  (1) The resampling coefficient tables + the two-pass fixed-point algorithm restated in
attngan/feeder.py reproduce Pillow's own Image.resize(BILINEAR) bit for bit (that algorithm is what the HIP kernels run);
(2) draw_crop makes the same draws / box arithmetic as datasets.crop_imgs (code/coco/attngan/datasets.py:95-137)."""
import numpy as np
import pytest
import torch
from PIL import Image

from helpers import load_pkg

load_pkg()
from mogan_amd.attngan import datasets, feeder  # noqa: E402


@pytest.mark.parametrize("out_size", [64, 128, 100, 255])
def test_resample_tables_reproduce_pillow_bit_for_bit(out_size):
    rng = np.random.RandomState(out_size)
    for kind in ("noise", "smooth"):
        img = rng.randint(0, 256, (256, 256, 3)).astype(np.uint8)
        if kind == "smooth":
            yy, xx = np.mgrid[0:256, 0:256]
            img = np.stack([(xx + yy) // 2, 255 - xx, (yy * 3) % 256], -1).astype(np.uint8)
        want = np.asarray(Image.fromarray(img).resize((out_size, out_size), Image.BILINEAR))
        got = feeder.resample_u8_reference(img, out_size)
        assert np.array_equal(got, want), "max diff %d" % int(np.abs(got.astype(int) - want.astype(int)).max())


def test_coefficient_tables_shape_and_sum():
    for s in (64, 128):
        bounds, kk = feeder.pil_bilinear_coeffs(256, s)
        assert bounds.shape == (s, 2) and kk.shape[0] == s
        assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= 256).all()
        sums = kk.sum(1)
        assert np.abs(sums - (1 << feeder.PRECISION_BITS)).max() <= kk.shape[1]      # rounding of each tap


def test_draw_crop_equals_crop_imgs():
    bbox = np.array([[0.1, 0.2, 0.5, 0.6], [0.6, 0.55, 0.39, 0.44], [-1, -1, -1, -1]], dtype=np.float32)
    img = torch.rand(3, 268, 268)
    for seed in range(20):
        r1, r2 = np.random.RandomState(seed), np.random.RandomState(seed)
        crop, want = datasets.crop_imgs(img, bbox, rng=r1)
        (h1, w1, flip), got = feeder.draw_crop(bbox, r2)
        np.testing.assert_array_equal(got, want)
        ref = img[:, w1:w1 + 256, h1:h1 + 256]
        if flip:
            ref = torch.flip(ref, dims=[2])
        assert torch.equal(crop, ref)
