"""Attention-map grids: mogan_amd.attngan.miscc.vis against the fixture captured from the
reference's build_super_images / build_super_images2 (tests/golden/make_golden_vis.py).  Host side only: PIL + numpy."""
import os

import numpy as np
import pytest
import torch
from PIL import ImageFont

import mogan_loader
mogan_loader.load()
from mogan_amd.attngan.miscc import vis          # noqa: E402
from vis_cases import vis_inputs                 # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vis.npz")


@pytest.fixture()
def builtin_font(monkeypatch):
    # the fixture was drawn with PIL's built-in font (FreeMono.ttf is not in this image)
    monkeypatch.setattr(vis, "_font", lambda: ImageFont.load_default())


def test_pyramid_expand_properties():
    x = np.full((5, 7, 3), 0.25)
    y = vis.pyramid_expand(x, upscale=4, sigma=20)
    assert y.shape == (20, 28, 3)
    np.testing.assert_allclose(y, 0.25, atol=1e-12)                   # smoothing with 'reflect' keeps constants
    r = np.random.RandomState(0).rand(6, 6, 3)
    y = vis.pyramid_expand(r, upscale=2)
    np.testing.assert_allclose(y.mean((0, 1)), r.mean((0, 1)), rtol=0.05)
    assert np.all(y[..., 0] != y[..., 1])                              # the channel axis is not mixed
    assert y.min() >= r.min() - 1e-12 and y.max() <= r.max() + 1e-12


def test_build_super_images_matches_reference(builtin_font):
    g = np.load(GOLD)
    c = vis_inputs()
    sup, sent = vis.build_super_images(c["img"], c["captions"], c["ixtoword"], c["attn"], c["att_sze"],
                                       batch_size=c["B"], max_word_num=c["T"])
    assert sup.dtype == np.uint8 and sup.shape == (8 * (vis.FONT_MAX + 2 * 32), (c["T"] + 2) * 34, 3)
    np.testing.assert_array_equal(sup, g["sup"])
    assert [" ".join(s) for s in sent] == list(g["sentences"])
    sup_lr, _ = vis.build_super_images(c["img"], c["captions"], c["ixtoword"], c["attn"], c["att_sze"],
                                       lr_imgs=c["lr"], batch_size=c["B"], max_word_num=c["T"])
    np.testing.assert_array_equal(sup_lr, g["sup_lr"])
    assert not np.array_equal(sup, sup_lr)


def test_build_super_images2_matches_reference(builtin_font):
    g = np.load(GOLD)
    c = vis_inputs()
    sup2, _ = vis.build_super_images2(c["img"][:3], c["captions"][:3], c["cap_lens"][:3], c["ixtoword"],
                                      [a[:n] for a, n in zip(c["attn"][:3], c["cap_lens"][:3])], c["att_sze"],
                                      vis_size=32, topK=2)
    np.testing.assert_array_equal(sup2, g["sup2"])


def test_inputs_not_modified_and_width_mismatch():
    c = vis_inputs()
    before = c["img"].clone()
    vis.build_super_images(c["img"], c["captions"], c["ixtoword"], c["attn"], c["att_sze"], batch_size=c["B"],
                           max_word_num=c["T"])
    assert torch.equal(before, c["img"])
