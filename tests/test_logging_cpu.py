"""Host-side logging helpers of the StackGAN-family trainers."""
import json

import numpy as np
import torch

import mogan_loader
mogan_loader.load()
from mogan_amd.stackgan import logging_utils as LU      # noqa: E402


def test_make_grid_layout_and_normalisation():
    t = torch.arange(5 * 3 * 4 * 6, dtype=torch.float32).view(5, 3, 4, 6) - 100.0
    g = LU.make_grid(t, nrow=3, padding=2, normalize=True)
    assert g.shape == (3, 2 * (4 + 2) + 2, 3 * (6 + 2) + 2)
    lo, hi = float(t.min()), float(t.max())
    np.testing.assert_allclose(g[:, 2:6, 2:8].numpy(), ((t[0] - lo) / (hi - lo + 1e-5)).numpy(), rtol=1e-6)
    np.testing.assert_allclose(g[:, 8:12, 10:16].numpy(), ((t[4] - lo) / (hi - lo + 1e-5)).numpy(), rtol=1e-6)   # row 1, col 1
    assert float(g[:, :2].abs().max()) == 0 and float(g[:, 8:12, 18:].abs().max()) == 0     # padding, empty cell
    assert LU.make_grid(torch.rand(2, 1, 4, 4)).shape[0] == 3                                # grey -> RGB


def test_save_img_results_and_scalars(tmp_path):
    from PIL import Image
    real, fake = torch.rand(10, 3, 8, 8) * 2 - 1, torch.rand(10, 3, 8, 8) * 2 - 1
    paths = LU.save_img_results(real, fake, 7, str(tmp_path), vis_count=9)
    assert [p.split("/")[-1] for p in paths] == ["real_samples.png", "fake_samples_epoch_007.png"]
    assert Image.open(paths[1]).size == (8 * 10 + 2, 2 * 10 + 2)
    assert LU.save_img_results(None, fake, 1, str(tmp_path))[0].endswith("lr_fake_samples_epoch_001.png")
    w = LU.ScalarWriter(str(tmp_path))
    w.add_scalar("D_loss", torch.tensor(1.5), 3)
    w.add_scalar("G_loss", 0.25, 3)
    w.close()
    rows = [json.loads(l) for l in open(str(tmp_path / "scalars.jsonl"))]
    assert rows == [{"tag": "D_loss", "value": 1.5, "step": 3}, {"tag": "G_loss", "value": 0.25, "step": 3}]


def test_sample_row_helpers():
    """trainer.py:556-576 restated by hand: rectangles of int(256 * v) per value with w, h capped at 255, stop at the
    first absent box; the caption is the words up to the first 0 joined by blanks with non-ASCII characters dropped."""
    import numpy as np
    import torch
    from mogan_amd.attngan.trainer import caption_sentence, draw_bbox_lines
    img = torch.zeros(3, 2, 256, 256)
    boxes = np.array([[0.25, 0.5, 0.125, 0.25], [0.0, 0.0, 1.2, 0.999], [-1, -1, -1, -1], [0.5, 0.5, 0.1, 0.1]], dtype=np.float32)
    draw_bbox_lines(img, boxes, 256)
    want = torch.zeros(256, 256)
    for (x, y, w, h) in ((64, 128, 32, 64), (0, 0, 255, 255)):
        want[y, x:x + w] = 1; want[y:y + h, x] = 1; want[y + h, x:x + w] = 1; want[y:y + h, x + w] = 1
    assert all(torch.equal(img[i, c], want) for i in range(3) for c in range(2))
    assert want[128 + 13, 128] == 0                        # the box after the absent one is not drawn
    ix = {0: '<end>', 1: 'a', 2: u'caf\u00e9', 3: 'zebra'}
    assert caption_sentence(np.array([3, 2, 1, 0, 3]), ix) == "zebra caf a"
    assert caption_sentence(np.array([0, 1]), ix) == ""
