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
