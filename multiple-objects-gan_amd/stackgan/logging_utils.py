"""Host-side logging of the StackGAN-family trainers: the scalar summaries
(S/trainer.py:238-251: D_loss, D_loss_real, D_loss_wrong, D_loss_fake, G_loss, KL_loss every 500 iterations) and the
sample grids of `save_img_results` (S/miscc/utils.py:144-160, C/miscc/utils.py:163-179, M/miscc/utils.py:142-158).

The reference writes tensorboard event files and calls torchvision.utils.save_image; neither package is installed here
(nor vendored in the reference), so: scalars go to <Log>/scalars.jsonl (one {"tag","value","step"} object per line) and,
when torch.utils.tensorboard is importable, also to an event file; `save_image` restates torchvision 0.2.1's
make_grid(normalize=True) + save (global min/max normalisation, 8 images per row, 2-pixel black padding, truncation to
uint8) -- not pinned against torchvision itself."""
import json
import math
import os

import torch


class ScalarWriter(object):
    def __init__(self, log_dir):
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self._f = open(self.path, "a")
        self._tb = None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir)
        except Exception:           # tensorboard not installed
            self._tb = None

    def add_scalar(self, tag, value, step):
        value = float(value)
        self._f.write(json.dumps({"tag": tag, "value": value, "step": int(step)}) + "\n")
        self._f.flush()
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)

    def close(self):
        self._f.close()
        if self._tb is not None:
            self._tb.close()


def make_grid(tensor, nrow=8, padding=2, normalize=True):
    """(N,C,H,W) -> (3, rows*(H+pad)+pad, cols*(W+pad)+pad) in [0,1]; single-channel images are repeated to RGB."""
    t = tensor.detach().float().cpu()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.size(1) == 1:
        t = t.repeat(1, 3, 1, 1)
    if normalize:
        lo, hi = float(t.min()), float(t.max())
        t = (t.clamp(lo, hi) - lo) / (hi - lo + 1e-5)
    n = t.size(0)
    if n == 1:
        return t[0]
    cols = min(nrow, n)
    rows = int(math.ceil(float(n) / cols))
    h, w = t.size(2) + padding, t.size(3) + padding
    grid = t.new_zeros(3, h * rows + padding, w * cols + padding)
    for k in range(n):
        y, x = k // cols, k % cols
        grid[:, y * h + padding:y * h + h, x * w + padding:x * w + w] = t[k]
    return grid


def save_image(tensor, path, nrow=8, padding=2, normalize=True):
    from PIL import Image
    grid = make_grid(tensor, nrow, padding, normalize)
    Image.fromarray(grid.mul(255).clamp(0, 255).byte().permute(1, 2, 0).numpy()).save(path)
    return path


def save_img_results(data_img, fake, epoch, image_dir, vis_count=64):
    """real_samples.png + fake_samples_epoch_NNN.png, or lr_fake_samples_epoch_NNN.png when no real batch is given."""
    fake = fake[0:vis_count]
    if data_img is not None:
        return [save_image(data_img[0:vis_count], '%s/real_samples.png' % image_dir),
                save_image(fake, '%s/fake_samples_epoch_%03d.png' % (image_dir, epoch))]
    return [save_image(fake, '%s/lr_fake_samples_epoch_%03d.png' % (image_dir, epoch))]
