"""Deterministic inputs of the attention-grid tests (shared by tests/golden/make_golden_vis.py and test_vis_cpu.py)."""
import numpy as np
import torch

from helpers import det_array


def vis_inputs():
    B, T, att = 8, 4, 16
    img = torch.from_numpy(det_array("vis_img", (B, 3, 32, 32))).clamp(-1, 1)
    lr = torch.from_numpy(det_array("vis_lr", (B, 3, 16, 16))).clamp(-1, 1)
    cap_lens = np.array([4, 4, 3, 3, 2, 2, 2, 1])
    captions = torch.zeros(B, T, dtype=torch.int64)
    for i in range(B):
        for j in range(cap_lens[i]):
            captions[i, j] = 1 + (3 * i + j) % 7
    ixtoword = {i: w for i, w in enumerate(["<end>", "a", "zebra", "standing", "giraffe", "on", "grassland", "bus"])}
    attn = torch.from_numpy(np.abs(det_array("vis_attn", (B, T, att, att)))).float()
    attn = attn / attn.sum(1, keepdim=True)
    return dict(B=B, T=T, att_sze=att, img=img, lr=lr, captions=captions, cap_lens=cap_lens, ixtoword=ixtoword,
                attn=[attn[i] for i in range(B)])
