#!/usr/bin/env python
"""Golden fixture tests/golden/vis.npz: the REFERENCE's build_super_images / build_super_images2
(code/coco/attngan/miscc/utils.py:88-316) run on CPU through ref_shim.py on deterministic inputs.

Two things the reference needs are absent in this image and are handed to it from this repo, so the fixture pins
everything AROUND them (canvas layout, colours, normalisation, blending, ordering), not them:
  * skimage.transform.pyramid_expand  <- mogan_amd.attngan.miscc.vis.pyramid_expand
  * ImageFont.truetype('Pillow/Tests/fonts/FreeMono.ttf', 50)  <- PIL's built-in font
Usage: python tests/golden/make_golden_vis.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import ref_shim                                   # noqa: E402
from vis_cases import vis_inputs                  # noqa: E402
import mogan_loader                               # noqa: E402
mogan_loader.load()
from mogan_amd.attngan.miscc import vis           # noqa: E402
from PIL import ImageFont                         # noqa: E402


def main():
    ns = ref_shim.load()
    u = ns.utils
    u.skimage.transform.pyramid_expand = vis.pyramid_expand
    default, orig = ImageFont.load_default(), ImageFont.truetype
    u.ImageFont.truetype = lambda font=None, *a, **k: default if isinstance(font, str) else orig(font, *a, **k)
    c = vis_inputs()
    sup, sent = u.build_super_images(c["img"].clone(), c["captions"], c["ixtoword"], c["attn"], c["att_sze"],
                                     batch_size=c["B"], max_word_num=c["T"])
    sup_lr, _ = u.build_super_images(c["img"].clone(), c["captions"], c["ixtoword"], c["attn"], c["att_sze"],
                                     lr_imgs=c["lr"].clone(), batch_size=c["B"], max_word_num=c["T"])
    sup2, _ = u.build_super_images2(c["img"][:3].clone(), c["captions"][:3], c["cap_lens"][:3], c["ixtoword"],
                                    [a[:n] for a, n in zip(c["attn"][:3], c["cap_lens"][:3])], c["att_sze"],
                                    vis_size=32, topK=2)
    path = os.path.join(HERE, "vis.npz")
    np.savez_compressed(path, sup=sup, sup_lr=sup_lr, sup2=sup2, sentences=np.array([" ".join(s) for s in sent]))
    print("wrote vis.npz %.1f KB" % (os.path.getsize(path) / 1024), sup.shape, sup_lr.shape, sup2.shape)


if __name__ == "__main__":
    main()
