#!/usr/bin/env python
"""Golden fixture tests/golden/datasets.npz: the REFERENCE's data-contract functions
(code/coco/attngan/datasets.py) run on CPU through ref_shim.py on the deterministic inputs of tests/datasets_cases.py:

  * crop_imgs (95-137): 268 -> 256 random crop, random horizontal flip, box rescale with both clamps, absent objects --
    per case the numpy seed, the scaled boxes (float64, exact) and four corner samples + the sum of the cropped image;
  * TextDataset.get_one_hot_labels (341-349): -1 -> class 80, one-hot (3, 81);
  * TextDataset.get_caption (311-329): zero padding / the random WORDS_NUM-subset of a longer caption (seeded);
  * TextDataset.get_transformation_matrices (331-339) on the scaled boxes of the crop cases;
  * prepare_data (28-68), train and eval form: the sort by caption length (ties included) applied to every field.

TextDataset.__init__ needs torchvision.transforms and the data files; its methods above do not touch `self` beyond
max_objects / captions, so they are called on a stand-in object.  get_imgs' ToPILImage / Resize (torchvision) is not
importable here: the multi-scale resize is pinned against PIL itself in tests/test_feeder_cpu.py instead.
Usage: python tests/golden/make_golden_datasets.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import ref_shim                                   # noqa: E402
import datasets_cases as C                        # noqa: E402


def main():
    ns = ref_shim.load()
    ns.cfg.CUDA = False
    ns.cfg.TEXT.WORDS_NUM = C.T_WORDS
    ref = importlib.import_module("datasets")     # /root/reference/code/coco/attngan/datasets.py (sys.path set by the shim)
    out = {}
    img = C.crop_image()
    boxes, probes, mats = [], [], []
    stand_in = types.SimpleNamespace(max_objects=3)
    for case in range(C.N_CROP):
        np.random.seed(500 + case)
        crop, scaled = ref.crop_imgs(img, C.crop_boxes(case))
        assert tuple(crop.shape) == (3, 256, 256) and scaled.dtype == np.float64
        boxes.append(scaled)
        probes.append([float(crop[0, 0, 0]), float(crop[0, 0, 255]), float(crop[2, 255, 0]), float(crop[1, 255, 255]),
                       float(crop.double().sum())])
        tm, tmi = ref.TextDataset.get_transformation_matrices(stand_in, scaled)
        mats.append(np.stack([tm.numpy(), tmi.numpy()]))
    out["crop_boxes"] = np.stack(boxes)
    out["crop_probe"] = np.asarray(probes, dtype=np.float64)
    out["crop_mats"] = np.stack(mats)
    out["onehot"] = np.stack([ref.TextDataset.get_one_hot_labels(stand_in, lab).numpy() for lab in C.label_cases()])
    caps, lens = [], []
    for i, cap in enumerate(C.caption_cases()):
        np.random.seed(900 + i)
        x, n = ref.TextDataset.get_caption(types.SimpleNamespace(captions=[cap]), 0)
        caps.append(x)
        lens.append(n)
    out["cap_x"], out["cap_len"] = np.stack(caps), np.asarray(lens)
    for tag, ev in (("pd", False), ("pde", True)):
        imgs, caps_, lens_, cls, keys, tms, label, bbox = C.batch_case()
        data = [list(imgs), caps_, lens_, cls, keys, list(tms), label] + ([bbox] if ev else [])
        res = ref.prepare_data(data, eval=ev)
        for i, im in enumerate(res[0]):
            out["%s_img%d" % (tag, i)] = im.numpy()
        out[tag + "_captions"], out[tag + "_lens"] = res[1].numpy(), res[2].numpy()
        out[tag + "_class_ids"], out[tag + "_keys"] = np.asarray(res[3]), np.array(res[4])
        out[tag + "_tm"], out[tag + "_tmi"], out[tag + "_label"] = res[5][0].numpy(), res[5][1].numpy(), res[6].numpy()
        if ev:
            out[tag + "_bbox"] = res[7].numpy()
    path = os.path.join(HERE, "datasets.npz")
    np.savez_compressed(path, **out)
    print("wrote datasets.npz %.1f KB, %d arrays" % (os.path.getsize(path) / 1024, len(out)))


if __name__ == "__main__":
    main()
