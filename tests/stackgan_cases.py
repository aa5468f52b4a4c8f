"""Case table shared by the StackGAN-family tests (mirrors tests/golden/make_golden_stackgan.py CASES)."""
import os

import numpy as np
import torch

from helpers import GOLDEN, det_array
from oracle import stackgan_oracle as S

# case -> (tree, stage, batch, oracle cfg)
CASES = {
    "coco_s1": ("coco", 1, 3, dict(gf_dim=4, df_dim=4, cond_dim=128, text_dim=12)),
    "coco_s2": ("coco", 2, 2, dict(gf_dim=192, df_dim=4, cond_dim=128, text_dim=12, r_num=1)),
    "clevr": ("clevr", 1, 3, dict(gf_dim=4, df_dim=4, cond_dim=16)),
    "mnist": ("mnist", 1, 3, dict(gf_dim=4, df_dim=4, cond_dim=128)),
}


def T(name, shape, scale=1.0):
    return torch.from_numpy(det_array(name, shape, scale))


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def oracle_cfg(case):
    tree, stage, B, kw = CASES[case]
    return S.SCfg(tree, stage=stage, **kw), tree, stage, B


def specs(cfg):
    if cfg.stage == 2:
        return S.stage2_g_spec(cfg), S.stage2_d_spec(cfg)
    return S.stage1_g_spec(cfg), S.stage1_d_spec(cfg)


def det_state(spec, tag):
    """Same deterministic fill as helpers.det_fill_state, from a key->shape spec."""
    sd = {}
    for k, shp in spec.items():
        name = tag + k
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            sd[k] = torch.from_numpy(det_array(name, shp, 0.1))
        elif k.endswith("running_var"):
            sd[k] = torch.from_numpy(np.abs(det_array(name, shp, 0.1)) + 1.0)
        elif len(shp) == 1 and k.endswith("weight"):
            sd[k] = torch.from_numpy(det_array(name, shp, 0.1, 1.0))
        elif len(shp) == 1:
            sd[k] = torch.from_numpy(det_array(name, shp, 0.1))
        else:
            sd[k] = torch.from_numpy(det_array(name, shp, 1.0 / np.sqrt(int(np.prod(shp[1:])))))
    return sd


def sub(img):
    s = max(1, img.shape[-1] // 16)
    return img[:, :, ::s, ::s]


def frozen_keys(cfg):
    """parameters the reference excludes from optimizerG (S/model.py:318-319)."""
    return "STAGE1_G." if cfg.stage == 2 else None
