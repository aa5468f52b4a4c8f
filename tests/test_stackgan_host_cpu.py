"""Host-side checks of the StackGAN-family mirror that need no GPU: module construction on CPU gives the
reference's state_dict keys (read from the fixtures the reference wrote), the yml files load into each tree's
cfg with the reference's widths, the synthetic datasets collate into what each reference train loop unpacks,
and the product refuses to compute without a GPU."""
import importlib
import os

import numpy as np
import pytest
import torch

from helpers import load_pkg
from stackgan_cases import CASES, golden

load_pkg()
from mogan_amd.hip.lib import MoganHipError  # noqa: E402
from mogan_amd.stackgan import synthetic  # noqa: E402
from mogan_amd.stackgan.datasets_synth import SyntheticDataset  # noqa: E402

PKG = {"coco": "coco", "clevr": "clevr", "mnist": "multi_mnist"}


def _mods(tree):
    model = importlib.import_module("mogan_amd.stackgan.%s.model" % PKG[tree])
    config = importlib.import_module("mogan_amd.stackgan.%s.miscc.config" % PKG[tree])
    return model, config


@pytest.mark.parametrize("case", list(CASES))
def test_state_dict_keys_match_reference(case):
    tree, stage, B, kw = CASES[case]
    model, config = _mods(tree)
    cfg = config.cfg
    cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.GAN.CONDITION_DIM = kw["gf_dim"], kw["df_dim"], kw["cond_dim"]
    cfg.GAN.R_NUM = kw.get("r_num", 2)
    if tree == "coco":
        cfg.TEXT.DIMENSION = kw["text_dim"]
    G = model.STAGE2_G(model.STAGE1_G()) if stage == 2 else model.STAGE1_G()
    D = model.STAGE2_D() if stage == 2 else model.STAGE1_D()
    g = golden("stackgan_%s_nets" % case)
    assert list(G.state_dict().keys()) == [str(k) for k in g["g_keys"]]
    assert list(D.state_dict().keys()) == [str(k) for k in g["d_keys"]]
    if stage == 2:                     # S/model.py:318-319: stage I is frozen inside STAGE2_G
        assert all(not p.requires_grad for p in G.STAGE1_G.parameters())


@pytest.mark.parametrize("tree,yml,gf,df,cd", [("coco", "coco_s1_train.yml", 192, 96, 128),
                                               ("coco", "coco_s2_train.yml", 192, 96, 128),
                                               ("clevr", "clevr_train.yml", 96, 48, 16),
                                               ("mnist", "mnist_train.yml", 128, 64, 128)])
def test_yml_loads(tree, yml, gf, df, cd):
    model, config = _mods(tree)
    config.cfg_from_file(os.path.join(os.path.dirname(model.__file__), "cfg", yml))
    cfg = config.cfg
    assert (cfg.GAN.GF_DIM, cfg.GAN.DF_DIM, cfg.GAN.CONDITION_DIM) == (gf, df, cd)
    assert cfg.TRAIN.GENERATOR_LR == 0.0002 and cfg.USE_BBOX_LAYOUT is True
    with pytest.raises(KeyError):
        from mogan_amd.stackgan.config import _merge_a_into_b
        _merge_a_into_b({"NOT_A_KEY": 1}, cfg)


def test_synthetic_batches_follow_the_dataset_rules():
    b = synthetic.make_batch("coco", 8, stage=2, seed=3)
    for key in ("bbox", "bbox_s2"):
        bb = b[key].numpy()
        present = bb[..., 0] >= 0
        assert np.all(bb[present][:, 0] + bb[present][:, 2] <= 0.9991)       # S/miscc/datasets.py:120-121
        assert np.all(bb[~present] == -1.0)
    assert b["label_one_hot"].shape == (8, 3, 81) and torch.all(b["label_one_hot"].sum(-1) == 1)
    assert b["real_imgs"].shape == (8, 3, 256, 256) and b["txt_embedding"].shape == (8, 1024)
    c = synthetic.make_batch("clevr", 8, seed=3)
    assert c["label_one_hot"].shape == (8, 4, 13) and torch.all(c["label_one_hot"].sum(-1) == 2)   # shape + colour
    absent = c["bbox"][..., 0] < 0
    assert torch.all(c["label_one_hot"][absent][:, 3] == 1) and torch.all(c["label_one_hot"][absent][:, 12] == 1)
    m = synthetic.make_batch("mnist", 8, seed=3)
    assert m["bbox"].dtype == torch.float64 and m["tm"].dtype == torch.float32 and m["real_imgs"].shape[1] == 1


@pytest.mark.parametrize("tree,stage", [("coco", 1), ("coco", 2), ("clevr", 1), ("mnist", 1)])
def test_synthetic_dataset_collates_like_the_reference_loader(tree, stage):
    dl = torch.utils.data.DataLoader(SyntheticDataset(tree, stage, 4, text_dim=16), batch_size=2)
    d = next(iter(dl))
    size = 256 if stage == 2 else 64
    assert d[0].shape == (2, 1 if tree == "mnist" else 3, size, size)
    if tree == "coco":
        assert len(d) == 4 and d[2].shape == (2, 3, 1) and d[3].shape == (2, 16)
        assert (isinstance(d[1], list) and len(d[1]) == 2) if stage == 2 else d[1].shape == (2, 3, 4)
    elif tree == "clevr":
        assert len(d) == 4 and d[1][0].shape == (2, 4, 2, 3) and d[2].shape == (2, 4, 13)
    else:
        assert len(d) == 3 and d[1].dtype == torch.float64 and d[2].shape == (2, 3, 10)


def test_no_cpu_fallback():
    model, config = _mods("clevr")
    config.cfg.GAN.GF_DIM, config.cfg.GAN.DF_DIM, config.cfg.GAN.CONDITION_DIM = 4, 4, 16
    G = model.STAGE1_G()
    b = synthetic.make_batch("clevr", 2, seed=0)
    with pytest.raises(MoganHipError):
        G(b["z"], b["tmi"], b["label_one_hot"])
