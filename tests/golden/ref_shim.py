"""Test-process shim that imports the *reference's own* AttnGAN python modules in THIS
container (CPU, torch 2.x, py3) so golden vectors can be captured from them.

NOT shipped, NOT imported by any -m gpu test, bench.py or smoke(): /root/reference does
not exist on the GPU box. Only tests/golden/make_golden.py (and the optional
reference-vs-oracle cross-check test, skipped when the reference is absent) use it.

What is shimmed and why:
  * easydict / torchvision / skimage / nltk are not installed -> tiny stub modules;
  * torch.cuda.FloatTensor is hard-coded in model.py (model.py:106,336,388,391,684)
    -> aliased to torch.FloatTensor so the model runs on CPU;
  * torch.ByteTensor masks are rejected by masked_fill_ in torch 2.x -> bool;
  * nn.parallel.data_parallel needs CUDA devices -> direct call.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("MOGAN_REFERENCE", "/root/reference")
ATTNGAN_DIR = os.path.join(REF_ROOT, "code", "coco", "attngan")


def reference_available():
    return os.path.isfile(os.path.join(ATTNGAN_DIR, "model.py"))


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)
        super().__setattr__(k, v)

    __setitem__ = __setattr__


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _install_stubs():
    _stub("easydict", EasyDict=_EasyDict)
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.utils = _stub("torchvision.utils")
    tv.transforms = _stub("torchvision.transforms")
    sk = _stub("skimage")
    sk.transform = _stub("skimage.transform")
    nl = _stub("nltk")
    nl.tokenize = _stub("nltk.tokenize", RegexpTokenizer=object)


_ORIG = {}


def _patch_torch(align_corners):
    """Stubs + torch patches shared by every tree. `align_corners` patches the default of
    affine_grid/grid_sample (torch 0.4.1 semantics = True, torch>=1.3 default = False)."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    _install_stubs()
    if "FloatTensor" not in _ORIG:
        _ORIG["FloatTensor"] = torch.cuda.FloatTensor
        _ORIG["affine_grid"] = torch.nn.functional.affine_grid
        _ORIG["grid_sample"] = torch.nn.functional.grid_sample
        _ORIG["data_parallel"] = nn.parallel.data_parallel
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.DoubleTensor = torch.DoubleTensor
    torch.ByteTensor = lambda a: torch.as_tensor(a).bool()

    def _dp(module, inputs, gpus=None):
        return module(*inputs) if isinstance(inputs, tuple) else module(inputs)
    nn.parallel.data_parallel = _dp

    ag, gs = _ORIG["affine_grid"], _ORIG["grid_sample"]
    torch.nn.functional.affine_grid = \
        lambda theta, size, align_corners=align_corners: ag(theta, size, align_corners=align_corners)
    torch.nn.functional.grid_sample = \
        lambda inp, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners: \
        gs(inp, grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners)


def load(align_corners=False):
    """Returns a namespace with the reference's AttnGAN modules: model, GlobalAttention, losses, utils, cfg."""
    _patch_torch(align_corners)
    if ATTNGAN_DIR not in sys.path:
        sys.path.insert(0, ATTNGAN_DIR)
    ns = types.SimpleNamespace()
    ns.config = importlib.import_module("miscc.config")
    ns.cfg = ns.config.cfg
    ns.GlobalAttention = importlib.import_module("GlobalAttention")
    ns.model = importlib.import_module("model")
    ns.losses = importlib.import_module("miscc.losses")
    ns.utils = importlib.import_module("miscc.utils")
    return ns


def set_cfg(cfg, **kw):
    """Set the cfg fields directly (cfg_from_file is py2-only: config.py:74-76)."""
    cfg.CUDA = False
    cfg.TRAIN.FLAG = True
    cfg.TREE.BRANCH_NUM = kw.get("BRANCH_NUM", 3)
    cfg.GAN.GF_DIM = kw.get("GF_DIM", 48)
    cfg.GAN.DF_DIM = kw.get("DF_DIM", 96)
    cfg.GAN.Z_DIM = kw.get("Z_DIM", 100)
    cfg.GAN.CONDITION_DIM = 100
    cfg.GAN.R_NUM = kw.get("R_NUM", 3)
    cfg.TEXT.EMBEDDING_DIM = kw.get("EMBEDDING_DIM", 256)
    cfg.TEXT.WORDS_NUM = kw.get("WORDS_NUM", 12)
    cfg.TRAIN.SMOOTH.GAMMA1 = 4.0
    cfg.TRAIN.SMOOTH.GAMMA2 = 5.0
    cfg.TRAIN.SMOOTH.GAMMA3 = 10.0
    cfg.TRAIN.SMOOTH.LAMBDA = 50.0


# ------------------------------------------------------------------ StackGAN-family trees
TREE_DIRS = {"coco": os.path.join(REF_ROOT, "code", "coco", "stackgan"),
             "clevr": os.path.join(REF_ROOT, "code", "clevr"),
             "mnist": os.path.join(REF_ROOT, "code", "multi-mnist")}
_ALL_TREE_DIRS = [ATTNGAN_DIR] + list(TREE_DIRS.values())


def load_tree(tree, align_corners=False):
    """Import model / miscc.config / miscc.utils of one of the sibling trees (they all use the same
    top-level module names, so previously imported ones are purged first).  Extra stubs: cPickle,
    torchfile, tensorboard (py2 / uninstalled imports of miscc/utils.py and trainer.py)."""
    _patch_torch(align_corners)
    import pickle
    sys.modules["cPickle"] = pickle
    _stub("torchfile")
    _stub("tensorboard", summary=object, FileWriter=object)
    for name in [n for n in sys.modules if n in ("model", "trainer", "GlobalAttention", "datasets")
                 or n == "miscc" or n.startswith("miscc.")]:
        del sys.modules[name]
    for d in _ALL_TREE_DIRS:
        while d in sys.path:
            sys.path.remove(d)
    sys.path.insert(0, TREE_DIRS[tree])
    ns = types.SimpleNamespace()
    ns.config = importlib.import_module("miscc.config")
    ns.cfg = ns.config.cfg
    ns.utils = importlib.import_module("miscc.utils")
    ns.model = importlib.import_module("model")
    return ns


def set_tree_cfg(cfg, tree, stage=1, **kw):
    cfg.CUDA = False
    cfg.TRAIN.FLAG = True
    cfg.USE_BBOX_LAYOUT = kw.get("USE_BBOX_LAYOUT", True)
    cfg.Z_DIM = kw.get("Z_DIM", 100)
    cfg.GAN.GF_DIM = kw["GF_DIM"]
    cfg.GAN.DF_DIM = kw["DF_DIM"]
    cfg.GAN.CONDITION_DIM = kw["CONDITION_DIM"]
    cfg.GAN.R_NUM = kw.get("R_NUM", 2)
    if tree == "coco":
        cfg.STAGE = stage
        cfg.TEXT.DIMENSION = kw.get("TEXT_DIM", 1024)
        cfg.TRAIN.COEFF.KL = 2.0
