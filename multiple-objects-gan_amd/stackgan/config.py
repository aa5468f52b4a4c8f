"""Default option trees of the three StackGAN-family variants, same keys / defaults / yml schema as
code/coco/stackgan/miscc/config.py:9-57, code/clevr/miscc/config.py:9-45 and
code/multi-mnist/miscc/config.py:9-47 (restated for py3; each per-tree `miscc/config.py` instantiates
its own global `cfg` from here)."""
from ..attngan.miscc.config import AttrDict, _merge_a_into_b


def make_cfg(tree):
    c = AttrDict()
    c.DATASET_NAME = {"coco": "coco", "clevr": "clevr", "multi_mnist": "multi-mnist"}[tree]
    c.CONFIG_NAME = ''
    c.GPU_ID = '0'
    c.CUDA = True
    c.WORKERS = 4 if tree == "multi_mnist" else 6
    c.NET_G = ''
    c.NET_D = ''
    c.DATA_DIR = ''
    c.VIS_COUNT = 64
    c.Z_DIM = 100
    c.IMSIZE = 64
    c.USE_LOCAL_PATHWAY = True
    c.USE_BBOX_LAYOUT = True
    c.TRAIN = AttrDict(FLAG=True, BATCH_SIZE=64, MAX_EPOCH=600, SNAPSHOT_INTERVAL=50, LR_DECAY_EPOCH=600,
                       DISCRIMINATOR_LR=2e-4, GENERATOR_LR=2e-4)
    c.GAN = AttrDict(CONDITION_DIM=128, DF_DIM=64, GF_DIM=128, R_NUM=4)
    if tree != "clevr":
        c.TRAIN.PRETRAINED_MODEL = ''
        c.TRAIN.PRETRAINED_EPOCH = 600
    if tree == "coco":
        c.EMBEDDING_TYPE = 'cnn-rnn'
        c.STAGE1_G = ''
        c.IMG_DIR = ''
        c.STAGE = 1
        c.TRAIN.COEFF = AttrDict(KL=2.0)
        c.TEXT = AttrDict(DIMENSION=1024)
    # --- additions of this implementation (see attngan/miscc/config.py) ---
    c.STN_ALIGN_CORNERS = False
    c.ADAM_EPS_MODE = 0
    return c


def cfg_from_file(filename, cfg):
    import yaml
    with open(filename, 'r') as f:
        yaml_cfg = yaml.safe_load(f)
    _merge_a_into_b(yaml_cfg, cfg)
