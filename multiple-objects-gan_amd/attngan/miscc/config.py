"""Global `cfg` of the coco-attngan variant: same keys/defaults and yml schema as the reference's
easydict config (code/coco/attngan/miscc/config.py:9-106), restated for py3 (the reference's
`_merge_a_into_b` uses `iteritems`/`has_key` and `yaml.load` without a Loader)."""
import numpy as np


class AttrDict(dict):
    """Minimal attribute-dict (easydict is not a dependency)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


__C = AttrDict()
cfg = __C

__C.DATASET_NAME = 'birds'
__C.CONFIG_NAME = ''
__C.DATA_DIR = ''
__C.IMG_DIR = ''
__C.GPU_ID = '0'
__C.CUDA = True
__C.WORKERS = 6
__C.RNN_TYPE = 'LSTM'
__C.B_VALIDATION = False

__C.TREE = AttrDict(BRANCH_NUM=3, BASE_SIZE=64)

__C.TRAIN = AttrDict(
    BATCH_SIZE=64, MAX_EPOCH=600, SNAPSHOT_INTERVAL=2000, DISCRIMINATOR_LR=2e-4, GENERATOR_LR=2e-4,
    ENCODER_LR=2e-4, RNN_GRAD_CLIP=0.25, FLAG=True, NET_E='', NET_G='', B_NET_D=True,
    GLOBAL_BATCH_LOSS=False,       # addition: see attngan/parallel.py
    SMOOTH=AttrDict(GAMMA1=5.0, GAMMA3=10.0, GAMMA2=5.0, LAMBDA=1.0))

__C.GAN = AttrDict(DF_DIM=64, GF_DIM=128, Z_DIM=100, CONDITION_DIM=100, R_NUM=2, B_ATTENTION=True,
                   B_DCGAN=False)

__C.TEXT = AttrDict(CAPTIONS_PER_IMAGE=10, EMBEDDING_DIM=256, WORDS_NUM=18)

# --- additions of this implementation (absent keys in a reference yml keep these defaults) ---
# affine_grid/grid_sample default changed between torch 0.4.1 (True) and >=1.3 (False)
__C.STN_ALIGN_CORNERS = False
# 0 = the reference's mask indexing (row b*Q+q uses mask[(b*Q+q) % B]); 1 = mask[b]
__C.ATT_MASK_MODE = 0
# Adam epsilon placement: 0 = torch>=1.x, 1 = torch 0.4.1 (see include/mogan_hip.h)
__C.ADAM_EPS_MODE = 0


def _merge_a_into_b(a, b):
    """Same contract as config.py:67-99: unknown keys and type mismatches are errors."""
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old_type = type(b[k])
        if isinstance(b[k], dict):
            if not isinstance(v, dict):
                raise ValueError('Type mismatch for config key: {}'.format(k))
            try:
                _merge_a_into_b(v, b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
            continue
        if old_type is not type(v):
            if isinstance(b[k], np.ndarray):
                v = np.array(v, dtype=b[k].dtype)
            elif isinstance(b[k], float) and isinstance(v, int):
                v = float(v)
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(b[k]), type(v), k))
        b[k] = v


def cfg_from_file(filename):
    """Load a yml config file and merge it into the default options (config.py:102-106)."""
    import yaml
    with open(filename, 'r') as f:
        yaml_cfg = yaml.safe_load(f)
    _merge_a_into_b(yaml_cfg, __C)


def set_coco_train_defaults():
    """cfg/coco_train.yml values (the benchmark configuration)."""
    cfg.TREE.BRANCH_NUM = 3
    cfg.TRAIN.SMOOTH.GAMMA1, cfg.TRAIN.SMOOTH.GAMMA2 = 4.0, 5.0
    cfg.TRAIN.SMOOTH.GAMMA3, cfg.TRAIN.SMOOTH.LAMBDA = 10.0, 50.0
    cfg.GAN.DF_DIM, cfg.GAN.GF_DIM, cfg.GAN.Z_DIM, cfg.GAN.R_NUM = 96, 48, 100, 3
    cfg.TEXT.EMBEDDING_DIM, cfg.TEXT.CAPTIONS_PER_IMAGE, cfg.TEXT.WORDS_NUM = 256, 5, 12
