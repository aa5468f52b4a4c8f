"""Global `cfg` of this tree (see ../../config.py for the defaults and their reference lines)."""
from ... import config as _shared

cfg = _shared.make_cfg("coco")


def cfg_from_file(filename):
    _shared.cfg_from_file(filename, cfg)
