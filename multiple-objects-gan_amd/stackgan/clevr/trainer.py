"""`GANTrainer` of this tree (code/clevr/trainer.py): same constructor and `train` entry point; the loop body
lives in ..trainer_base / ..engine."""
from ..trainer_base import GANTrainerBase
from . import model as _model
from .miscc.config import cfg as _cfg


class GANTrainer(GANTrainerBase):
    cfg, model, tree = _cfg, _model, "clevr"
