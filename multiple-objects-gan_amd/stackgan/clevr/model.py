"""clevr networks with the reference's class surface (code/clevr/model.py): `STAGE1_G()`, `STAGE1_D()`,
bound to this tree's global cfg.  Bodies live in ..nets."""
from .. import nets
from ..nets import ResBlock, conv3x3, upBlock  # noqa: F401
from .miscc.config import cfg

VARIANT = nets.CLEVR


class D_GET_LOGITS(nets.D_GET_LOGITS):
    """C/model.py:44-71: the conditioning vector is the 13-dim label sum; the (unused) unconditional head is
    hard-coded to 109 input channels."""

    def __init__(self, ndf, nef, bcondition=True):
        super(D_GET_LOGITS, self).__init__(ndf, cfg.GAN.CONDITION_DIM, bcondition, cond_dim=13, uncond_in=109)


class BBOX_NET(nets.BBOX_NET):
    def __init__(self):
        super(BBOX_NET, self).__init__(cfg, cfg.GAN.CONDITION_DIM, cfg.GAN.CONDITION_DIM)


class STAGE1_G(nets.STAGE1_G):
    def __init__(self):
        super(STAGE1_G, self).__init__(cfg, VARIANT)

    def forward(self, noise, transf_matrices_inv, label_one_hot, num_objects=4):
        """-> fake_img   (C/model.py:158-192)"""
        return self.generate(None, noise, transf_matrices_inv, label_one_hot, num_objects)[0]


class STAGE1_D(nets.STAGE1_D):
    def __init__(self):
        super(STAGE1_D, self).__init__(cfg, VARIANT, cond_dim=13, uncond_in=109)
