"""multi-mnist networks with the reference's class surface (code/multi-mnist/model.py): `STAGE1_G()`,
`STAGE1_D()`, bound to this tree's global cfg.  Bodies live in ..nets."""
from .. import nets
from ..nets import ResBlock, conv3x3, upBlock  # noqa: F401
from .miscc.config import cfg

VARIANT = nets.MNIST


class D_GET_LOGITS(nets.D_GET_LOGITS):
    pass


class BBOX_NET(nets.BBOX_NET):
    def __init__(self):
        super(BBOX_NET, self).__init__(cfg, 10, 128)


class STAGE1_G(nets.STAGE1_G):
    def __init__(self):
        super(STAGE1_G, self).__init__(cfg, VARIANT)

    def forward(self, noise, transf_matrices_inv, label_one_hot, num_digits_per_image=3):
        """-> (None, fake_img)   (M/model.py:158-190).  `self.label` exists (state_dict keys) but the
        reference feeds the one-hot itself to the object pathway, so it never receives a gradient."""
        return None, self.generate(None, noise, transf_matrices_inv, label_one_hot, num_digits_per_image)[0]


class STAGE1_D(nets.STAGE1_D):
    def __init__(self):
        super(STAGE1_D, self).__init__(cfg, VARIANT)
