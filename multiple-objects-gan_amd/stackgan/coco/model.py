"""coco-stackgan networks with the reference's class surface (code/coco/stackgan/model.py):
`STAGE1_G()`, `STAGE1_D()`, `STAGE2_G(stage1_g)`, `STAGE2_D()`, bound to this tree's global cfg.
Bodies live in ..nets (shared with clevr / multi-mnist)."""
from .. import nets
from ..nets import ResBlock, conv3x3, upBlock  # noqa: F401  (names the reference module exports)
from .miscc.config import cfg

VARIANT = nets.COCO


def stn(image, transformation_matrix, size):
    """S/model.py:107-111 with the align_corners choice made explicit."""
    from ...hip import ops
    return ops.stn(image, transformation_matrix, tuple(size), bool(cfg.STN_ALIGN_CORNERS))


class CA_NET(nets.CA_NET):
    def __init__(self):
        super(CA_NET, self).__init__(cfg)


class D_GET_LOGITS(nets.D_GET_LOGITS):
    pass


class BBOX_NET(nets.BBOX_NET):
    def __init__(self):
        super(BBOX_NET, self).__init__(cfg, cfg.GAN.CONDITION_DIM, cfg.GAN.CONDITION_DIM)


class STAGE1_G(nets.STAGE1_G):
    def __init__(self):
        super(STAGE1_G, self).__init__(cfg, VARIANT)

    def forward(self, text_embedding, noise, transf_matrices_inv, label_one_hot, max_objects=3, eps=None):
        """-> (None, fake_img, mu, logvar, local_labels)   (S/model.py:201-235)"""
        c_code, mu, logvar = self.ca_net(text_embedding, eps)
        fake_img, local_labels = self.generate(c_code, noise, transf_matrices_inv, label_one_hot, max_objects)
        return None, fake_img, mu, logvar, local_labels


class STAGE1_D(nets.STAGE1_D):
    def __init__(self):
        super(STAGE1_D, self).__init__(cfg, VARIANT)


class STAGE2_G(nets.STAGE2_G):
    def __init__(self, STAGE1_G):
        super(STAGE2_G, self).__init__(cfg, VARIANT, STAGE1_G)


class STAGE2_D(nets.STAGE2_D):
    def __init__(self):
        super(STAGE2_D, self).__init__(cfg, VARIANT)
