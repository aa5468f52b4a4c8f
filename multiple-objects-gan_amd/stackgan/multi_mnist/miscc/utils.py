"""Train-path helpers of code/multi-mnist/miscc/utils.py with the reference's names and signatures (bbox -> affine
matrices, compute_discriminator_loss / compute_generator_loss [M/miscc/utils.py:71-123], weights_init, save_model, mkdir_p).
The logits head is conditioned on the sum of the per-object one-hot labels.  `gpus` is accepted and ignored."""
from ....attngan.miscc.utils import (compute_transformation_matrix, compute_transformation_matrix_inverse,  # noqa: F401
                                    mkdir_p)
from ... import losses as _losses
from ...trainer_base import save_model, weights_init  # noqa: F401

_CLAMP = False


def compute_discriminator_loss(netD, real_imgs, fake_imgs, real_labels, fake_labels, local_label, transf_matrices,
                               transf_matrices_inv, gpus=None):
    cond = _losses.label_condition(local_label.detach(), _CLAMP)
    errD, r, w, f = _losses.discriminator_loss(netD, real_imgs, fake_imgs, local_label, transf_matrices,
                                               transf_matrices_inv, cond)
    return errD, r.item(), w.item(), f.item()


def compute_generator_loss(netD, fake_imgs, real_labels, local_label, transf_matrices, transf_matrices_inv, gpus=None):
    cond = _losses.label_condition(local_label.detach(), _CLAMP)
    return _losses.generator_loss(netD, fake_imgs, local_label, transf_matrices, transf_matrices_inv, cond)


def load_validation_data(datapath):
    """M/miscc/utils.py:59-68: labels, boxes of <datapath>/normal/{labels,bboxes}.pickle as tensors."""
    from ...datasets import load_validation_data as _lvd
    return _lvd(datapath, tree="mnist")
