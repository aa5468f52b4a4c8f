"""Train-path helpers of code/coco/stackgan/miscc/utils.py with the reference's names and signatures
(bbox -> affine matrices :19-52, KL_loss :62-65, compute_discriminator_loss :68-107, compute_generator_loss
:110-125, weights_init :129-139, save_model :162-176, mkdir_p :179-186).  `gpus` is accepted and ignored: data
parallelism is one process per GPU (..engine).  save_img_results (:144-160) and load_validation_data (:54-64) are the
host-side helpers of the sample grids and of `GANTrainer.sample`."""
from ....attngan.miscc.utils import (compute_transformation_matrix, compute_transformation_matrix_inverse,  # noqa: F401
                                    mkdir_p)
from ... import losses as _losses
from ...datasets import load_validation_data  # noqa: F401
from ...logging_utils import save_img_results  # noqa: F401
from ...trainer_base import save_model, weights_init  # noqa: F401

KL_loss = _losses.KL_loss


def compute_discriminator_loss(netD, real_imgs, fake_imgs, real_labels, fake_labels, local_label, transf_matrices,
                               transf_matrices_inv, conditions, gpus=None):
    errD, r, w, f = _losses.discriminator_loss(netD, real_imgs, fake_imgs, local_label, transf_matrices,
                                               transf_matrices_inv, conditions)
    return errD, r.item(), w.item(), f.item()


def compute_generator_loss(netD, fake_imgs, real_labels, local_label, transf_matrices, transf_matrices_inv, conditions,
                           gpus=None):
    return _losses.generator_loss(netD, fake_imgs, local_label, transf_matrices, transf_matrices_inv, conditions)
