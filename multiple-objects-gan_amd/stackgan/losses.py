"""Losses of the StackGAN-family train steps, shared by coco-stackgan, clevr and multi-mnist
(code/coco/stackgan/miscc/utils.py:61-125, code/clevr/miscc/utils.py:91-144,
code/multi-mnist/miscc/utils.py:71-123).  The three reference files are the same computation up to
what conditions the 4x4 logits head:

    coco   : the caption's mu (detached)                                  S/miscc/utils.py:75,83
    clevr  : sum over objects of the 13-dim one-hot labels, clamped >= 0  C/miscc/utils.py:98-99
    mnist  : sum over objects of the 10-dim one-hot labels                M/miscc/utils.py:78

`real_labels` / `fake_labels` of the reference are constant 1 / 0 vectors (S/trainer.py:121-122), so the
BCEWithLogits kernels take the target as a scalar.  nn.parallel.data_parallel becomes a direct call
(one process per GPU; gradients are all-reduced over RCCL in engine.py).
"""
import torch

from ..hip import ops


def label_condition(local_label, clamp):
    """Sum of the per-object one-hots (B,K,L) -> (B,L); clevr additionally zeroes negatives."""
    cond = local_label.sum(1)
    return cond.clamp(min=0) if clamp else cond


def discriminator_loss(netD, real_imgs, fake_imgs, local_label, transf_matrices, transf_matrices_inv, cond,
                       real_features=None):
    """-> (errD, errD_real, errD_wrong, errD_fake) as 0-dim tensors (the per-tree wrappers call .item()).
    real and fake go through netD as two separate calls: separate BN batch statistics, like the reference."""
    B = real_imgs.size(0)
    cond, fake, local_label = cond.detach(), fake_imgs.detach(), local_label.detach()
    if real_features is None:        # (the engine may have run D(real) already, beside the G forward)
        real_features = netD(real_imgs, local_label, transf_matrices, transf_matrices_inv)
    fake_features = netD(fake, local_label, transf_matrices, transf_matrices_inv)
    errD_real = ops.bce_with_logits(netD.get_cond_logits(real_features, cond), 1.0)
    errD_wrong = ops.bce_with_logits(netD.get_cond_logits(real_features[:B - 1], cond[1:]), 0.0)
    errD_fake = ops.bce_with_logits(netD.get_cond_logits(fake_features, cond), 0.0)
    if netD.get_uncond_logits is not None:
        uncond_real = ops.bce_with_logits(netD.get_uncond_logits(real_features), 1.0)
        uncond_fake = ops.bce_with_logits(netD.get_uncond_logits(fake_features), 0.0)
        errD = (errD_real + uncond_real) / 2. + (errD_fake + errD_wrong + uncond_fake) / 3.
        errD_real = (errD_real + uncond_real) / 2.
        errD_fake = (errD_fake + uncond_fake) / 2.
    else:
        errD = errD_real + (errD_fake + errD_wrong) * 0.5
    return errD, errD_real.detach(), errD_wrong.detach(), errD_fake.detach()


def generator_loss(netD, fake_imgs, local_label, transf_matrices, transf_matrices_inv, cond):
    fake_features = netD(fake_imgs, local_label, transf_matrices, transf_matrices_inv)
    errG = ops.bce_with_logits(netD.get_cond_logits(fake_features, cond.detach()), 1.0)
    if netD.get_uncond_logits is not None:
        errG = errG + ops.bce_with_logits(netD.get_uncond_logits(fake_features), 1.0)
    return errG


def KL_loss(mu, logvar):
    """-0.5 * mean(1 + logvar - mu^2 - exp(logvar))   (S/miscc/utils.py:62-65)"""
    return ops.kl_loss(mu, logvar)
