"""Losses of the coco-attngan train step (mirror of code/coco/attngan/miscc/losses.py).

Same function names / arguments / return values as the reference.  Differences underneath:
  * `nn.parallel.data_parallel(netD, inputs, gpus)` (losses.py:146,152,193) becomes a direct call --
    data parallelism is one process per GPU with an RCCL gradient all-reduce (trainer.py);
  * words_loss evaluates all (image, caption) pairs in three batched launches instead of the
    reference's B-iteration python loop over func_attention (losses.py:72-112).
"""
import contextlib
import os

import numpy as np
import torch

from ...hip import ops
from .. import parallel
from .config import cfg


_nullctx = contextlib.nullcontext


def _call_d(netD, img, local_labels, transf_matrices, transf_matrices_inv):
    if local_labels is not None:
        return netD(img, local_labels, transf_matrices, transf_matrices_inv)
    return netD(img)


def _class_mask(class_ids, batch_size, device):
    """losses.py:24-35,69-71,116-121: mis-matched samples of the same class are masked out."""
    if class_ids is None:
        return None
    ids = np.asarray(class_ids).reshape(-1)
    m = (ids[:, None] == ids[None, :])
    np.fill_diagonal(m, False)
    if not m.any():
        return None            # COCO: class_ids are unique per sample (datasets.py:293-299)
    return torch.from_numpy(m).to(device)


def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """losses.py:11-17."""
    w12 = torch.sum(x1 * x2, dim)
    return (w12 / (torch.norm(x1, 2, dim) * torch.norm(x2, 2, dim)).clamp(min=eps)).squeeze()


def _labels_and_mask(labels, class_ids, batch_size, device):
    masks = _class_mask(class_ids, batch_size, device)
    if masks is not None:
        masks = masks.to(torch.uint8).contiguous()
    if labels is None:
        labels = torch.arange(batch_size, device=device)
    return labels.to(device=device, dtype=torch.int64).contiguous(), masks


def _image_side_only(t, what):
    """The fused DAMSM kernels differentiate with respect to the IMAGE side only -- in the generator step the text encoder
    is frozen (trainer.py:281-289).  A caller that wants text-encoder gradients (DAMSM pre-training, pretrain_DAMSM.py: out
    of scope) must not get silent zeros."""
    if torch.is_grad_enabled() and t.requires_grad:
        raise NotImplementedError("%s requires grad: the fused DAMSM losses give gradients to the image features only (the "
                                  "text encoder is frozen in the GAN train step); detach() the text embeddings" % what)


def _damsm_shape_check(C, S, T):
    """Limits of mogan_damsm_words_* (csrc/mogan_damsm.hip): T <= 32 words, S <= 640 regions, 64 KB of LDS."""
    lds = 4 * (C * T + T * (S + 1) + 3 * 10 * 32 + 4 * 32)
    if T > 32 or S > 640 or lds > 64 * 1024:
        raise ValueError("words_loss: caption length T=%d (<= 32), region count S=%d (<= 640) or the kernel's LDS need of "
                         "%d bytes (<= 65536; nef=%d) exceed the fused DAMSM kernels' limits" % (T, S, lds, C))


def sent_loss(cnn_code, rnn_code, labels, class_ids, batch_size, eps=1e-8):
    """losses.py:20-59: cosine matrix of the image / sentence codes * gamma3 and cross-entropy both ways, fused
    (mogan_damsm_sent_* + mogan_damsm_ce_*: 2 launches forward, 2 backward).  Gradient: cnn_code."""
    if labels is None:
        return None, None
    _image_side_only(rnn_code, "sent_loss: rnn_code")
    lab, masks = _labels_and_mask(labels, class_ids, batch_size, cnn_code.device)
    return ops.damsm_sent(cnn_code, rnn_code.detach(), lab, masks, cfg.TRAIN.SMOOTH.GAMMA3, eps)


def words_loss(img_features, words_emb, labels, cap_lens, class_ids, batch_size):
    """losses.py:62-132.  words_emb (B,nef,T), img_features (B,nef,17,17).
    similarities[b, i] = log sum_t exp(gamma2 * cos(word_{i,t}, context_{b,i,t})) * gamma3, where the region context
    comes from func_attention(word_i, feature_b) (GlobalAttention.py:31-69) -- all B*B pairs in one fused launch
    (mogan_damsm_words_fwd) instead of the reference's B-iteration python loop, then the two cross-entropies.
    Gradient: img_features (the text encoder is frozen in the generator step, trainer.py:281-289)."""
    B = batch_size
    ih, iw = img_features.shape[2], img_features.shape[3]
    dev = img_features.device
    _image_side_only(words_emb, "words_loss: words_emb")
    _damsm_shape_check(img_features.shape[1], ih * iw, words_emb.shape[2])
    lens = cap_lens.to(dev).to(torch.int32).reshape(B).contiguous()
    lab, masks = _labels_and_mask(labels, class_ids, B, dev)
    l0, l1, a2 = ops.damsm_words(img_features, words_emb.detach(), lens, lab, masks, cfg.TRAIN.SMOOTH.GAMMA1,
                                 cfg.TRAIN.SMOOTH.GAMMA2, cfg.TRAIN.SMOOTH.GAMMA3)
    att_maps = None
    if labels is None or not torch.is_grad_enabled():         # visualisation (losses.py:87-91): caption i on image i
        lens_h = [int(v) for v in cap_lens.tolist()]
        att_maps = [a2[i, i, :lens_h[i]].reshape(1, lens_h[i], ih, iw).contiguous() for i in range(B)]
    if labels is None:
        return None, None, att_maps
    return l0, l1, att_maps


# D(real) and D(fake.detach()) of a discriminator update as ONE pass over the batch [real; fake] with BatchNorm statistics per
# half (round 5): the reference makes two calls (losses.py:146,152), i.e. two sets of batch statistics and two running-
# statistics updates, real first -- the grouped BatchNorm / deep-block kernels (groups = 2) compute exactly those, while every
# convolution sees 2B images: the deep layers' weights (up to 300 MB per layer of D_NET256) stream once per direction instead
# of twice and their GEMMs have twice the columns.  OPT-IN and off by default (module attribute, no environment switch; the
# benchmarked step makes the reference's two calls): measured neutral in the step, profiles/r05_ab.txt; kept under test.  Networks
# without the PAIRED attribute (D_NET64: the object pathway already runs as a 3-group batch) always keep the two calls.
D_PAIR = False


def paired(netD):
    return D_PAIR and bool(getattr(netD, "PAIRED", False)) and netD.training and torch.is_grad_enabled()


def _paired_features(netD, real_imgs, fake_imgs):
    B = real_imgs.shape[0]
    x = ops._cat_batch(real_imgs.detach(), fake_imgs.detach())
    trace0 = len(ops.ACT_TRACE) if ops.ACT_TRACE is not None else None
    ops.TRACE_GROUPS = 2
    try:
        f = netD(x, groups=2)
    finally:
        ops.TRACE_GROUPS = 1
    if trace0 is not None:
        # test hook: the recorded activations back into the reference's call order -- all of D(real), then all of D(fake)
        new = []
        for e in ops.ACT_TRACE[trace0:]:
            new.extend([e] if len(e) == 3 else [(e[0], e[1][:B], 0), (e[0], e[1][B:], 1)])
        del ops.ACT_TRACE[trace0:]
        for g in (0, 1):
            ops.ACT_TRACE.extend((e[0], e[1]) for e in new if e[2] == g)
    return ops.split_batch(f, B)


def discriminator_loss(netD, real_imgs, fake_imgs, conditions, real_labels, fake_labels, gpus=None,
                       local_labels=None, transf_matrices=None, transf_matrices_inv=None, real_features=None):
    """losses.py:136-174.  D(real) and D(fake.detach()) are two separate calls in the reference (separate BN batch
    statistics) -- and so they are here by default; with losses.D_PAIR = True (opt-in) one pass over [real; fake] with per-half
    statistics where the network supports it; real_labels/fake_labels are the constant 1/0 vectors of prepare_labels."""
    if real_features is None and local_labels is None and paired(netD) and real_imgs.shape == fake_imgs.shape:
        real_features, fake_features = _paired_features(netD, real_imgs, fake_imgs)
    else:
        if real_features is None:   # (the engine may have run D(real) already, concurrently with the G forward)
            real_features = _call_d(netD, real_imgs, local_labels, transf_matrices, transf_matrices_inv)
        fake_features = _call_d(netD, fake_imgs.detach(), local_labels, transf_matrices, transf_matrices_inv)
    if parallel.enabled():          # heads, wrong-pair shift and BCE means over the gathered batch (parallel.py)
        real_features, fake_features = parallel.gather_cat(real_features), parallel.gather_cat(fake_features)
        conditions = parallel.gather_const(conditions)
    batch_size = real_features.size(0)
    cond_real_errD = ops.bce(netD.COND_DNET(real_features, conditions), 1.0)
    cond_fake_errD = ops.bce(netD.COND_DNET(fake_features, conditions), 0.0)
    cond_wrong_errD = ops.bce(netD.COND_DNET(real_features[:(batch_size - 1)], conditions[1:batch_size]), 0.0)
    if netD.UNCOND_DNET is not None:
        real_errD = ops.bce(netD.UNCOND_DNET(real_features), 1.0)
        fake_errD = ops.bce(netD.UNCOND_DNET(fake_features), 0.0)
        # ((real + cond_real) / 2 + (fake + cond_fake + cond_wrong) / 3) in one launch
        return ops.scalar_sum([real_errD, cond_real_errD, fake_errD, cond_fake_errD, cond_wrong_errD],
                              [0.5, 0.5, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0])
    return ops.scalar_sum([cond_real_errD, cond_fake_errD, cond_wrong_errD], [1.0, 0.5, 0.5])


# The discriminator loss in two halves (round 5).  errD = R(real images) + F(fake images): losses.py:146-174 evaluates
#   R = (BCE(uncond(real), 1) + BCE(cond(real), 1)) / 2 + BCE(cond(real[:-1], conditions[1:]), 0) / 3      [no fake image in it]
#   F = (BCE(uncond(fake), 0) + BCE(cond(fake), 0)) / 3
# (without the unconditional head: R = cond_real + cond_wrong / 2, F = cond_fake / 2).  d errD / d theta = dR + dF, and the
# parameter gradients accumulate in place, so R can be evaluated AND back-propagated as soon as the batch is on the device --
# beside the generator's forward (and the tail of the previous step) -- and only F waits for the fake images: one of the two
# backward passes through D_i leaves the window between the generator's forward and its backward, which the D_NET256 branch
# bounds (trainer.TrainEngine).  Same terms, same gradients up to the order of two additions.  BatchNorm running statistics
# keep the reference's CALL order (D(real), D(fake), cond(real), cond(fake), cond(wrong)): only the conditional head's
# BatchNorm sees more than one call per half, and its wrong-pair call -- made early, with the real half -- defers its running
# update behind the fake call's (hip/ops.BN_DEFER, mogan_bn_running_update).
def discriminator_loss_real(netD, real_imgs, conditions, local_labels=None, transf_matrices=None, transf_matrices_inv=None):
    """R of the comment above; returns (R, pending running-statistics updates of the wrong-pair head call)"""
    real_features = _call_d(netD, real_imgs, local_labels, transf_matrices, transf_matrices_inv)
    batch_size = real_features.size(0)
    cond_real_errD = ops.bce(netD.COND_DNET(real_features, conditions), 1.0)
    pending, ops.BN_DEFER = [], []
    try:
        cond_wrong_errD = ops.bce(netD.COND_DNET(real_features[:(batch_size - 1)], conditions[1:batch_size]), 0.0)
    finally:
        pending, ops.BN_DEFER = ops.BN_DEFER, None
    if netD.UNCOND_DNET is not None:
        real_errD = ops.bce(netD.UNCOND_DNET(real_features), 1.0)
        return ops.scalar_sum([real_errD, cond_real_errD, cond_wrong_errD], [0.5, 0.5, 1.0 / 3.0]), pending
    return ops.scalar_sum([cond_real_errD, cond_wrong_errD], [1.0, 0.5]), pending


def discriminator_loss_fake(netD, fake_imgs, conditions, pending, local_labels=None, transf_matrices=None,
                            transf_matrices_inv=None):
    """F of the comment above; applies the deferred running-statistics updates behind the fake call of the conditional head"""
    fake_features = _call_d(netD, fake_imgs.detach(), local_labels, transf_matrices, transf_matrices_inv)
    cond_fake_errD = ops.bce(netD.COND_DNET(fake_features, conditions), 0.0)
    if pending:
        with torch.no_grad():
            ops.bn_apply_deferred(pending)
    if netD.UNCOND_DNET is not None:
        fake_errD = ops.bce(netD.UNCOND_DNET(fake_features), 0.0)
        return ops.scalar_sum([fake_errD, cond_fake_errD], [1.0 / 3.0, 1.0 / 3.0])
    return ops.scalar_sum([cond_fake_errD], [0.5])


# The discriminator loss in two halves (round 5): OPT-IN and off by default (module attribute; -1 % in the step, profiles/r05_ab.txt).
D_SPLIT = False


def split_d_loss():
    """may the engine take the two-halves form?  (not with the gathered global-batch heads, not with the paired pass)"""
    return D_SPLIT and not D_PAIR and not parallel.enabled()


def generator_d_branch(netD, fake_img, sent_emb, local_labels=None, transf_matrices=None, transf_matrices_inv=None):
    """One term of losses.py:187-203: BCE(cond logits of D(fake), 1) [+ BCE(uncond logits, 1)]."""
    features = _call_d(netD, fake_img, local_labels, transf_matrices, transf_matrices_inv)
    if parallel.enabled():
        features, sent_emb = parallel.gather_cat(features), parallel.gather_const(sent_emb)
    g_loss = ops.bce(netD.COND_DNET(features, sent_emb), 1.0)
    if netD.UNCOND_DNET is not None:
        g_loss = ops.scalar_sum([ops.bce(netD.UNCOND_DNET(features), 1.0), g_loss])
    return g_loss


def generator_damsm_branch(image_encoder, fake_img, words_embs, sent_emb, match_labels, cap_lens, class_ids, batch_size):
    """losses.py:205-221: Inception features of the last fake image -> DAMSM word and sentence losses (* LAMBDA)."""
    region_features, cnn_code = image_encoder(fake_img)
    if parallel.enabled():          # DAMSM contrast over the gathered batch
        region_features, cnn_code = parallel.gather_cat(region_features), parallel.gather_cat(cnn_code)
        words_embs, sent_emb = parallel.gather_const(words_embs), parallel.gather_const(sent_emb)
        cap_lens = parallel.gather_const(cap_lens.to(region_features.device))
        class_ids = parallel.gather_ids(class_ids)
        batch_size = region_features.size(0)
        match_labels = torch.arange(batch_size, device=region_features.device)
    lam = cfg.TRAIN.SMOOTH.LAMBDA
    w_loss0, w_loss1, _ = words_loss(region_features, words_embs, match_labels, cap_lens, class_ids, batch_size)
    w_loss = ops.scalar_sum([w_loss0, w_loss1], [lam, lam])
    s_loss0, s_loss1 = sent_loss(cnn_code, sent_emb, match_labels, class_ids, batch_size)
    s_loss = ops.scalar_sum([s_loss0, s_loss1], [lam, lam])
    return w_loss, s_loss


def generator_total(parts, numDs):
    """Sum in the reference's order: g_loss0, g_loss1, g_loss2, then w_loss, s_loss (losses.py:203,221)."""
    return ops.scalar_sum([parts['g_loss%d' % i] for i in range(numDs)] + [parts['w_loss'], parts['s_loss']])


def generator_loss(netsD, image_encoder, fake_imgs, real_labels, words_embs, sent_emb, match_labels,
                   cap_lens, class_ids, gpus=None, local_labels=None, transf_matrices=None,
                   transf_matrices_inv=None, return_logs=True, streams=None):
    """losses.py:177-226.  Returns (errG_total, logs) like the reference; `return_logs=False` skips the
    .item() host syncs (needed under hipGraph capture) and returns a dict of 0-dim tensors instead.
    `streams` (len(netsD)+1 side streams, already waiting on the current one): the D_i branches and the
    Inception/DAMSM branch are independent until their losses are summed, so each runs on its own stream and
    autograd replays the backward of each branch on that same stream."""
    numDs = len(netsD)
    batch_size = real_labels.size(0)
    parts = {}

    def on(k):
        return torch.cuda.stream(streams[k]) if streams is not None else _nullctx()

    for i in range(numDs):
        with on(i):
            kw = dict(local_labels=local_labels, transf_matrices=transf_matrices,
                      transf_matrices_inv=transf_matrices_inv) if i == 0 else {}
            parts['g_loss%d' % i] = generator_d_branch(netsD[i], fake_imgs[i], sent_emb, **kw)
    with on(numDs):
        w_loss, s_loss = generator_damsm_branch(image_encoder, fake_imgs[numDs - 1], words_embs, sent_emb,
                                                match_labels, cap_lens, class_ids, batch_size)
    if streams is not None:
        cur = torch.cuda.current_stream()
        for st in streams[:numDs + 1]:
            cur.wait_stream(st)
    parts['w_loss'], parts['s_loss'] = w_loss, s_loss
    errG_total = generator_total(parts, numDs)
    if not return_logs:
        return errG_total, parts
    logs = ''.join('%s: %.2f ' % (k, v.item()) for k, v in parts.items())
    return errG_total, logs


def KL_loss(mu, logvar):
    """losses.py:230-234."""
    return ops.kl_loss(mu, logvar)
