"""The helpers of code/coco/attngan/miscc/utils.py that sit on the train path (SURVEY.md §8(a)
rows 26-27): bbox -> affine matrices, weights_init, EMA parameter copy/swap.  The visualisation
half of that file (build_super_images*, drawCaption) is host-side PIL drawing and out of scope."""
import errno
import os
from copy import deepcopy

import torch
import torch.nn as nn

from ..synthetic import bbox_to_theta


def compute_transformation_matrix_inverse(bbox):
    """miscc/utils.py:16-31: [[1/w,0,(2/w)(0.5-(x+w/2))],[0,1/h,(2/h)(0.5-(y+h/2))]]"""
    if bbox.is_cuda:
        from ...hip import ops
        return ops.bbox_to_theta(bbox)[1]
    return bbox_to_theta(bbox)[1]


def compute_transformation_matrix(bbox):
    """miscc/utils.py:34-49: [[w,0,2(x+w/2)-1],[0,h,2(y+h/2)-1]]"""
    if bbox.is_cuda:
        from ...hip import ops
        return ops.bbox_to_theta(bbox)[0]
    return bbox_to_theta(bbox)[0]


def weights_init(m):
    """miscc/utils.py:321-331: orthogonal conv/linear weights, BN gamma ~ N(1,0.02), beta = 0."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        nn.init.orthogonal_(m.weight.data, 1.0)
    elif classname.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
    elif classname.find('Linear') != -1:
        nn.init.orthogonal_(m.weight.data, 1.0)
        if m.bias is not None:
            m.bias.data.fill_(0.0)


def load_params(model, new_param):
    for p, new_p in zip(model.parameters(), new_param):
        p.data.copy_(new_p)
    from ...hip import ops
    ops.invalidate_all_packs()           # packed weight copies (hip/ops.WeightPacks) follow the new values at their next use


def copy_G_params(model):
    return deepcopy(list(p.data for p in model.parameters()))


def mkdir_p(path):
    try:
        os.makedirs(path)
    except OSError as exc:
        if exc.errno == errno.EEXIST and os.path.isdir(path):
            pass
        else:
            raise
