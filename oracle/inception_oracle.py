"""CPU oracle for the CNN_ENCODER trunk (reference: code/coco/attngan/model.py:207-313, which wraps
torchvision.models.inception_v3 -- torchvision==0.2.1, requirements.txt:31, NOT vendored under
/root/reference and not installed here).  PARITY UNPINNED for the Inception arithmetic: there is no
reference implementation or fixture to check against, so this is a functional torch-CPU restatement
of the published architecture (Szegedy et al., "Rethinking the Inception Architecture for Computer
Vision", 2015; torchvision layer names) that the HIP trunk is compared with.  What is downstream of
it (emb_features, emb_cnn_code, words_loss, sent_loss) IS pinned by tests/golden/losses.npz.
TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).

The encoder is frozen and in eval mode in the train step (trainer.py:62-66): BN uses running stats.
`sd` is a state_dict with torchvision key names (e.g. Mixed_5b.branch1x1.conv.weight).
"""
import torch
import torch.nn.functional as F


def _bc(sd, p, x, stride=1, padding=0):
    """BasicConv2d: conv(bias=False) -> BN(eps=1e-3, eval) -> ReLU."""
    x = F.conv2d(x, sd[p + ".conv.weight"], None, stride, padding)
    x = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                     sd[p + ".bn.bias"], False, 0.0, 0.001)
    return F.relu(x)


def _a(sd, p, x):
    b1 = _bc(sd, p + ".branch1x1", x)
    b5 = _bc(sd, p + ".branch5x5_2", _bc(sd, p + ".branch5x5_1", x), 1, 2)
    b3 = _bc(sd, p + ".branch3x3dbl_1", x)
    b3 = _bc(sd, p + ".branch3x3dbl_3", _bc(sd, p + ".branch3x3dbl_2", b3, 1, 1), 1, 1)
    bp = _bc(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1))
    return torch.cat([b1, b5, b3, bp], 1)


def _b(sd, p, x):
    b3 = _bc(sd, p + ".branch3x3", x, 2)
    bd = _bc(sd, p + ".branch3x3dbl_2", _bc(sd, p + ".branch3x3dbl_1", x), 1, 1)
    bd = _bc(sd, p + ".branch3x3dbl_3", bd, 2)
    return torch.cat([b3, bd, F.max_pool2d(x, 3, 2)], 1)


def _c(sd, p, x):
    b1 = _bc(sd, p + ".branch1x1", x)
    b7 = _bc(sd, p + ".branch7x7_1", x)
    b7 = _bc(sd, p + ".branch7x7_2", b7, 1, (0, 3))
    b7 = _bc(sd, p + ".branch7x7_3", b7, 1, (3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_1", x)
    for n, pad in (("2", (3, 0)), ("3", (0, 3)), ("4", (3, 0)), ("5", (0, 3))):
        bd = _bc(sd, p + ".branch7x7dbl_" + n, bd, 1, pad)
    bp = _bc(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1))
    return torch.cat([b1, b7, bd, bp], 1)


def _d(sd, p, x):
    b3 = _bc(sd, p + ".branch3x3_2", _bc(sd, p + ".branch3x3_1", x), 2)
    b7 = _bc(sd, p + ".branch7x7x3_1", x)
    b7 = _bc(sd, p + ".branch7x7x3_2", b7, 1, (0, 3))
    b7 = _bc(sd, p + ".branch7x7x3_3", b7, 1, (3, 0))
    b7 = _bc(sd, p + ".branch7x7x3_4", b7, 2)
    return torch.cat([b3, b7, F.max_pool2d(x, 3, 2)], 1)


def _e(sd, p, x):
    b1 = _bc(sd, p + ".branch1x1", x)
    b3 = _bc(sd, p + ".branch3x3_1", x)
    b3 = torch.cat([_bc(sd, p + ".branch3x3_2a", b3, 1, (0, 1)), _bc(sd, p + ".branch3x3_2b", b3, 1, (1, 0))], 1)
    bd = _bc(sd, p + ".branch3x3dbl_2", _bc(sd, p + ".branch3x3dbl_1", x), 1, 1)
    bd = torch.cat([_bc(sd, p + ".branch3x3dbl_3a", bd, 1, (0, 1)), _bc(sd, p + ".branch3x3dbl_3b", bd, 1, (1, 0))], 1)
    bp = _bc(sd, p + ".branch_pool", F.avg_pool2d(x, 3, 1, 1))
    return torch.cat([b1, b3, bd, bp], 1)


def cnn_encoder(sd, x):
    """model.py:252-313 -> (regions (B,nef,17,17), code (B,nef))."""
    x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
    x = _bc(sd, "Conv2d_1a_3x3", x, 2)
    x = _bc(sd, "Conv2d_2a_3x3", x)
    x = _bc(sd, "Conv2d_2b_3x3", x, 1, 1)
    x = F.max_pool2d(x, 3, 2)
    x = _bc(sd, "Conv2d_3b_1x1", x)
    x = _bc(sd, "Conv2d_4a_3x3", x)
    x = F.max_pool2d(x, 3, 2)
    for n in ("5b", "5c", "5d"):
        x = _a(sd, "Mixed_" + n, x)
    x = _b(sd, "Mixed_6a", x)
    for n in ("6b", "6c", "6d", "6e"):
        x = _c(sd, "Mixed_" + n, x)
    feat = x
    x = _d(sd, "Mixed_7a", x)
    x = _e(sd, "Mixed_7b", x)
    x = _e(sd, "Mixed_7c", x)
    x = F.avg_pool2d(x, 8).reshape(x.shape[0], -1)
    code = F.linear(x, sd["emb_cnn_code.weight"], sd["emb_cnn_code.bias"])
    return F.conv2d(feat, sd["emb_features.weight"]), code
