"""Inception-v3 trunk for CNN_ENCODER (reference: code/coco/attngan/model.py:207-313 builds it from
torchvision.models.inception_v3 -- torchvision==0.2.1, requirements.txt:31, not vendored).

Own restatement of the published architecture (Szegedy et al. 2015, "Rethinking the Inception
Architecture"; torchvision layer names so that DAMSM `image_encoder*.pth` checkpoints load):
BasicConv2d = conv(bias=False) + BN(eps=1e-3) + ReLU; Conv2d_1a..4a, Mixed_5b/c/d (A), 6a (B),
6b-e (C, 7x7 widths 128/160/160/192), 7a (D), 7b/c (E).  The encoder is frozen and in eval mode in
the train step (trainer.py:62-66), so BN is a per-channel affine on running statistics, fused with the
ReLU in one HIP launch after each conv; gradients flow to the input image only (dgrad, no wgrad).
Parity: the Inception arithmetic itself is unpinned by the reference; tests check
this module against its torch-CPU restatement in tests/.
"""
import math
import os

import torch
import torch.nn as nn

from ..hip import ops
from .model_base import HipConv2d, HipBatchNorm2d


class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = HipConv2d(cin, cout, bias=False, **kw)
        self.bn = HipBatchNorm2d(cout, eps=0.001)
        self._folded = None

    def folded(self):
        """eval-mode BN as (scale, shift); cached -- the encoder is frozen."""
        if self._folded is None or self._folded[0].device != self.bn.weight.device:
            with torch.no_grad():
                scale = (self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)).contiguous()
                shift = (self.bn.bias - self.bn.running_mean * scale).contiguous()
            self._folded = (scale, shift)
        return self._folded

    def forward(self, x):
        if self.training:
            return self.bn.fused(self.conv(x), ops.ACT_RELU)
        scale, shift = self.folded()             # eval: BN folded; affine + ReLU ride in the conv epilogue
        return ops.conv2d_affine_relu(x, self.conv.weight, scale, shift, self.conv.stride[0], self.conv.padding)


_BRANCH_STREAMS = []
# measured: alone, the captured encoder runs 10 % faster with parallel branches (fwd 5.2 -> 4.7 ms, bwd 6.5 -> 5.8 ms);
# inside the train step, whose other streams already fill the GPU, it is 2.5 % SLOWER (251.6 vs 257.9 img/s) -> off
PARALLEL_BRANCHES = os.environ.get("MOGAN_INCEPTION_STREAMS", "0") != "0"


def _parallel(fns):
    """Evaluate the independent branches of a Mixed block.  While the encoder is being captured into a hipGraph (the
    train engine replays it, trainer.py:_encoder) the branches are forked onto side streams and become parallel
    branches of the graph -- the block's small GEMMs (1-2 GFLOP each) then overlap instead of running one after the
    other with split-K to fill the chip; autograd replays each branch's backward on its stream.  Outside a capture
    the branches simply run in order on the current stream."""
    if not (PARALLEL_BRANCHES and torch.cuda.is_current_stream_capturing()) or len(fns) < 2:
        return [f() for f in fns]
    while len(_BRANCH_STREAMS) < len(fns) - 1:
        _BRANCH_STREAMS.append(torch.cuda.Stream())
    cur = torch.cuda.current_stream()
    outs = [None] * len(fns)
    for i, f in enumerate(fns[1:]):
        st = _BRANCH_STREAMS[i]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs[i + 1] = f()
    outs[0] = fns[0]()
    for i in range(len(fns) - 1):
        cur.wait_stream(_BRANCH_STREAMS[i])
    return outs


class InceptionA(nn.Module):
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1 = BasicConv2d(cin, 48, kernel_size=1)
        self.branch5x5_2 = BasicConv2d(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, padding=1)
        self.branch_pool = BasicConv2d(cin, pool_features, kernel_size=1)

    def forward(self, x):
        b3, b5, b1, bp = _parallel([
            lambda: self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),      # longest chain first
            lambda: self.branch5x5_2(self.branch5x5_1(x)),
            lambda: self.branch1x1(x),
            lambda: self.branch_pool(ops.avg_pool2d(x, 3, 1, 1))])
        return torch.cat([b1, b5, b3, bp], 1)


class InceptionB(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        bd, b3, mp = _parallel([
            lambda: self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
            lambda: self.branch3x3(x),
            lambda: ops.max_pool2d(x, 3, 2)])
        return torch.cat([b3, bd, mp], 1)


class InceptionC(nn.Module):
    def __init__(self, cin, channels_7x7):
        super().__init__()
        c7 = channels_7x7
        self.branch1x1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7_2 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        def dbl():
            bd = self.branch7x7dbl_1(x)
            for m in (self.branch7x7dbl_2, self.branch7x7dbl_3, self.branch7x7dbl_4, self.branch7x7dbl_5):
                bd = m(bd)
            return bd
        bd, b7, b1, bp = _parallel([
            dbl,
            lambda: self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x))),
            lambda: self.branch1x1(x),
            lambda: self.branch_pool(ops.avg_pool2d(x, 3, 1, 1))])
        return torch.cat([b1, b7, bd, bp], 1)


class InceptionD(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch3x3_2 = BasicConv2d(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        def b7f():
            b7 = self.branch7x7x3_1(x)
            for m in (self.branch7x7x3_2, self.branch7x7x3_3, self.branch7x7x3_4):
                b7 = m(b7)
            return b7
        b7, b3, mp = _parallel([b7f, lambda: self.branch3x3_2(self.branch3x3_1(x)), lambda: ops.max_pool2d(x, 3, 2)])
        return torch.cat([b3, b7, mp], 1)


class InceptionE(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        def b3f():
            b3 = self.branch3x3_1(x)
            return torch.cat([self.branch3x3_2a(b3), self.branch3x3_2b(b3)], 1)

        def bdf():
            bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
            return torch.cat([self.branch3x3dbl_3a(bd), self.branch3x3dbl_3b(bd)], 1)
        bd, b3, b1, bp = _parallel([bdf, b3f, lambda: self.branch1x1(x),
                                    lambda: self.branch_pool(ops.avg_pool2d(x, 3, 1, 1))])
        return torch.cat([b1, b3, bd, bp], 1)


TRUNK = (("Conv2d_1a_3x3", lambda: BasicConv2d(3, 32, kernel_size=3, stride=2)),
         ("Conv2d_2a_3x3", lambda: BasicConv2d(32, 32, kernel_size=3)),
         ("Conv2d_2b_3x3", lambda: BasicConv2d(32, 64, kernel_size=3, padding=1)),
         ("Conv2d_3b_1x1", lambda: BasicConv2d(64, 80, kernel_size=1)),
         ("Conv2d_4a_3x3", lambda: BasicConv2d(80, 192, kernel_size=3)),
         ("Mixed_5b", lambda: InceptionA(192, 32)), ("Mixed_5c", lambda: InceptionA(256, 64)),
         ("Mixed_5d", lambda: InceptionA(288, 64)), ("Mixed_6a", lambda: InceptionB(288)),
         ("Mixed_6b", lambda: InceptionC(768, 128)), ("Mixed_6c", lambda: InceptionC(768, 160)),
         ("Mixed_6d", lambda: InceptionC(768, 160)), ("Mixed_6e", lambda: InceptionC(768, 192)),
         ("Mixed_7a", lambda: InceptionD(768)), ("Mixed_7b", lambda: InceptionE(1280)),
         ("Mixed_7c", lambda: InceptionE(2048)))


def init_trunk(module):
    """Random init for runs without the DAMSM checkpoint: He-normal conv weights (keeps activations
    O(1) through the 47-conv-deep eval-mode trunk; torchvision's own truncnorm(0.1) init overflows fp32
    when BN uses identity running statistics), BN gamma=1, beta=0."""
    for m in module.modules():
        if isinstance(m, HipConv2d):
            fan_in = m.weight[0].numel()
            nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_in))
        elif isinstance(m, HipBatchNorm2d):
            nn.init.constant_(m.weight, 1.0)
            nn.init.constant_(m.bias, 0.0)
