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


def _versions(*tensors):
    """(device, storage address, in-place version counter) of every tensor: the key of the frozen encoder's derived-weight
    caches.  load_state_dict / copy_ / a broadcast into the parameters (TrainEngine.sync_replicas) bump the version
    counter, .to(device) replaces the storage -- either way the cache is rebuilt instead of serving stale weights."""
    return tuple((str(t.device), t.data_ptr(), t._version) for t in tensors)


class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = HipConv2d(cin, cout, bias=False, **kw)
        self.bn = HipBatchNorm2d(cout, eps=0.001)
        self._folded = None

    def folded(self):
        """eval-mode BN as (scale, shift); cached -- the encoder is frozen."""
        key = _versions(self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        if self._folded is None or self._folded[2] != key:
            with torch.no_grad():
                scale = (self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)).contiguous()
                shift = (self.bn.bias - self.bn.running_mean * scale).contiguous()
            self._folded = (scale, shift, key)
        return self._folded[:2]

    def forward(self, x):
        if self.training:
            return self.bn.fused(self.conv(x), ops.ACT_RELU)
        scale, shift = self.folded()             # eval: BN folded; affine + ReLU ride in the conv epilogue
        return ops.conv2d_affine_relu(x, self.conv.weight, scale, shift, self.conv.stride[0], self.conv.padding)


_BRANCH_STREAMS = []
# measured: alone, the captured encoder runs 10 % faster with parallel branches (fwd 5.2 -> 4.7 ms, bwd 6.5 -> 5.8 ms);
# inside the train step, whose other streams already fill the GPU, it is 2.5 % SLOWER (251.6 vs 257.9 img/s) -> off
PARALLEL_BRANCHES = False


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


# ======================================================================================================================
# Frozen trunk: explicit forward / backward (the train step's path; model.py:62-66 freezes the encoder, trainer.py:329-333
# sends only the image gradient through it).
#
# The module-by-module evaluation above costs, per Mixed block, one launch per BasicConv2d, a torch.cat of the branch
# outputs, and in the backward pass a ReLU/affine kernel per convolution plus the slice copies and additions autograd
# inserts around cat and around an activation with several consumers.  With the weights frozen none of that bookkeeping is
# needed:
#   * the 1x1 convolutions that read the same block input (branch1x1, branch5x5_1 / 7x7_1 / 3x3_1, ...dbl_1) run as ONE
#     convolution over the concatenated filters (mogan_conv2d_affine_fwd_ex: output channels below `msplit` go straight
#     into the block's output tensor, the rest into the block's scratch tensor); their data gradient is ONE convolution
#     with the output-channel axis (= K of the GEMM) concatenated;
#   * every branch writes its result straight into its channel slice of the block output (no cat), every data gradient
#     reads its slice of the output gradient (no split) and ADDS into the gradient of its input (no add kernels);
#   * ReLU backward is fused into the data-gradient kernels: a convolution whose input is a ReLU output zeroes its result
#     where that input is 0 (mogan_conv2d_dgrad_ex relu_of), so a gradient buffer is always "already masked";
#   * eval-mode BN: y = relu(scale * conv(x; W) + shift) forward; backward uses W' = scale * W (precomputed once).
# Same arithmetic as the modules above up to the rounding of W' (checked against the module path and the CPU restatement).
FAST_TRUNK = True


class _Slice:
    """channels [c0, c0 + C) of a dense (B, Ctot, H, W) tensor"""
    __slots__ = ("t", "c0", "C")

    def __init__(self, t, c0=0, C=None):
        self.t, self.c0, self.C = t, c0, (t.shape[1] - c0 if C is None else C)

    @property
    def H(self):
        return self.t.shape[2]

    @property
    def W(self):
        return self.t.shape[3]

    @property
    def ptr(self):
        return self.t.data_ptr() + 4 * self.c0 * self.t.shape[2] * self.t.shape[3]

    @property
    def bstride(self):
        return self.t.stride(0)              # (a gradient buffer may itself be a channel slice: _Tape.scratch_for_group)

    @property
    def dense(self):
        return self.c0 == 0 and self.C == self.t.shape[1] and self.t.is_contiguous()


FLIPPED_DGRAD = True


class _FrozenConv:
    """one BasicConv2d (or a group of same-input 1x1 ones) with eval-mode BN folded: forward (w, scale, shift), backward w * scale"""

    def __init__(self, mods):
        mods = list(mods)
        m0 = mods[0].conv
        self.k, self.stride, self.pad = m0.kernel_size, m0.stride[0], m0.padding
        for m in mods[1:]:
            assert m.conv.kernel_size == self.k and m.conv.stride[0] == self.stride and m.conv.padding == self.pad
        with torch.no_grad():
            self.w = torch.cat([m.conv.weight for m in mods], 0).contiguous()
            folded = [m.folded() for m in mods]
            self.scale = torch.cat([f[0] for f in folded]).contiguous()
            self.shift = torch.cat([f[1] for f in folded]).contiguous()
            self.wb = (self.w * self.scale.view(-1, 1, 1, 1)).contiguous()
            # stride-1 convolutions with a spatial extent: the data gradient runs as a FORWARD convolution of dY with the
            # flipped, (ci, co)-transposed filters (K-contiguous filter operand instead of a stride-KH*KW gather)
            self.wflip = None
            if FLIPPED_DGRAD and self.stride == 1 and self.k != (1, 1):
                self.wflip = self.wb.flip(2, 3).transpose(0, 1).contiguous()
        self.cout, self.cin = self.w.shape[0], self.w.shape[1]

    def out_hw(self, H, W):
        return ((H + 2 * self.pad[0] - self.k[0]) // self.stride + 1, (W + 2 * self.pad[1] - self.k[1]) // self.stride + 1)


class _Op:
    """one launch-able piece of the trunk: a convolution (BasicConv2d, eval-mode BN folded), or a pooling"""
    __slots__ = ("kind", "fc", "x", "y", "y2", "relu_in", "plain", "k", "s", "pad", "idx")

    def __init__(self, kind, x, y, fc=None, y2=None, relu_in=True, plain=False, k=0, s=0, pad=0):
        self.kind, self.fc, self.x, self.y, self.y2, self.relu_in, self.plain = kind, fc, x, y, y2, relu_in, plain
        self.k, self.s, self.pad, self.idx = k, s, pad, None


GROUPED = True      # 0: one launch per convolution (A/B measurements)


class _Tape:
    """forward launches + what the backward pass needs; one per forward call.  Work is issued in LEVELS: the operations of
    a level are independent, its convolutions go out as ONE grouped launch (mogan_conv2d_*_group)."""

    def __init__(self, B, device):
        self.B, self.dev, self.grads, self.bwd_levels = B, device, {}, []

    def new(self, C, H, W):
        return torch.empty((self.B, C, H, W), dtype=torch.float32, device=self.dev)

    def conv(self, fc, x, y, y2=None, relu_in=True, plain=False):
        assert x.C == fc.cin and y.C + (y2.C if y2 is not None else 0) == fc.cout
        return _Op("conv", x, y, fc=fc, y2=y2, relu_in=relu_in, plain=plain)

    @staticmethod
    def avgpool(x, y, k, s, pad, relu_in=True):
        assert x.dense and y.dense
        return _Op("avgpool", x, y, relu_in=relu_in, k=k, s=s, pad=pad)

    @staticmethod
    def maxpool(x, y, k, s, relu_in=True):
        assert x.dense
        return _Op("maxpool", x, y, relu_in=relu_in, k=k, s=s)

    # ---- forward ----------------------------------------------------------------------------------------------
    def _fwd_args(self, op):
        from ..hip.lib import ConvFwdArgs
        fc, x, y, y2 = op.fc, op.x, op.y, op.y2
        return ConvFwdArgs(x.ptr, x.bstride, fc.w.data_ptr(), fc.scale.data_ptr(), fc.shift.data_ptr(), y.ptr, y.bstride,
                           y2.ptr if y2 is not None else None, y2.bstride if y2 is not None else 0, y.C, self.B, fc.cin,
                           x.H, x.W, fc.cout, fc.k[0], fc.k[1], fc.stride, fc.pad[0], fc.pad[1], 1)

    def run(self, level):
        """issue one level of the forward pass"""
        import ctypes
        from ..hip.lib import ConvFwdArgs, call, stream_ptr, workspace
        convs = []
        for op in level:
            if op.kind == "avgpool":
                call("mogan_avgpool_fwd", op.x.ptr, op.y.ptr, self.B * op.x.C, op.x.H, op.x.W, op.k, op.s, op.pad, stream_ptr())
            elif op.kind == "maxpool":
                op.idx = torch.empty((self.B, op.x.C, op.y.H, op.y.W), dtype=torch.uint8, device=self.dev)
                call("mogan_maxpool_fwd_ex", op.x.ptr, op.y.ptr, op.y.bstride, op.idx.data_ptr(), self.B, op.x.C, op.x.H,
                     op.x.W, op.k, op.s, stream_ptr())
            elif op.plain:
                fc, x, y = op.fc, op.x, op.y
                assert x.dense and y.dense and op.y2 is None
                wsp, wsn = workspace(self.dev)
                call("mogan_conv2d_affine_fwd", x.ptr, fc.w.data_ptr(), fc.scale.data_ptr(), fc.shift.data_ptr(), y.ptr,
                     self.B, fc.cin, x.H, x.W, fc.cout, fc.k[0], fc.k[1], fc.stride, fc.pad[0], fc.pad[1], 1, wsp, wsn,
                     stream_ptr())
            else:
                convs.append(op)
        wsp, wsn = workspace(self.dev)
        for chunk in ([convs] if GROUPED else [[c] for c in convs]):
            for i in range(0, len(chunk), 4):
                part = chunk[i:i + 4]
                arr = (ConvFwdArgs * len(part))(*[self._fwd_args(op) for op in part])
                call("mogan_conv2d_affine_fwd_group", len(part), ctypes.cast(arr, ctypes.c_void_p), wsp, wsn, stream_ptr())

    def plan_backward(self, levels):
        """the backward schedule of what was just run: levels from the outputs towards the inputs"""
        self.bwd_levels.append(levels)

    # ---- backward ---------------------------------------------------------------------------------------------
    def grad_of(self, t):
        g = self.grads.get(id(t))
        if g is None:
            g = self.grads[id(t)] = [torch.empty_like(t), set()]      # buffer, written channel ranges
        return g

    def _dst(self, x):
        """gradient slice of x + whether a contribution is already there (then: accumulate)"""
        g, written = self.grad_of(x.t)
        key = (x.c0, x.C)
        acc = key in written
        written.add(key)
        return _Slice(g, x.c0, x.C), acc

    def seed(self, t, g, relu=True):
        """gradient arriving from outside for tensor t (a ReLU output): masked, added to what the tape already holds"""
        from ..hip.lib import call, stream_ptr
        dst, acc = self._dst(_Slice(t))
        g = g.contiguous()
        call("mogan_relu_bwd", t.data_ptr(), g.data_ptr(), dst.ptr, t.numel(), 1 if acc else 0, stream_ptr())

    def _bwd_level(self, level):
        import ctypes
        from ..hip.lib import ConvDgradArgs, call, stream_ptr, workspace
        wsp, wsn = workspace(self.dev)
        group = []
        for op in level:
            gy, _ = self.grad_of(op.y.t)
            if op.kind == "avgpool":
                dx, acc = self._dst(op.x)
                call("mogan_avgpool_bwd_ex", gy.data_ptr(), dx.ptr, op.x.ptr if op.relu_in else None, 1 if acc else 0,
                     self.B * op.x.C, op.x.H, op.x.W, op.k, op.s, op.pad, stream_ptr())
                continue
            if op.kind == "maxpool":
                dy = _Slice(gy, op.y.c0, op.y.C)
                dx, acc = self._dst(op.x)
                call("mogan_maxpool_bwd_ex", op.idx.data_ptr(), dy.ptr, dy.bstride, dx.ptr, op.x.ptr if op.relu_in else None,
                     1 if acc else 0, self.B, op.x.C, op.x.H, op.x.W, op.k, op.s, stream_ptr())
                continue
            fc, x, y, y2 = op.fc, op.x, op.y, op.y2
            if y2 is not None:
                # the group's gradient = [slice of the block-output gradient | gradient of the scratch tensor]: bring the
                # first part next to the second (the scratch gradient was allocated with room in front)
                g2, _ = self.grad_of(y2.t)
                full = g2._mogan_full
                call("mogan_copy_strided", _Slice(gy, y.c0, y.C).ptr, y.bstride, full.data_ptr(), full.stride(0), self.B,
                     y.C * y.H * y.W, stream_ptr())
                dy = _Slice(full)
            else:
                dy = _Slice(gy, y.c0, y.C)
            dx, acc = self._dst(x)
            if op.plain:
                assert not acc and dx.dense and dy.dense
                call("mogan_conv2d_dgrad", dy.ptr, fc.wb.data_ptr(), dx.ptr, self.B, fc.cin, x.H, x.W, fc.cout, fc.k[0],
                     fc.k[1], fc.stride, fc.pad[0], fc.pad[1], 0, wsp, wsn, stream_ptr())
                if op.relu_in:
                    call("mogan_relu_bwd", x.t.data_ptr(), dx.ptr, dx.ptr, x.t.numel(), 0, stream_ptr())
                continue
            if fc.wflip is not None:
                call("mogan_conv2d_fwd_ex", dy.ptr, dy.bstride, fc.wflip.data_ptr(), dx.ptr, dx.bstride,
                     x.ptr if op.relu_in else None, x.bstride, 1 if acc else 0, self.B, fc.cout, y.H, y.W, fc.cin, fc.k[0],
                     fc.k[1], 1, fc.k[0] - 1 - fc.pad[0], fc.k[1] - 1 - fc.pad[1], wsp, wsn, stream_ptr())
                continue
            group.append(ConvDgradArgs(dy.ptr, dy.bstride, fc.wb.data_ptr(), dx.ptr, dx.bstride,
                                       x.ptr if op.relu_in else None, x.bstride, 1 if acc else 0, self.B, fc.cin, x.H, x.W,
                                       fc.cout, fc.k[0], fc.k[1], fc.stride, fc.pad[0], fc.pad[1]))
        for chunk in ([group] if GROUPED else [[a] for a in group]):
            for i in range(0, len(chunk), 4):
                part = chunk[i:i + 4]
                arr = (ConvDgradArgs * len(part))(*part)
                call("mogan_conv2d_dgrad_group", len(part), ctypes.cast(arr, ctypes.c_void_p), wsp, wsn, stream_ptr())

    def backward(self):
        for levels in reversed(self.bwd_levels):
            for level in levels:
                self._bwd_level(level)

    def scratch_for_group(self, n_front, C, H, W):
        """scratch tensor of a grouped 1x1 convolution whose first n_front output channels live in the block output: its
        GRADIENT buffer gets n_front channels of room in front, so that the group's data gradient reads one tensor"""
        t = self.new(C, H, W)
        full = torch.empty((self.B, n_front + C, H, W), dtype=torch.float32, device=self.dev)
        g = full[:, n_front:]                    # the scratch gradient as a channel slice of `full`
        g._mogan_full = full
        self.grads[id(t)] = [g, set()]
        return t


class FrozenTrunk:
    """Folded weights of an eval-mode, frozen CNN_ENCODER trunk + its explicit forward/backward (see the section comment)."""

    def __init__(self, enc):
        g = lambda *names: _FrozenConv([getattr(blk, n) for n in names])
        self.stem = {n: _FrozenConv([getattr(enc, n)]) for n in ("Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3",
                                                               "Conv2d_3b_1x1", "Conv2d_4a_3x3")}
        self.blocks = []
        for name, _ in TRUNK[5:]:
            blk = getattr(enc, name)
            if isinstance(blk, InceptionA):
                fcs = dict(g1=g("branch1x1", "branch5x5_1", "branch3x3dbl_1"), b5=g("branch5x5_2"), d2=g("branch3x3dbl_2"),
                           d3=g("branch3x3dbl_3"), bp=g("branch_pool"))
            elif isinstance(blk, InceptionB):
                fcs = dict(b3=g("branch3x3"), d1=g("branch3x3dbl_1"), d2=g("branch3x3dbl_2"), d3=g("branch3x3dbl_3"))
            elif isinstance(blk, InceptionC):
                fcs = dict(g1=g("branch1x1", "branch7x7_1", "branch7x7dbl_1"), s2=g("branch7x7_2"), s3=g("branch7x7_3"),
                           d2=g("branch7x7dbl_2"), d3=g("branch7x7dbl_3"), d4=g("branch7x7dbl_4"), d5=g("branch7x7dbl_5"),
                           bp=g("branch_pool"))
            elif isinstance(blk, InceptionD):
                fcs = dict(g1=g("branch3x3_1", "branch7x7x3_1"), b2=g("branch3x3_2"), s2=g("branch7x7x3_2"),
                           s3=g("branch7x7x3_3"), s4=g("branch7x7x3_4"))
            else:
                fcs = dict(g1=g("branch1x1", "branch3x3_1", "branch3x3dbl_1"), a=g("branch3x3_2a"), b=g("branch3x3_2b"),
                           d2=g("branch3x3dbl_2"), da=g("branch3x3dbl_3a"), db=g("branch3x3dbl_3b"), bp=g("branch_pool"))
            self.blocks.append((name, type(blk).__name__, fcs))

    # each block: x (dense _Slice) -> dense output tensor.  Forward levels run as they are built; the backward levels list
    # the same operations from the outputs towards the input, grouped so that no two members of a level write the same
    # gradient elements (the members of a level run concurrently in one launch).
    @staticmethod
    def _block(tp, kind, f, x):
        H, W = x.H, x.W
        if kind == "InceptionA":
            n1, n5, nd = 64, 48, 64
            O = tp.new(n1 + f["b5"].cout + f["d3"].cout + f["bp"].cout, H, W)
            T = tp.scratch_for_group(n1, n5 + nd, H, W)
            U, P = tp.new(96, H, W), tp.new(x.C, H, W)
            g1 = tp.conv(f["g1"], x, _Slice(O, 0, n1), _Slice(T))
            pool = tp.avgpool(x, _Slice(P), 3, 1, 1)
            b5 = tp.conv(f["b5"], _Slice(T, 0, n5), _Slice(O, n1, 64))
            d2 = tp.conv(f["d2"], _Slice(T, n5, nd), _Slice(U))
            bp = tp.conv(f["bp"], _Slice(P), _Slice(O, n1 + 64 + 96, f["bp"].cout), relu_in=False)
            d3 = tp.conv(f["d3"], _Slice(U), _Slice(O, n1 + 64, 96))
            for level in ([g1, pool], [b5, d2, bp], [d3]):
                tp.run(level)
            tp.plan_backward([[bp, b5, d3], [d2], [pool, g1]])
            return O
        if kind == "InceptionB":
            oh, ow = f["b3"].out_hw(H, W)
            O = tp.new(384 + 96 + x.C, oh, ow)
            T, U = tp.new(64, H, W), tp.new(96, H, W)
            b3 = tp.conv(f["b3"], x, _Slice(O, 0, 384))
            d1 = tp.conv(f["d1"], x, _Slice(T))
            mp = tp.maxpool(x, _Slice(O, 480, x.C), 3, 2)
            d2 = tp.conv(f["d2"], _Slice(T), _Slice(U))
            d3 = tp.conv(f["d3"], _Slice(U), _Slice(O, 384, 96))
            for level in ([b3, d1, mp], [d2], [d3]):
                tp.run(level)
            tp.plan_backward([[mp, d3], [d2], [d1], [b3]])          # d1 and b3 both add into dX: separate levels
            return O
        if kind == "InceptionC":
            c7 = f["s2"].cin
            O = tp.new(768, H, W)
            T = tp.scratch_for_group(192, 2 * c7, H, W)
            V, W1, W2, W3 = (tp.new(c7, H, W) for _ in range(4))
            P = tp.new(x.C, H, W)
            g1 = tp.conv(f["g1"], x, _Slice(O, 0, 192), _Slice(T))
            pool = tp.avgpool(x, _Slice(P), 3, 1, 1)
            s2 = tp.conv(f["s2"], _Slice(T, 0, c7), _Slice(V))
            d2 = tp.conv(f["d2"], _Slice(T, c7, c7), _Slice(W1))
            bp = tp.conv(f["bp"], _Slice(P), _Slice(O, 576, 192), relu_in=False)
            s3 = tp.conv(f["s3"], _Slice(V), _Slice(O, 192, 192))
            d3 = tp.conv(f["d3"], _Slice(W1), _Slice(W2))
            d4 = tp.conv(f["d4"], _Slice(W2), _Slice(W3))
            d5 = tp.conv(f["d5"], _Slice(W3), _Slice(O, 384, 192))
            for level in ([g1, pool], [s2, d2, bp], [s3, d3], [d4], [d5]):
                tp.run(level)
            tp.plan_backward([[bp, s3, d5], [s2, d4], [d3], [d2], [pool, g1]])
            return O
        if kind == "InceptionD":
            oh, ow = f["b2"].out_hw(H, W)
            O = tp.new(320 + 192 + x.C, oh, ow)
            T, V1, V2 = tp.new(384, H, W), tp.new(192, H, W), tp.new(192, H, W)
            g1 = tp.conv(f["g1"], x, _Slice(T))
            mp = tp.maxpool(x, _Slice(O, 512, x.C), 3, 2)
            b2 = tp.conv(f["b2"], _Slice(T, 0, 192), _Slice(O, 0, 320))
            s2 = tp.conv(f["s2"], _Slice(T, 192, 192), _Slice(V1))
            s3 = tp.conv(f["s3"], _Slice(V1), _Slice(V2))
            s4 = tp.conv(f["s4"], _Slice(V2), _Slice(O, 320, 192))
            for level in ([g1, mp], [b2, s2], [s3], [s4]):
                tp.run(level)
            tp.plan_backward([[mp, b2, s4], [s3], [s2], [g1]])
            return O
        # InceptionE
        O = tp.new(2048, H, W)
        T = tp.scratch_for_group(320, 384 + 448, H, W)
        U, P = tp.new(384, H, W), tp.new(x.C, H, W)
        g1 = tp.conv(f["g1"], x, _Slice(O, 0, 320), _Slice(T))
        pool = tp.avgpool(x, _Slice(P), 3, 1, 1)
        a = tp.conv(f["a"], _Slice(T, 0, 384), _Slice(O, 320, 384))
        b = tp.conv(f["b"], _Slice(T, 0, 384), _Slice(O, 704, 384))
        d2 = tp.conv(f["d2"], _Slice(T, 384, 448), _Slice(U))
        bp = tp.conv(f["bp"], _Slice(P), _Slice(O, 1856, 192), relu_in=False)
        da = tp.conv(f["da"], _Slice(U), _Slice(O, 1088, 384))
        db = tp.conv(f["db"], _Slice(U), _Slice(O, 1472, 384))
        for level in ([g1, pool], [a, b, d2, bp], [da, db]):
            tp.run(level)
        tp.plan_backward([[bp, a, da], [b, db], [d2], [pool, g1]])    # a/b both add into dT[:, :384], da/db into dU
        return O

    def run_stem(self, x299):
        """Conv2d_1a .. the second max-pool -> (tape, (B,192,35,35) slice)"""
        tp = _Tape(x299.shape[0], x299.device)
        tp.x299 = x299
        st = self.stem
        cur = _Slice(x299)
        plan = (("Conv2d_1a_3x3", False, True), ("Conv2d_2a_3x3", True, False), ("Conv2d_2b_3x3", True, False), "pool",
                ("Conv2d_3b_1x1", True, False), ("Conv2d_4a_3x3", True, True), "pool")
        for step in plan:
            if step == "pool":
                y = _Slice(tp.new(cur.C, (cur.H - 3) // 2 + 1, (cur.W - 3) // 2 + 1))
                op = tp.maxpool(cur, y, 3, 2)
            else:
                name, relu_in, plain_bwd = step
                fc = st[name]
                oh, ow = fc.out_hw(cur.H, cur.W)
                y = _Slice(tp.new(fc.cout, oh, ow))
                # forward through the ordinary entry point (Winograd / streaming kernels where they apply); `plain` also
                # keeps the backward on it where that matters (first convolution: streaming kernel; 4a: Winograd)
                op = tp.conv(fc, cur, y, relu_in=relu_in, plain=True)
            tp.run([op])
            if op.kind == "conv" and not plain_bwd:
                op.plain = False
            tp.plan_backward([[op]])
            cur = y
        return tp, cur

    def forward(self, x299):
        """x299: (B,3,299,299) dense.  -> tape, features (B,768,17,17) = Mixed_6e output, Mixed_7c output (B,2048,8,8)"""
        tp, cur = self.run_stem(x299)
        feats = None
        for name, kind, fcs in self.blocks:
            cur = _Slice(self._block(tp, kind, fcs, cur))
            if name == "Mixed_6e":
                feats = cur.t
        tp.outs = (feats, cur.t)
        return tp, feats, cur.t

    @staticmethod
    def backward(tp, gfeat, glast):
        """gradients of the two outputs (None = zero) -> gradient of x299"""
        feats, last = tp.outs
        tp.seed(last, glast if glast is not None else torch.zeros_like(last))
        if gfeat is not None:
            tp.seed(feats, gfeat)
        tp.backward()
        return tp.grad_of(tp.x299)[0]


# ======================================================================================================================
# Panel trunk: the Mixed blocks on the packed-operand GEMM (hip: mogan_pk_group + mogan_panel_tail_group).
#
# FrozenTrunk above still splits both operands of every product into bf16 pieces inside the GEMM kernel, launch after launch,
# although the weights never change and every activation is read by several K-tiles of several convolutions.  Here
#   * every filter is packed ONCE at load, in both directions (forward: rows = Cout; data gradient: rows = Cin, eval-mode BN scale
#     folded), zero-padded where a channel slice is not a multiple of 32;
#   * activations and gradients travel as pixel panels (channels-last bf16 pieces); a tensor's channel slices start at multiples of
#     32, so concatenation is addressing (cg0) and a convolution reads its slice in place;
#   * one dependency level of a Mixed block = ONE grouped GEMM launch + ONE tail launch (K-split sum, BN affine + ReLU or ReLU
#     mask, gradient accumulation, fp32 copy, next panel);
#   * the pool branch avg_pool(3,1,1) -> 1x1 conv is evaluated as 1x1 conv -> avg_pool (both linear, they commute): the 1x1 joins
#     the block's first GEMM (one more row range) and the pool runs on 192 instead of 768 channels inside the tail; backward the
#     same way round (box filter of the slice's gradient, then the shared data-gradient GEMM);
#   * the four stride-2 convolutions run forward on the packed kernel; their data gradients (odd maps, 3x3 stride 2) and the two
#     max-pools keep the kernels FrozenTrunk uses, on the fp32 copies.
# Same arithmetic as FrozenTrunk up to fp32 reassociation (pool/conv order, K-split order); checked against it and the CPU
# restatement in tests/.
PANEL_TRUNK = True
# blocks a grouped GEMM launch aims at (K-split).  Measured in the B = 16 step: 256 / 128 / none 407 img/s, 512 404-405, 768 402 (alone
# on the GPU the launches are fastest at 512: 3.0 vs 3.5 ms per step at 128)
PANEL_TARGET = 256


def _up32(n):
    return (n + 31) // 32 * 32


class _PT:
    """(B, Cp, H, W): channel slices `sizes` laid out at multiples of 32; fp32 NCHW copy and / or pixel panel"""

    def __init__(self, B, dev, sizes, H, W, f32=True, panel=True):
        self.sizes, self.H, self.W, self.B = list(sizes), H, W, B
        self.off, c = [], 0
        for n in self.sizes:
            self.off.append(c)
            c += _up32(n)
        self.Cp = c
        self.f32 = torch.empty((B, c, H, W), dtype=torch.float32, device=dev) if f32 is True else (None if f32 is False else f32)
        self.panel = torch.empty(B * H * W * c * 6, dtype=torch.uint8, device=dev) if panel else None
        self.written = False                                  # gradient tensors: the fp32 copy holds a contribution already

    def sl(self, i):
        return _PS(self, self.off[i], self.sizes[i])

    def whole(self):
        return _PS(self, 0, self.Cp)


class _PS:
    """channels [c0, c0 + n) of a _PT"""
    __slots__ = ("t", "c0", "n")

    def __init__(self, t, c0, n):
        self.t, self.c0, self.n = t, c0, n

    @property
    def f32ptr(self):
        return self.t.f32.data_ptr() + 4 * self.c0 * self.t.H * self.t.W

    @property
    def bs(self):
        return self.t.Cp * self.t.H * self.t.W


class _PConv:
    """one convolution of the panel trunk (or a group of same-input 1x1 ones: rows concatenated): packed filters, folded BN"""

    def __init__(self, mods, dgrad=True):
        from ..hip.lib import call, load, stream_ptr
        mods = list(mods)
        m0 = mods[0].conv
        self.k, self.stride, self.pad = m0.kernel_size, m0.stride[0], m0.padding
        for m in mods[1:]:
            assert m.conv.kernel_size == self.k and m.conv.stride[0] == self.stride and m.conv.padding == self.pad
        self.couts = [m.conv.weight.shape[0] for m in mods]
        self.cin = m0.weight.shape[1]
        self.M, self.cin_p = sum(self.couts), _up32(self.cin)
        L = load()
        with torch.no_grad():
            w = torch.cat([m.conv.weight for m in mods], 0)
            folded = [m.folded() for m in mods]
            self.scale = torch.cat([f[0] for f in folded]).contiguous()
            self.shift = torch.cat([f[1] for f in folded]).contiguous()
            wf = torch.zeros((self.M, self.cin_p) + tuple(self.k), dtype=torch.float32, device=w.device)
            wf[:, :self.cin] = w
            self.wf = torch.empty(L.mogan_pk_weight_bytes(self.M, self.cin_p, self.k[0], self.k[1], 1, 0), dtype=torch.uint8,
                                  device=w.device)
            call("mogan_pk_weight_pack", wf.data_ptr(), self.wf.data_ptr(), self.M, self.cin_p, self.k[0], self.k[1], 1,
                 self.pad[0], self.pad[1], 0, stream_ptr())
            self.wb = (w * self.scale.view(-1, 1, 1, 1)).contiguous()                 # backward: BN scale folded
            self.wd, self.Mp = None, sum(_up32(c) for c in self.couts)
            if dgrad and self.stride == 1:
                wp = torch.zeros((self.Mp, self.cin) + tuple(self.k), dtype=torch.float32, device=w.device)
                r, rp = 0, 0
                for c in self.couts:
                    wp[rp:rp + c] = self.wb[r:r + c]
                    r, rp = r + c, rp + _up32(c)
                self.wd = torch.empty(L.mogan_pk_weight_bytes(self.Mp, self.cin, self.k[0], self.k[1], 1, 1), dtype=torch.uint8,
                                      device=w.device)
                call("mogan_pk_weight_pack", wp.data_ptr(), self.wd.data_ptr(), self.Mp, self.cin, self.k[0], self.k[1], 1,
                     self.pad[0], self.pad[1], 1, stream_ptr())
                self.wb = None
            torch.cuda.current_stream().synchronize()                                  # wf / wp die here
        self.device = w.device

    def out_hw(self, H, W):
        return ((H + 2 * self.pad[0] - self.k[0]) // self.stride + 1, (W + 2 * self.pad[1] - self.k[1]) // self.stride + 1)

    def rows(self, i):
        """(first row, rows) of member i of a grouped convolution"""
        return sum(self.couts[:i]), self.couts[i]


class _Raw:
    """result of one GEMM of a grouped launch: nsplit slabs of (B, M, H, W) fp32"""
    __slots__ = ("t", "M", "H", "W", "B", "nsplit")

    def __init__(self, B, M, H, W, nsplit, dev):
        self.t = torch.empty((nsplit, B, M, H, W), dtype=torch.float32, device=dev)
        self.M, self.H, self.W, self.B, self.nsplit = M, H, W, B, nsplit

    def rows(self, r0=0, n=None):
        return (self.t.data_ptr() + 4 * r0 * self.H * self.W, self.M * self.H * self.W, self.B * self.M * self.H * self.W, self.nsplit)


class _PTape:
    """launch helpers + the tensors the backward pass needs"""

    def __init__(self, B, dev):
        self.B, self.dev, self.blocks = B, dev, []
        self.g = {}

    def new(self, sizes, H, W, f32=True, panel=True):
        return _PT(self.B, self.dev, sizes, H, W, f32, panel)

    def grad(self, t, f32=True, panel=True):
        """gradient tensor of forward tensor t (same slices)"""
        g = self.g.get(id(t))
        if g is None:
            g = self.g[id(t)] = _PT(self.B, self.dev, t.sizes, t.H, t.W, f32, panel)
        return g

    # ---- grouped GEMM: members = [(conv, dgrad, K-range slice, (outH, outW))] -> [_Raw]
    def gemm(self, members):
        import ctypes
        from ..hip.lib import PkArgs, call, stream_ptr
        tiles, geo = 0, []
        for fc, dgrad, x, (oh, ow) in members:
            M = fc.cin if dgrad else fc.M
            Cp = _up32(x.n)
            assert x.c0 % 32 == 0 and Cp == (fc.Mp if dgrad else fc.cin_p), (x.c0, x.n, fc.Mp, fc.cin_p, dgrad)
            tiles += -(-M // 128) * -(-(self.B * oh * ow) // 64)
            geo.append((M, Cp, fc.k[0] * fc.k[1] * Cp // 32))
        want = max(1, -(-PANEL_TARGET // tiles)) if tiles < PANEL_TARGET else 1
        arr = (PkArgs * len(members))()
        raws = []
        for i, ((fc, dgrad, x, (oh, ow)), (M, Cp, ntile)) in enumerate(zip(members, geo)):
            ns = max(1, min(want, ntile // 6))
            kt_per = -(-(-(-ntile // ns)) // 6) * 6
            ns = -(-ntile // kt_per)
            raw = _Raw(self.B, M, oh, ow, ns, self.dev)
            raws.append(raw)
            a = arr[i]
            a.wpk, a.panel, a.raw = (fc.wd if dgrad else fc.wf).data_ptr(), x.t.panel.data_ptr(), raw.t.data_ptr()
            a.B, a.M, a.Cp, a.CGp, a.cg0 = self.B, M, Cp, x.t.Cp // 32, x.c0 // 32
            a.PH, a.PW, a.outH, a.outW = x.t.H, x.t.W, oh, ow
            a.KH, a.KW, a.stride, a.ph, a.pw, a.dgrad, a.nsplit = fc.k[0], fc.k[1], fc.stride, fc.pad[0], fc.pad[1], 1 if dgrad else 0, ns
        call("mogan_pk_group", len(members), ctypes.cast(arr, ctypes.c_void_p), stream_ptr())
        for i, raw in enumerate(raws):
            assert 1 <= arr[i].nsplit <= raw.nsplit, (arr[i].nsplit, raw.nsplit)
            raw.nsplit = arr[i].nsplit                                                 # slabs actually written
        return raws

    # ---- tail: members = dicts(n, H, W, srcs=[(ptr, bs, slab, nsplit)], dst=_PS|None (fp32), panel=_PS|None, ...)
    def tail(self, members):
        import ctypes
        from ..hip.lib import TailArgs, call, stream_ptr
        for i0 in range(0, len(members), 8):
            part = members[i0:i0 + 8]
            arr = (TailArgs * len(part))()
            for a, m in zip(arr, part):
                for j, (ptr, bs, slab, ns) in enumerate(m["srcs"]):
                    a.src[j], a.src_bs[j], a.src_slab[j], a.src_nsplit[j] = ptr, bs, slab, ns
                a.nsrc = len(m["srcs"])
                out = m["out"]
                a.B, a.n, a.H, a.W = self.B, out.n if "n" not in m else m["n"], out.t.H, out.t.W
                if m.get("add") is not None:
                    a.add, a.add_bs = m["add"].f32ptr, m["add"].bs
                if m.get("mask") is not None:
                    a.mask, a.mask_bs = m["mask"].f32ptr, m["mask"].bs
                if m.get("scale") is not None:
                    a.scale, a.shift = m["scale"], m["shift"]
                a.relu, a.box = int(m.get("relu", 0)), int(m.get("box", 0))
                if m.get("f32", True) and out.t.f32 is not None:
                    a.dst, a.dst_bs = out.f32ptr, out.bs
                if m.get("panel", True) and out.t.panel is not None:
                    a.panel, a.CGp, a.cg0 = out.t.panel.data_ptr(), out.t.Cp // 32, out.c0 // 32
            call("mogan_panel_tail_group", len(part), ctypes.cast(arr, ctypes.c_void_p), stream_ptr())


def _f32src(s):
    """a fp32 slice as a tail source"""
    return (s.f32ptr, s.bs, 0, 1)


class PanelTrunk(FrozenTrunk):
    """Stem as FrozenTrunk (wide maps, few channels: the streaming / Winograd kernels); Mixed_5b .. 7c on pixel panels."""

    def __init__(self, enc):
        self.stem = {n: _FrozenConv([getattr(enc, n)]) for n in ("Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3",
                                                               "Conv2d_3b_1x1", "Conv2d_4a_3x3")}
        self.blocks = []
        for name, _ in TRUNK[5:]:
            blk = getattr(enc, name)
            g = lambda *names, **kw: _PConv([getattr(blk, n) for n in names], **kw)
            if isinstance(blk, InceptionA):
                fcs = dict(g1=g("branch1x1", "branch5x5_1", "branch3x3dbl_1", "branch_pool"), b5=g("branch5x5_2"),
                           d2=g("branch3x3dbl_2"), d3=g("branch3x3dbl_3"))
            elif isinstance(blk, InceptionB):
                fcs = dict(b3=g("branch3x3"), d1=g("branch3x3dbl_1"), d2=g("branch3x3dbl_2"), d3=g("branch3x3dbl_3"))
            elif isinstance(blk, InceptionC):
                fcs = dict(g1=g("branch1x1", "branch7x7_1", "branch7x7dbl_1", "branch_pool"), s2=g("branch7x7_2"),
                           s3=g("branch7x7_3"), d2=g("branch7x7dbl_2"), d3=g("branch7x7dbl_3"), d4=g("branch7x7dbl_4"),
                           d5=g("branch7x7dbl_5"))
            elif isinstance(blk, InceptionD):
                fcs = dict(g1=g("branch3x3_1", "branch7x7x3_1"), b2=g("branch3x3_2"), s2=g("branch7x7x3_2"),
                           s3=g("branch7x7x3_3"), s4=g("branch7x7x3_4"))
            else:
                fcs = dict(g1=g("branch1x1", "branch3x3_1", "branch3x3dbl_1", "branch_pool"), a=g("branch3x3_2a"),
                           b=g("branch3x3_2b"), d2=g("branch3x3dbl_2"), da=g("branch3x3dbl_3a"), db=g("branch3x3dbl_3b"))
            self.blocks.append((name, type(blk).__name__, fcs))

    # ---- tail members -------------------------------------------------------------------------------------------------
    @staticmethod
    def _fwd(raw, fc, i, out, box=0):
        """rows of member i of (grouped) convolution fc -> BN affine + ReLU -> out (fp32 + panel)"""
        r0, n = fc.rows(i)
        return dict(srcs=[raw.rows(r0)], out=out, n=n, scale=fc.scale.data_ptr() + 4 * r0, shift=fc.shift.data_ptr() + 4 * r0, relu=1,
                    box=box)

    @staticmethod
    def _bwd(raws, out, mask, add=None, f32=False, panel=True):
        """sum of data-gradient GEMM results (+ what the fp32 gradient holds already) -> ReLU mask of the input -> out"""
        return dict(srcs=[r.rows(0) for r in raws], out=out, mask=mask, add=add, f32=f32, panel=panel)

    @staticmethod
    def _final(raws, dX, xin):
        """the last contribution to a block input's gradient: + what the fp32 copy holds (seed, max-pool, stride-2 data gradients),
        ReLU mask of the input -> fp32 + panel"""
        m = dict(srcs=[r.rows(0) for r in raws], out=dX.whole(), mask=xin, add=dX.whole() if dX.written else None, f32=True)
        dX.written = True
        return m

    # ---- blocks: forward runs at once and returns (O, backward closure) --------------------------------------------------
    def _block(self, tp, kind, f, X):
        B, H, W = tp.B, X.H, X.W
        hw = (H, W)
        xin = X.whole()
        if kind in ("InceptionA", "InceptionC", "InceptionE"):
            g1 = f["g1"]
            n1, npool = g1.couts[0], g1.couts[3]
            G = tp.new(g1.couts[1:3], H, W)
            if kind == "InceptionA":
                O = tp.new([n1, 64, 96, npool], H, W)
                ipool = 3
            elif kind == "InceptionC":
                O = tp.new([192, 192, 192, 192], H, W)
                ipool = 3
            else:
                O = tp.new([320, 384, 384, 384, 384, 192], H, W)
                ipool = 5
            (r,) = tp.gemm([(g1, 0, xin, hw)])
            tp.tail([self._fwd(r, g1, 0, O.sl(0)), self._fwd(r, g1, 1, G.sl(0)), self._fwd(r, g1, 2, G.sl(1)),
                     self._fwd(r, g1, 3, O.sl(ipool), box=1)])
            if kind == "InceptionA":
                U = tp.new([96], H, W)
                ra, rb = tp.gemm([(f["b5"], 0, G.sl(0), hw), (f["d2"], 0, G.sl(1), hw)])
                tp.tail([self._fwd(ra, f["b5"], 0, O.sl(1)), self._fwd(rb, f["d2"], 0, U.sl(0))])
                (rc,) = tp.gemm([(f["d3"], 0, U.sl(0), hw)])
                tp.tail([self._fwd(rc, f["d3"], 0, O.sl(2))])

                def backward(dO, dX):
                    DG = tp.new(g1.couts, H, W, f32=False)
                    dU = tp.new([96], H, W, f32=False)
                    ra, rc = tp.gemm([(f["b5"], 1, dO.sl(1), hw), (f["d3"], 1, dO.sl(2), hw)])
                    tp.tail([self._bwd([ra], DG.sl(1), G.sl(0)), self._bwd([rc], dU.sl(0), U.sl(0)),
                             dict(srcs=[_f32src(dO.sl(0))], out=DG.sl(0)), dict(srcs=[_f32src(dO.sl(3))], out=DG.sl(3), box=1)])
                    (rb,) = tp.gemm([(f["d2"], 1, dU.sl(0), hw)])
                    tp.tail([self._bwd([rb], DG.sl(2), G.sl(1))])
                    (rg,) = tp.gemm([(g1, 1, DG.whole(), hw)])
                    tp.tail([self._final([rg], dX, xin)])
            elif kind == "InceptionC":
                c7 = g1.couts[1]
                V, W1, W2, W3 = (tp.new([c7], H, W) for _ in range(4))
                ra, rb = tp.gemm([(f["s2"], 0, G.sl(0), hw), (f["d2"], 0, G.sl(1), hw)])
                tp.tail([self._fwd(ra, f["s2"], 0, V.sl(0)), self._fwd(rb, f["d2"], 0, W1.sl(0))])
                ra, rb = tp.gemm([(f["s3"], 0, V.sl(0), hw), (f["d3"], 0, W1.sl(0), hw)])
                tp.tail([self._fwd(ra, f["s3"], 0, O.sl(1)), self._fwd(rb, f["d3"], 0, W2.sl(0))])
                (rb,) = tp.gemm([(f["d4"], 0, W2.sl(0), hw)])
                tp.tail([self._fwd(rb, f["d4"], 0, W3.sl(0))])
                (rb,) = tp.gemm([(f["d5"], 0, W3.sl(0), hw)])
                tp.tail([self._fwd(rb, f["d5"], 0, O.sl(2))])

                def backward(dO, dX):
                    DG = tp.new(g1.couts, H, W, f32=False)
                    dV, dW1, dW2, dW3 = (tp.new([c7], H, W, f32=False) for _ in range(4))
                    ra, rb = tp.gemm([(f["s3"], 1, dO.sl(1), hw), (f["d5"], 1, dO.sl(2), hw)])
                    tp.tail([self._bwd([ra], dV.sl(0), V.sl(0)), self._bwd([rb], dW3.sl(0), W3.sl(0)),
                             dict(srcs=[_f32src(dO.sl(0))], out=DG.sl(0)), dict(srcs=[_f32src(dO.sl(3))], out=DG.sl(3), box=1)])
                    ra, rb = tp.gemm([(f["s2"], 1, dV.sl(0), hw), (f["d4"], 1, dW3.sl(0), hw)])
                    tp.tail([self._bwd([ra], DG.sl(1), G.sl(0)), self._bwd([rb], dW2.sl(0), W2.sl(0))])
                    (rb,) = tp.gemm([(f["d3"], 1, dW2.sl(0), hw)])
                    tp.tail([self._bwd([rb], dW1.sl(0), W1.sl(0))])
                    (rb,) = tp.gemm([(f["d2"], 1, dW1.sl(0), hw)])
                    tp.tail([self._bwd([rb], DG.sl(2), G.sl(1))])
                    (rg,) = tp.gemm([(g1, 1, DG.whole(), hw)])
                    tp.tail([self._final([rg], dX, xin)])
            else:
                U = tp.new([384], H, W)
                ra, rb, rc = tp.gemm([(f["a"], 0, G.sl(0), hw), (f["b"], 0, G.sl(0), hw), (f["d2"], 0, G.sl(1), hw)])
                tp.tail([self._fwd(ra, f["a"], 0, O.sl(1)), self._fwd(rb, f["b"], 0, O.sl(2)), self._fwd(rc, f["d2"], 0, U.sl(0))])
                ra, rb = tp.gemm([(f["da"], 0, U.sl(0), hw), (f["db"], 0, U.sl(0), hw)])
                tp.tail([self._fwd(ra, f["da"], 0, O.sl(3)), self._fwd(rb, f["db"], 0, O.sl(4))])

                def backward(dO, dX):
                    DG = tp.new(g1.couts, H, W, f32=False)
                    dU = tp.new([384], H, W, f32=False)
                    ra, rb, rc, rd = tp.gemm([(f["a"], 1, dO.sl(1), hw), (f["b"], 1, dO.sl(2), hw), (f["da"], 1, dO.sl(3), hw),
                                              (f["db"], 1, dO.sl(4), hw)])
                    tp.tail([self._bwd([ra, rb], DG.sl(1), G.sl(0)), self._bwd([rc, rd], dU.sl(0), U.sl(0)),
                             dict(srcs=[_f32src(dO.sl(0))], out=DG.sl(0)), dict(srcs=[_f32src(dO.sl(5))], out=DG.sl(3), box=1)])
                    (rb,) = tp.gemm([(f["d2"], 1, dU.sl(0), hw)])
                    tp.tail([self._bwd([rb], DG.sl(2), G.sl(1))])
                    (rg,) = tp.gemm([(g1, 1, DG.whole(), hw)])
                    tp.tail([self._final([rg], dX, xin)])
            return O, backward
        import ctypes
        from ..hip.lib import ConvDgradArgs, call, stream_ptr, workspace

        def old_dgrads(items):
            """stride-2 data gradients on the fp32 copies: items = [(conv, dy slice, dx slice, ReLU-mask slice, accumulate)]"""
            arr = (ConvDgradArgs * len(items))(*[
                ConvDgradArgs(dy.f32ptr, dy.bs, fc.wb.data_ptr(), dx.f32ptr, dx.bs, mk.f32ptr, mk.bs, 1 if acc else 0, B, fc.cin,
                              mk.t.H, mk.t.W, fc.M, fc.k[0], fc.k[1], fc.stride, fc.pad[0], fc.pad[1])
                for fc, dy, dx, mk, acc in items])
            wsp, wsn = workspace(tp.dev)
            call("mogan_conv2d_dgrad_group", len(items), ctypes.cast(arr, ctypes.c_void_p), wsp, wsn, stream_ptr())

        def maxpool_fwd(dst):
            idx = torch.empty((B, X.Cp, dst.t.H, dst.t.W), dtype=torch.uint8, device=tp.dev)
            call("mogan_maxpool_fwd_ex", X.f32.data_ptr(), dst.f32ptr, dst.bs, idx.data_ptr(), B, X.Cp, H, W, 3, 2, stream_ptr())
            return idx

        def maxpool_bwd(idx, dy, dX):
            call("mogan_maxpool_bwd_ex", idx.data_ptr(), dy.f32ptr, dy.bs, dX.f32.data_ptr(), X.f32.data_ptr(), 1 if dX.written else 0,
                 B, X.Cp, H, W, 3, 2, stream_ptr())
            dX.written = True

        if kind == "InceptionB":
            ohw = f["b3"].out_hw(H, W)
            O = tp.new([384, 96, X.Cp], *ohw)
            T, U = tp.new([64], H, W), tp.new([96], H, W)
            idx = maxpool_fwd(O.sl(2))
            ra, rb = tp.gemm([(f["b3"], 0, xin, ohw), (f["d1"], 0, xin, hw)])
            tp.tail([self._fwd(ra, f["b3"], 0, O.sl(0)), self._fwd(rb, f["d1"], 0, T.sl(0)),
                     dict(srcs=[_f32src(O.sl(2))], out=O.sl(2), f32=False)])
            (rb,) = tp.gemm([(f["d2"], 0, T.sl(0), hw)])
            tp.tail([self._fwd(rb, f["d2"], 0, U.sl(0))])
            (rb,) = tp.gemm([(f["d3"], 0, U.sl(0), ohw)])
            tp.tail([self._fwd(rb, f["d3"], 0, O.sl(1))])

            def backward(dO, dX):
                dU, dT = tp.new([96], H, W), tp.new([64], H, W, f32=False)
                maxpool_bwd(idx, dO.sl(2), dX)
                old_dgrads([(f["b3"], dO.sl(0), dX.whole(), xin, True), (f["d3"], dO.sl(1), dU.sl(0), U.sl(0), False)])
                tp.tail([dict(srcs=[_f32src(dU.sl(0))], out=dU.sl(0), f32=False)])
                (rb,) = tp.gemm([(f["d2"], 1, dU.sl(0), hw)])
                tp.tail([self._bwd([rb], dT.sl(0), T.sl(0))])
                (rb,) = tp.gemm([(f["d1"], 1, dT.sl(0), hw)])
                tp.tail([self._final([rb], dX, xin)])
            return O, backward
        # InceptionD
        g1 = f["g1"]
        ohw = f["b2"].out_hw(H, W)
        O = tp.new([320, 192, X.Cp], *ohw)
        G, V1, V2 = tp.new([192, 192], H, W), tp.new([192], H, W), tp.new([192], H, W)
        idx = maxpool_fwd(O.sl(2))
        (r,) = tp.gemm([(g1, 0, xin, hw)])
        tp.tail([self._fwd(r, g1, 0, G.sl(0)), self._fwd(r, g1, 1, G.sl(1)), dict(srcs=[_f32src(O.sl(2))], out=O.sl(2), f32=False)])
        ra, rb = tp.gemm([(f["b2"], 0, G.sl(0), ohw), (f["s2"], 0, G.sl(1), hw)])
        tp.tail([self._fwd(ra, f["b2"], 0, O.sl(0)), self._fwd(rb, f["s2"], 0, V1.sl(0))])
        (rb,) = tp.gemm([(f["s3"], 0, V1.sl(0), hw)])
        tp.tail([self._fwd(rb, f["s3"], 0, V2.sl(0))])
        (rb,) = tp.gemm([(f["s4"], 0, V2.sl(0), ohw)])
        tp.tail([self._fwd(rb, f["s4"], 0, O.sl(1))])

        def backward(dO, dX):
            DG, dV2, dV1 = tp.new([192, 192], H, W), tp.new([192], H, W), tp.new([192], H, W, f32=False)
            maxpool_bwd(idx, dO.sl(2), dX)
            old_dgrads([(f["b2"], dO.sl(0), DG.sl(0), G.sl(0), False), (f["s4"], dO.sl(1), dV2.sl(0), V2.sl(0), False)])
            tp.tail([dict(srcs=[_f32src(DG.sl(0))], out=DG.sl(0), f32=False), dict(srcs=[_f32src(dV2.sl(0))], out=dV2.sl(0), f32=False)])
            (rb,) = tp.gemm([(f["s3"], 1, dV2.sl(0), hw)])
            tp.tail([self._bwd([rb], dV1.sl(0), V1.sl(0))])
            (rb,) = tp.gemm([(f["s2"], 1, dV1.sl(0), hw)])
            tp.tail([self._bwd([rb], DG.sl(1), G.sl(1))])
            (rg,) = tp.gemm([(g1, 1, DG.whole(), hw)])
            tp.tail([self._final([rg], dX, xin)])
        return O, backward

    def forward(self, x299):
        """x299: (B,3,299,299) dense.  -> (stem tape, panel tape), features (B,768,17,17) = Mixed_6e output, Mixed_7c output"""
        tp, cur = self.run_stem(x299)
        pt = _PTape(x299.shape[0], x299.device)
        X = _PT(pt.B, pt.dev, [cur.C], cur.H, cur.W, f32=cur.t)
        pt.tail([dict(srcs=[_f32src(X.sl(0))], out=X.sl(0), f32=False)])
        pt.stem_out, pt.chain = cur.t, []
        feats = None
        for name, kind, fcs in self.blocks:
            O, bwd = self._block(pt, kind, fcs, X)
            pt.chain.append((X, O, bwd))
            X = O
            if name == "Mixed_6e":
                feats = O
        pt.feats, pt.last = feats, X
        return (tp, pt), feats.f32, X.f32

    @staticmethod
    def backward(tapes, gfeat, glast):
        """gradients of the two outputs (None = zero) -> gradient of x299"""
        tp, pt = tapes
        for out, g in ((pt.last, glast), (pt.feats, gfeat)):
            d = pt.grad(out)
            if g is None:
                d.f32.zero_()
                if out is pt.last:
                    d.panel.zero_()
            else:
                g = g.contiguous()
                gs = (g.data_ptr(), g.stride(0), 0, 1)
                # the seed is the gradient's first contribution: ReLU mask of the output; Mixed_7c's has no other one -> panel too
                pt.tail([dict(srcs=[gs], out=d.whole(), mask=out.whole(), panel=out is pt.last)])
            d.written = True
        for X, O, bwd in reversed(pt.chain):
            dO = pt.grad(O)
            if X.f32 is pt.stem_out:                         # the first block: its input gradient goes on into the stem's tape
                g, written = tp.grad_of(pt.stem_out)
                written.add((0, X.Cp))
                dX = _PT(pt.B, pt.dev, X.sizes, X.H, X.W, f32=g, panel=False)
            else:
                dX = pt.grad(X)
            bwd(dO, dX)
        tp.backward()
        return tp.grad_of(tp.x299)[0]


class _FrozenTrunkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x299, trunk):
        xc = x299.contiguous()
        tp, feats, last = trunk.forward(xc)
        ctx.tape, ctx.trunk = tp, trunk
        return feats, last

    @staticmethod
    def backward(ctx, gfeat, glast):
        g = ctx.trunk.backward(ctx.tape, gfeat, glast)
        ctx.tape = None
        return g, None


def frozen_trunk(enc, x299):
    """(features 768x17x17, Mixed_7c output 2048x8x8) of the frozen eval-mode trunk of `enc` (a CNN_ENCODER)."""
    ft = getattr(enc, "_frozen_trunk", None)
    key = _versions(*[t for t in list(enc.parameters()) + list(enc.buffers()) if t.is_floating_point()])
    cls = PanelTrunk if PANEL_TRUNK else FrozenTrunk
    if ft is None or type(ft) is not cls or getattr(enc, "_frozen_trunk_key", None) != key \
            or ft.stem["Conv2d_1a_3x3"].w.device != x299.device:
        ft = enc._frozen_trunk = cls(enc)
        enc._frozen_trunk_key = key
    return _FrozenTrunkFn.apply(x299, ft)
