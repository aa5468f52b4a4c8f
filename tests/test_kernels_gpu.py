"""-m gpu: every HIP entry point (through the C ABI, via hip/ops.py) against the CPU oracle's
primitives (oracle/attngan_oracle.py, i.e. the torch-CPU ops the reference composes), evaluated in
fp64 so that a kernel more accurate than torch-fp32 is not penalised.

Stated tolerances (fp32 tensors and accumulators; products on the bf16 MFMA pipe from exact 3-piece splits, csrc/mogan_mma.h):
  conv / bmm outputs      rel-L2 <= 2e-6,  max-abs <= 1e-5 * max|ref|*sqrt(K)/16 (see _check)
  BN / activations / STN / attention / softmax / pooling   max-abs <= 2e-5 (values are O(1))
  Adam                    max-abs <= 1e-6 on O(1) parameters after three steps
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import ROOT, det_array, load_pkg, max_abs, rel_l2
from oracle import attngan_oracle as O

load_pkg()
from mogan_amd.hip import lib, ops  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(name, shape, scale=1.0, shift=0.0):
    return torch.from_numpy(det_array(name, shape, scale, shift))


def _check(got, ref, rtol=2e-6, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    denom = ref.norm().item() + 1e-30
    rel = (got - ref).norm().item() / denom
    mx = (got - ref).abs().max().item()
    assert rel <= rtol, "%s: rel-L2 %.3e > %.1e (max-abs %.3e)" % (what, rel, rtol, mx)


CONV_CASES = [
    # B, Cin, H, W, Cout, k (kh,kw), stride, pad (ph,pw), up
    (2, 8, 8, 8, 16, (3, 3), 1, (1, 1), 0),
    (2, 8, 8, 8, 16, (3, 3), 1, (1, 1), 1),        # upBlock: fused nearest x2
    (3, 5, 9, 7, 7, (3, 3), 1, (1, 1), 1),         # ragged everything
    (2, 6, 16, 16, 12, (4, 4), 2, (1, 1), 0),      # D down conv
    (2, 84, 16, 16, 24, (4, 4), 1, (1, 1), 0),     # D_NET64 local conv -> 15x15
    (3, 100, 16, 16, 50, (3, 3), 2, (1, 1), 0),    # BBOX_NET (3x3 s2)
    (2, 12, 4, 4, 1, (4, 4), 4, (0, 0), 0),        # logits conv 4x4 s4 (full-map dot kernels)
    (5, 70, 4, 4, 3, (4, 4), 4, (0, 0), 0),        # the same with 3 outputs, K = 1120 (not a multiple of 256)
    (4, 20, 5, 1, 6, (1, 1), 1, (0, 0), 0),        # conv_context (1x1 on (B,cdf,T,1))
    (2, 3, 32, 32, 96, (4, 4), 2, (1, 1), 0),      # first D conv (Cin=3)
    (2, 48, 16, 16, 3, (3, 3), 1, (1, 1), 0),      # img head (Cout=3)
    (3, 13, 37, 70, 3, (3, 3), 1, (1, 1), 0),      # img head, ragged sizes (direct small-channel kernels)
    (2, 20, 13, 128, 3, (3, 3), 1, (1, 1), 0),     # img head on 128-pixel rows: four pixels per thread (sc_fwd3x3_w4), ragged height / channels
    (1, 8, 16, 256, 4, (3, 3), 1, (1, 1), 0),      # the same with two tiles per row, four output channels, one channel chunk
    (2, 17, 9, 128, 1, (3, 3), 1, (1, 1), 0),      # one output channel, 17 input channels (a chunk of one)
    (2, 20, 64, 64, 1, (3, 3), 1, (1, 1), 0),      # multi-mnist img head (Cout=1)
    (2, 3, 64, 96, 40, (4, 4), 2, (1, 1), 0),      # first D conv, non-square (dgrad = 2x2-block kernel)
    (2, 1, 32, 32, 24, (4, 4), 2, (1, 1), 0),      # multi-mnist first D conv (Cin=1)
    (2, 3, 37, 45, 20, (3, 3), 2, (0, 0), 0),      # Inception Conv2d_1a (3 -> 32, 3x3 s2 valid, odd sizes): streaming dgrad
    (2, 10, 17, 17, 12, (1, 7), 1, (0, 3), 0),     # Inception 1x7
    (2, 10, 17, 17, 12, (7, 1), 1, (3, 0), 0),     # Inception 7x1
    (2, 6, 35, 35, 8, (3, 3), 2, (0, 0), 0),       # Inception 3x3 s2 valid (odd size)
    (2, 6, 12, 12, 8, (5, 5), 1, (2, 2), 0),       # Inception 5x5
    (1, 96, 32, 32, 96, (3, 3), 1, (1, 1), 0),     # 96-wide tile config
    (2, 160, 8, 8, 130, (3, 3), 1, (1, 1), 0),     # 128x128 tiles with ragged edges + long K
    # fused Winograd F(2x2,3x3) (3x3 s1 p1, Cin % 16 == 0, >= 64 output channels, H % 4 == 0, W % 32 == 0): fwd and dgrad
    (2, 32, 8, 32, 64, (3, 3), 1, (1, 1), 0),      # one tile column, two chunks
    (3, 48, 12, 64, 100, (3, 3), 1, (1, 1), 0),    # ragged M (100 = 96 + 4), odd chunk pairs, image borders everywhere
    (1, 96, 32, 32, 192, (3, 3), 1, (1, 1), 0),    # ResBlock widths
    (2, 64, 16, 96, 64, (3, 3), 1, (1, 1), 0),     # dgrad also Winograd (Cout % 16 == 0, Cin >= 64)
    (2, 32, 30, 61, 64, (3, 3), 1, (1, 1), 0),     # ragged grid (odd width: scalar stores), pad 1
    (2, 80, 29, 63, 96, (3, 3), 1, (0, 0), 0),     # valid convolution (Inception 4a): forward pad 0, data gradient pad 2
    # fused Winograd F(2x2,2x2) for 4x4 s2 p1 (Cin % 8 == 0, >= 64 in / 96 out channels, H, W % 4 == 0): forward
    (4, 72, 8, 8, 130, (4, 4), 2, (1, 1), 0),      # 4x4 outputs: a block spans 8 images; ragged M (130 = 128 + 2)
    (2, 64, 16, 24, 100, (4, 4), 2, (1, 1), 0),    # non-square, partly filled tile block, M < 128
    (1, 128, 64, 64, 128, (4, 4), 2, (1, 1), 0),   # several tile blocks of one image
    (16, 256, 8, 8, 192, (4, 4), 2, (1, 1), 0),    # few tiles, long K: the K range is split (partial slabs + reduce)
    (3, 104, 12, 20, 64, (4, 4), 2, (1, 1), 0),    # data gradient through it (Cout % 32 == 0, Cin >= 96): ragged M = 104, non-square
    # shapes that take the direct (halo-tile) kernel when no tile config is forced (>= 64 channels each side)
    (2, 64, 32, 32, 72, (3, 3), 1, (1, 1), 0),     # 3x3 s1, Cw=32, ragged M (forward: Winograd; dgrad: direct, 72 % 16 != 0)
    (2, 72, 32, 32, 64, (3, 3), 1, (1, 1), 0),     # Cin % 16 != 0: forward stays on the direct kernel (Cw=32, 16-byte halo loads)
    (2, 64, 16, 16, 64, (3, 3), 1, (1, 1), 1),     # upBlock: 16x16 -> 32x32
    (3, 64, 16, 16, 100, (3, 3), 1, (1, 1), 0),    # Cw=16, R=8; 96-wide M tile + ragged M
    (1, 64, 64, 128, 64, (3, 3), 1, (1, 1), 0),    # non-square
    (2, 256, 16, 16, 64, (3, 3), 1, (1, 1), 0),    # split over channel chunks
    (2, 64, 64, 64, 72, (4, 4), 2, (1, 1), 0),     # 4x4 s2 -> 32x32; dgrad = four 2x2 parity convs in one launch
    (2, 64, 32, 32, 256, (4, 4), 2, (1, 1), 0),    # 4x4 s2 -> 16x16; parity dgrad with channel split
    # round 5: the pre-split direct kernel (csrc/mogan_dconv2.hip; 3x3 s1 and the 2x2 parity classes of a 4x4 s2 data gradient on
    # grids of 8 x 32 tiles, channels % 16 == 0); 3x3 reaches it where Winograd declines or with force (-2, 0)
    (2, 96, 32, 64, 160, (3, 3), 1, (1, 1), 0),    # 96-row channel blocks, ragged M = 160 (guarded stores), dgrad: 160 -> 96
    (1, 32, 8, 32, 64, (3, 3), 1, (1, 1), 0),      # one tile, 64-row block; dgrad declined (32 output channels)
    (1, 16, 8, 32, 128, (3, 3), 1, (1, 1), 0),     # 128-row block, one 16-channel stage
    (1, 256, 8, 32, 64, (3, 3), 1, (1, 1), 0),     # one tile, 16 stages: K split over the stages + reduce
    (3, 48, 16, 32, 80, (3, 3), 1, (1, 1), 0),     # several tiles per persistent block across images
    (2, 96, 32, 64, 64, (4, 4), 2, (1, 1), 0),     # data gradient: four 2x2 parity classes, two 16-channel sub-chunks per stage
    (2, 128, 16, 64, 16, (4, 4), 2, (1, 1), 0),    # data gradient with 16 input channels of dY (one stage), 128-row blocks
    # 16 x 16 spatial tiles (maps with 16-pixel rows) and the 4x4 s2 FORWARD as a 2x2 filter over the space-to-depth image
    (2, 96, 64, 128, 192, (4, 4), 2, (1, 1), 0),   # forward: 8 x 32 tiles, 96-row blocks, 12 stages of 8 channels
    (2, 192, 32, 32, 96, (4, 4), 2, (1, 1), 0),    # forward and data gradient on 16 x 16 tiles, K split
    (3, 24, 32, 64, 100, (4, 4), 2, (1, 1), 0),    # forward: ragged M = 100, 24 channels (3 stages)
    (1, 8, 64, 32, 64, (4, 4), 2, (1, 1), 0),      # forward: one stage, two tiles of 16 x 16
    (2, 64, 16, 16, 128, (3, 3), 1, (1, 1), 0),    # 3x3 on one 16 x 16 tile per image
]


def _conv_ref(x, w, stride, pad, up):
    x = x.double()
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv2d(x, w.double(), None, stride, pad)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("force", [(-1, 0), (-2, 0), (0, 3), (1, 1), (2, 2), (3, 1), (4, 5), (5, 2), (6, 3)])
def test_conv2d_fwd_dgrad_wgrad(case, force):
    B, Cin, H, W, Cout, k, s, pad, up = case
    # (-1, 0): default dispatch (small-channel / Winograd F(2,3) / direct / implicit GEMM by shape); (-2, 0): the same without the
    # Winograd kernels (3x3 shapes on the direct kernels, forward / data gradient / weight gradient); the others force an
    # implicit-GEMM tile config and split
    lib.load().mogan_gemm_debug_force(force[0], force[1])
    try:
        x = T("cx%s" % (case,), (B, Cin, H, W)).requires_grad_(True)
        w = T("cw%s" % (case,), (Cout, Cin) + k, 0.2).requires_grad_(True)
        ref = _conv_ref(x, w, s, pad, up)
        g = T("cg%s" % (case,), ref.shape)
        ref.backward(g.double())
        xd = x.detach().to(DEV).requires_grad_(True)
        wd = w.detach().to(DEV).requires_grad_(True)
        y = ops.conv2d(xd, wd, None, s, pad, bool(up))
        y.backward(g.to(DEV))
        torch.cuda.synchronize()
        _check(y, ref, what="fwd")
        _check(xd.grad, x.grad, what="dgrad")
        _check(wd.grad, w.grad, what="wgrad")
    finally:
        lib.load().mogan_gemm_debug_force(-1, 0)


UP_CASES = [(2, 8, 8, 8, 16), (3, 5, 9, 7, 7), (2, 64, 16, 16, 64), (2, 96, 32, 32, 96), (1, 72, 64, 64, 100),
            (2, 128, 4, 4, 192)]


@pytest.mark.parametrize("case", UP_CASES)
@pytest.mark.parametrize("as_k4", [True, False])
def test_upsample_conv3x3_both_formulations(case, as_k4, monkeypatch):
    """nearest x2 + conv3x3(p1): the transposed 4x4-s2 evaluation (mogan_upconv3x3_*, K = T w T^t) and the
    fused-upsample 3x3 kernels against the fp64 reference -- forward, data gradient (at the source resolution) and
    weight gradient incl. accumulation into an existing buffer."""
    B, Cin, H, W, Cout = case
    monkeypatch.setattr(ops, "UPCONV4", as_k4)
    x = T("ux%s" % (case,), (B, Cin, H, W)).requires_grad_(True)
    w = T("uw%s" % (case,), (Cout, Cin, 3, 3), 0.2).requires_grad_(True)
    ref = _conv_ref(x, w, 1, (1, 1), 1)
    g = T("ug%s" % (case,), ref.shape)
    ref.backward(g.double())
    xd, wd = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    y = ops.conv2d(xd, wd, None, 1, 1, True)
    y.backward(g.to(DEV))
    _check(y, ref, what="fwd"); _check(xd.grad, x.grad, what="dgrad"); _check(wd.grad, w.grad, what="wgrad")
    acc = wd.grad.clone()
    ops.conv2d_wgrad(g.to(DEV), xd.detach(), tuple(w.shape), 1, 1, 1, 1, out=acc, accumulate=True)
    _check(acc, 2 * w.grad, what="wgrad accumulate")


@pytest.mark.parametrize("case", [(2, 10, 17, 17, 12, (1, 7), 1, (0, 3)), (2, 48, 35, 35, 64, (5, 5), 1, (2, 2)),
                                  (2, 288, 35, 35, 96, (3, 3), 2, (0, 0)), (3, 768, 8, 8, 192, (1, 1), 1, (0, 0)),
                                  (2, 6, 9, 11, 5, (3, 3), 1, (1, 1))])
def test_conv_affine_relu_fused(case):
    """BasicConv2d of the frozen Inception trunk: conv + eval-BN affine + ReLU in the conv epilogue (direct store and
    split-K reduction), backward from the saved output."""
    B, Cin, H, W, Cout, k, s, pad = case
    x = T("fx%s" % (case,), (B, Cin, H, W)).requires_grad_(True)
    w = T("fw%s" % (case,), (Cout, Cin) + k, 0.2).requires_grad_(True)
    scale, shift = T("fs%s" % (case,), (Cout,), 0.5, 1.0), T("fb%s" % (case,), (Cout,), 0.3)
    ref = torch.relu(F.conv2d(x.double(), w.double(), None, s, pad) * scale.double().view(1, -1, 1, 1)
                     + shift.double().view(1, -1, 1, 1))
    g = T("fg%s" % (case,), ref.shape)
    ref.backward(g.double())
    xd, wd = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    z = ops.conv2d_affine_relu(xd, wd, scale.to(DEV), shift.to(DEV), s, pad)
    z.backward(g.to(DEV))
    _check(z, ref, what="fwd"); _check(xd.grad, x.grad, what="dgrad"); _check(wd.grad, w.grad, what="wgrad")


def test_conv_bias_and_wgrad_accumulate():
    x = T("cbx", (3, 12, 4, 4)).requires_grad_(True)
    w = T("cbw", (1, 12, 4, 4), 0.2).requires_grad_(True)
    b = T("cbb", (1,)).requires_grad_(True)
    ref = F.conv2d(x.double(), w.double(), b.double(), 4)
    ref.sum().backward()
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xd, wd, bd, 4, 0)
    y.sum().backward()
    _check(y, ref); _check(bd.grad, b.grad); _check(wd.grad, w.grad)
    acc = wd.grad.clone()
    ops.conv2d_wgrad(torch.ones_like(y), xd.detach(), tuple(w.shape), 4, 0, 0, 0, out=acc, accumulate=True)
    _check(acc, 2 * w.grad, what="accumulate")


@pytest.mark.parametrize("shape", [(16, 248, 1024), (3, 181, 100), (4, 256, 400), (130, 70, 33)])
def test_linear(shape):
    rows, fin, fout = shape
    x = T("lx%s" % (shape,), (rows, fin)).requires_grad_(True)
    w = T("lw%s" % (shape,), (fout, fin), 0.1).requires_grad_(True)
    b = T("lb%s" % (shape,), (fout,)).requires_grad_(True)
    ref = F.linear(x.double(), w.double(), b.double())
    g = T("lg%s" % (shape,), ref.shape)
    ref.backward(g.double())
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xd, wd, bd)
    y.backward(g.to(DEV))
    _check(y, ref); _check(xd.grad, x.grad); _check(wd.grad, w.grad); _check(bd.grad, b.grad)


def test_bmm_strided_views():
    a = T("ba", (3, 40, 50)).requires_grad_(True)
    b = T("bb", (3, 17, 50)).requires_grad_(True)       # used transposed
    ref = torch.bmm(a.double(), b.double().transpose(1, 2))
    g = T("bg", ref.shape)
    ref.backward(g.double())
    ad, bd = (t.detach().to(DEV).requires_grad_(True) for t in (a, b))
    y = ops.bmm(ad, bd.transpose(1, 2))
    y.backward(g.to(DEV))
    _check(y, ref); _check(ad.grad, a.grad); _check(bd.grad, b.grad)


ACTS = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "glu": ops.ACT_GLU}


def _act_ref(y, act):
    if act == "relu":
        return F.relu(y)
    if act == "lrelu":
        return F.leaky_relu(y, 0.2)
    if act == "glu":
        return O.glu(y)
    return y


@pytest.mark.parametrize("shape", [(3, 4, 6, 8, 8), (3, 16, 12, 15, 15), (2, 5, 10), (3, 16, 8, 16, 16), (2, 16, 8, 32, 32),
                                   (2, 3, 4, 70, 70)])
@pytest.mark.parametrize("act", ["none", "relu", "lrelu", "glu"])
def test_bn_act_grouped(shape, act):
    """mogan_bn_act_grouped_fwd/bwd: G BatchNorm(train)+activation calls on the G groups of one (G*B, C, ...) tensor in one launch
    each way = G separate calls in sequence (own statistics per group, running statistics updated group after group, d gamma / d beta
    summed) -- against fp64 and against the looped HIP calls; (2,5,10) is a BatchNorm1d.  The last two shapes have more than 4096
    values per channel and group: the large-map kernels once per group, in place (the [real; fake] batch of a discriminator
    update on the 64 x 64 ... 16 x 16 maps, miscc/losses.py:136-174)."""
    G, B, C = shape[:3]
    full = (G * B, C) + tuple(shape[3:])
    x = T("gbx%s" % (shape,), full, 1.5, 0.3).requires_grad_(True)
    gm = T("gbg%d" % C, (C,), 0.2, 1.0).requires_grad_(True)
    bt = T("gbb%d" % C, (C,), 0.2).requires_grad_(True)
    rm, rv = T("gbrm%d" % C, (C,), 0.1), T("gbrv%d" % C, (C,), 0.1).abs() + 1
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    outs = []
    for g in range(G):
        yb = F.batch_norm(x.double()[g * B:(g + 1) * B], rm_ref, rv_ref, gm.double(), bt.double(), True, 0.1, 1e-5)
        outs.append(_act_ref(yb, act))
    ref = torch.cat(outs)
    go = T("gbgo%s%s" % (shape, act), ref.shape)
    ref.backward(go.double())
    code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "glu": ops.ACT_GLU}[act]
    xd, gd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, gm, bt))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    assert ops.bn_groups_ok(xd, G) == (B * int(torch.tensor(shape[3:]).prod()) <= 4096)
    y = ops.bn_act(xd, gd, bd, rmd, rvd, code, 0.2, None, 1e-5, 0.1, groups=G)
    y.backward(go.to(DEV))
    torch.cuda.synchronize()
    _check(y, ref, what="y")
    _check(xd.grad, x.grad, what="dx")
    _check(gd.grad, gm.grad, what="dgamma")
    _check(bd.grad, bt.grad, what="dbeta")
    _check(rmd, rm_ref, what="running_mean")
    _check(rvd, rv_ref, what="running_var")
    # ... and bit for bit what G calls of the one-group path give
    x2, g2, b2 = (t.detach().to(DEV).requires_grad_(True) for t in (x, gm, bt))
    rm2, rv2 = rm.to(DEV), rv.to(DEV)
    y2 = torch.cat([ops.bn_act(x2[g * B:(g + 1) * B], g2, b2, rm2, rv2, code, 0.2, None, 1e-5, 0.1) for g in range(G)])
    y2.backward(go.to(DEV))
    torch.cuda.synchronize()
    if len(shape) > 3 and shape[3] * shape[4] >= 16:          # (the looped BatchNorm1d takes the thread-per-channel kernels)
        assert torch.equal(y, y2) and torch.equal(xd.grad, x2.grad) and torch.equal(rmd, rm2) and torch.equal(rvd, rv2)
    else:
        _check(y, y2.double().cpu(), what="y vs loop")
    assert not ops.bn_groups_ok(torch.empty(3 * 32, 4, 16, 16, device=DEV), 3)        # 8192 values per channel and group


def test_group_sum():
    x = T("gsx", (3 * 5, 7, 4, 4)).requires_grad_(True)
    ref = x.double().view(3, 5, 7, 4, 4).sum(0)
    g = T("gsg", ref.shape)
    ref.backward(g.double())
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops.group_sum(xd, 3)
    y.backward(g.to(DEV))
    torch.cuda.synchronize()
    _check(y, ref, what="sum")
    _check(xd.grad, x.grad, what="broadcast")
    v = xd.detach().view(3, 5, 7, 4, 4)
    assert torch.equal(y, (v[0] + v[1]) + v[2])               # the loop's order


# one-launch path (<= 4096 values per channel), two-launch path (HW % 4 == 0: finalize folded into the apply pass; (3,4,96,96): three
# tiles per plane, the last one partial), three-launch path ((20,4,15,15): odd plane size beyond the one-launch limit)
@pytest.mark.parametrize("shape", [(4, 8, 8, 8), (3, 6, 15, 15), (16, 24), (2, 10, 64, 64), (5, 4, 3, 3), (16, 6, 16, 16), (17, 6, 16, 16),
                                   (3, 4, 96, 96), (20, 4, 15, 15)])
@pytest.mark.parametrize("act", ["none", "relu", "lrelu", "glu"])
@pytest.mark.parametrize("res", [False, True])
def test_bn_act(shape, act, res):
    if res and act != "none":
        pytest.skip("residual only follows a plain BN in the reference (ResBlock)")
    C = shape[1]
    x = T("bnx%s" % (shape,), shape, 1.5, 0.3).requires_grad_(True)
    gm = T("bng%d" % C, (C,), 0.2, 1.0).requires_grad_(True)
    bt = T("bnb%d" % C, (C,), 0.2).requires_grad_(True)
    rm, rv = T("bnrm%d" % C, (C,), 0.1), T("bnrv%d" % C, (C,), 0.1).abs() + 1
    r = T("bnres%s" % (shape,), shape).requires_grad_(True) if res else None
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    yb = F.batch_norm(x.double(), rm_ref, rv_ref, gm.double(), bt.double(), True, 0.1, 1e-5)
    ref = _act_ref(yb, act) + (r.double() if res else 0)
    g = T("bngo%s%s" % (shape, act), ref.shape)
    ref.backward(g.double())
    xd, gd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, gm, bt))
    rd = r.detach().to(DEV).requires_grad_(True) if res else None
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    y = ops.bn_act(xd, gd, bd, rmd, rvd, ACTS[act], 0.2, rd)
    y.backward(g.to(DEV))
    _check(y, ref, 5e-6, "y")
    _check(rmd, rm_ref, 1e-6, "running_mean"); _check(rvd, rv_ref, 1e-6, "running_var")
    _check(xd.grad, x.grad, 2e-5, "dx"); _check(gd.grad, gm.grad, 2e-5, "dgamma"); _check(bd.grad, bt.grad, 2e-5, "dbeta")
    if res:
        _check(rd.grad, r.grad, 1e-7, "dres")


@pytest.mark.parametrize("act", ["relu", "lrelu", "tanh", "sigmoid", "glu"])
def test_act(act):
    x = T("ax" + act, (3, 8, 5, 7)).requires_grad_(True)
    code = dict(ACTS, tanh=ops.ACT_TANH, sigmoid=ops.ACT_SIGMOID)[act]
    ref = {"tanh": torch.tanh, "sigmoid": torch.sigmoid}.get(act, lambda v: _act_ref(v, act))(x.double())
    g = T("ag" + act, ref.shape)
    ref.backward(g.double())
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops.act(xd, code, 0.2)
    y.backward(g.to(DEV))
    _check(y, ref, 2e-6); _check(xd.grad, x.grad, 2e-6)


@pytest.mark.parametrize("ac", [False, True])
def test_stn(ac):
    bbox = torch.tensor([[0.1, 0.2, 0.3, 0.4], [-1, -1, -1, -1], [0.4, 0.1, 0.55, 0.8], [0.0, 0.0, 1.0, 1.0]])
    th, thi = ops.bbox_to_theta(bbox.to(DEV))
    from mogan_amd.attngan import synthetic
    th_ref, thi_ref = synthetic.bbox_to_theta(bbox)
    assert torch.equal(th.cpu(), th_ref) and torch.equal(thi.cpu(), thi_ref)
    rot = T("stn.rot", (4, 2, 3), 0.5) + torch.tensor([[1., 0, 0], [0, 1., 0]])
    for theta, insz, outsz in ((thi_ref, (4, 5, 8, 8), (4, 5, 8, 8)), (th_ref, (4, 3, 64, 64), (4, 3, 16, 16)),
                               (thi_ref, (4, 7, 15, 15), (4, 7, 16, 16)), (rot, (4, 4, 7, 9), (4, 4, 5, 6))):
        x = T("stnx%s" % (insz,), insz).requires_grad_(True)
        ref = O.stn(x.double(), theta.double(), outsz, align_corners=ac)
        g = T("stng%s" % (outsz,), outsz)
        ref.backward(g.double())
        xd = x.detach().to(DEV).requires_grad_(True)
        y = ops.stn(xd, theta.to(DEV), outsz, ac)
        y.backward(g.to(DEV))
        _check(y, ref, 1e-5, "stn y"); _check(xd.grad, x.grad, 1e-5, "stn dx")
        assert float(y[1].detach().abs().max()) == 0.0 or theta is rot    # absent object -> exactly 0


@pytest.mark.parametrize("B,Cin,Cout", [(16, 768, 1), (15, 32, 1), (5, 70, 3)])
def test_logits_head_and_conv_lrelu(B, Cin, Cout):
    """mogan_logits_head_fwd / _bwd (full-map Conv2d + bias + Sigmoid, model.py:626-627, 640-641) and mogan_conv2d_lrelu_fwd
    (first discriminator layer: Conv2d(3, ndf, 4, 2, 1) + LeakyReLU(0.2), model.py:597-598) against torch in fp64: value,
    input gradient, weight / bias gradients -- returned and accumulated into existing .grad buffers."""
    x = T("lh.x%d" % Cin, (B, Cin, 4, 4)).requires_grad_(True)
    w = T("lh.w%d" % Cin, (Cout, Cin, 4, 4), 0.05).requires_grad_(True)
    b = T("lh.b%d" % Cin, (Cout,), 0.1).requires_grad_(True)
    ref = torch.sigmoid(F.conv2d(x.double(), w.double(), b.double(), 4))
    g = T("lh.g%d" % Cin, ref.shape)
    ref.backward(g.double())
    xd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w, b))
    p = ops.logits_head(xd, wd, bd)
    p.backward(g.to(DEV))
    _check(p, ref, 2e-6, "head p"); _check(xd.grad, x.grad, 5e-6, "head dx")
    _check(wd.grad, w.grad, 5e-6, "head dw"); _check(bd.grad, b.grad, 5e-6, "head db")
    # a second backward accumulates straight into the existing dense .grad buffers (the trainer's flat buckets)
    p2 = ops.logits_head(xd.detach(), wd, bd)
    p2.backward(g.to(DEV))
    _check(wd.grad, 2 * w.grad, 5e-6, "head dw x2"); _check(bd.grad, 2 * b.grad, 5e-6, "head db x2")
    # conv + LeakyReLU in the epilogue
    xi = T("cl.x%d" % B, (min(B, 4), 3, 32, 32)).requires_grad_(True)
    wi = T("cl.w%d" % B, (24, 3, 4, 4), 0.2).requires_grad_(True)
    ref = F.leaky_relu(F.conv2d(xi.double(), wi.double(), None, 2, 1), 0.2)
    g = T("cl.g%d" % B, ref.shape)
    ref.backward(g.double())
    xid, wid = (t.detach().to(DEV).requires_grad_(True) for t in (xi, wi))
    z = ops.conv2d_lrelu(xid, wid, 2, 1, 0.2)
    z.backward(g.to(DEV))
    _check(z, ref, 2e-6, "conv+lrelu"); _check(xid.grad, xi.grad, 5e-6, "conv+lrelu dx"); _check(wid.grad, wi.grad, 5e-6, "conv+lrelu dw")


@pytest.mark.parametrize("B,H,W,Cout", [(2, 64, 64, 96), (3, 16, 128, 48), (1, 2, 64, 32), (2, 66, 192, 128), (2, 64, 64, 40)])
def test_first_discriminator_layer_streaming_kernel(B, H, W, Cout):
    """csrc/mogan_stem.hip (round 5): Conv2d(3, ndf, 4, 2, 1) [+ LeakyReLU(0.2)] of model.py:597-598, 660-661 on output rows that are
    multiples of 32 pixels -- one wave = 32 pixels x all channels, both image borders, a one-row output, 1-4 row tiles, the register
    sets alternating over several groups per wave -- against torch in fp64 and against the implicit-GEMM kernel it replaces
    (mogan_gemm_debug_force(0, 0) keeps every convolution on that kernel); input / weight gradients of the fused Function."""
    x = T("st.x%d" % H, (B, 3, H, W)).requires_grad_(True)
    w = T("st.w%d" % Cout, (Cout, 3, 4, 4), 0.2).requires_grad_(True)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), None, 2, 1), 0.2)
    g = T("st.g%d%d" % (H, Cout), ref.shape)
    ref.backward(g.double())
    xd, wd = (t.detach().to(DEV).requires_grad_(True) for t in (x, w))
    z = ops.conv2d_lrelu(xd, wd, 2, 1, 0.2)
    z.backward(g.to(DEV))
    _check(z, ref, 2e-6, "stem"); _check(xd.grad, x.grad, 5e-6, "stem dx"); _check(wd.grad, w.grad, 5e-6, "stem dw")
    y = ops.conv2d_forward(xd.detach(), wd.detach(), 2, 1, 1, 0)
    _check(y, F.conv2d(x.double(), w.double(), None, 2, 1), 2e-6, "stem, plain convolution")
    lib.load().mogan_gemm_debug_force(0, 0)
    try:
        z0 = ops.conv2d_lrelu(xd.detach(), wd.detach(), 2, 1, 0.2)
    finally:
        lib.load().mogan_gemm_debug_force(-1, 0)
    assert float((z.detach() - z0).abs().max()) <= 2e-6 * float(z0.abs().max())
    assert torch.equal(z.detach() > 0, z0 > 0) or float((z.detach() - z0).abs().max()) < 1e-6      # same side of the kink


def test_stn_shared_source_gradient_is_order_independent_to_rounding():
    """mogan_stn_bwd_ex adds every object's contribution to the ONE image-batch gradient with fp32 atomics (include/mogan_hip.h,
    "Determinism"): two runs agree to the rounding of a short sum, and both agree with the fixed-order sum of the per-object
    gradients to the same bound -- not bit for bit."""
    B, G, C, S = 5, 3, 7, 16
    x = T("sdx", (B, C, 32, 32)).to(DEV)
    th = (T("sdth", (B, G, 2, 3), 0.3) + torch.tensor([[1.0, 0, 0], [0, 1.0, 0]])).to(DEV)
    gy = T("sdg", (G * B, C, S, S)).to(DEV)
    outs = []
    for _ in range(3):
        xd = x.clone().requires_grad_(True)
        y = ops.stn_shared(xd, th, G * B, (32, 32), (S, S), False, False, G)
        y.backward(gy)
        outs.append(xd.grad.clone())
    ref = torch.zeros_like(x, dtype=torch.float64)
    for g in range(G):                                   # the reference: one stn per object, gradients added in object order
        xd = x.clone().requires_grad_(True)
        y = ops.stn(xd, th[:, g].contiguous(), (B, C, S, S))
        y.backward(gy[g * B:(g + 1) * B])
        ref += xd.grad.double()
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    for o in outs:
        assert float((o.double() - ref).abs().max()) <= 4e-6 * scale
    assert float((outs[0] - outs[1]).abs().max()) <= 4e-6 * scale and float((outs[1] - outs[2]).abs().max()) <= 4e-6 * scale


@pytest.mark.parametrize("ac", [False, True])
def test_stn_shared_and_constant_sources(ac):
    """mogan_stn_*_ex: the object pathways' transformers without their materialised inputs (model.py:109-111, 402-404, 663-671)
    -- one image batch read by every object (x index b % xB, gradients of all objects collected in the one dx), a label vector
    constant over the plane (dx = (B, C)), theta in the loader's (image, object) layout for an object-major batch -- against the
    plain transformer on the materialised tensors, forward and backward, in fp64."""
    Bp, G, C = 4, 3, 5
    bbox = torch.rand(Bp * G, 4, generator=torch.Generator().manual_seed(5)) * 0.5 + 0.05
    bbox[4] = -1.0                                                       # an absent object
    th, thi = (t.view(Bp, G, 2, 3) for t in __import__("mogan_amd.attngan.synthetic", fromlist=["x"]).bbox_to_theta(bbox))
    objmajor = lambda t: t.transpose(0, 1).reshape(G * Bp, 2, 3)         # sample n = g * Bp + b
    # (a) shared image, crop to 16 x 16
    x = T("stnsh.x", (Bp, C, 20, 24)).requires_grad_(True)
    ref = O.stn(x.double().repeat(G, 1, 1, 1), objmajor(th).double(), (G * Bp, C, 16, 16), align_corners=ac)
    g = T("stnsh.g", ref.shape)
    ref.backward(g.double())
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops.stn_shared(xd, th.to(DEV), G * Bp, (20, 24), (16, 16), ac, theta_G=G)
    y.backward(g.to(DEV))
    _check(y, ref, 1e-5, "shared y"); _check(xd.grad, x.grad, 1e-5, "shared dx")
    # (b) one map per (object, image), theta looked up object-major
    x = T("stnsh.x2", (G * Bp, C, 15, 15)).requires_grad_(True)
    ref = O.stn(x.double(), objmajor(thi).double(), (G * Bp, C, 16, 16), align_corners=ac)
    ref.backward(g.double())
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops.stn_shared(xd, thi.to(DEV), G * Bp, (15, 15), (16, 16), ac, theta_G=G)
    y.backward(g.to(DEV))
    _check(y, ref, 1e-5, "per-object y"); _check(xd.grad, x.grad, 1e-5, "per-object dx")
    # (c) constant source: a (G*Bp, C) label vector standing for its 16 x 16 repetition
    v = T("stnsh.v", (G * Bp, C)).requires_grad_(True)
    ref = O.stn(v.double().view(G * Bp, C, 1, 1).repeat(1, 1, 16, 16), objmajor(thi).double(), (G * Bp, C, 16, 16),
                align_corners=ac)
    ref.backward(g.double())
    vd = v.detach().to(DEV).requires_grad_(True)
    y = ops.stn_shared(vd, thi.to(DEV), G * Bp, (16, 16), (16, 16), ac, plane=True, theta_G=G)
    y.backward(g.to(DEV))
    _check(y, ref, 1e-5, "plane y"); _check(vd.grad, v.grad, 1e-5, "plane dx")
    assert float(y[1 * Bp + 1].detach().abs().max()) == 0.0             # bbox 4 = (image 1, object 1): absent -> exactly 0


@pytest.mark.parametrize("spatial", [(), (4, 4), (16, 16), (5, 3)])
def test_cat_channels(spatial):
    """mogan_concat_fwd / _bwd: torch.cat(..., 1) whose parts are plain tensors, codes repeated over the plane, one tensor
    repeated for every object and per-object slices of a (B, G, C) tensor (model.py:400-401, 418, 457, 633-634, 666, 703) --
    forward bit-exact (it only moves values), gradients against autograd of the torch expression in fp64."""
    B, G = 4, 3
    N = G * B
    full = T("cat.full%s" % (spatial,), (N, 6) + spatial)
    plane = T("cat.plane", (N, 5))
    rep = T("cat.rep%s" % (spatial,), (B, 3) + spatial)
    objp = T("cat.objp", (B, G, 7))
    ones = (1,) * len(spatial)
    def torch_expr(full, plane, rep, objp):
        a = plane.view(N, 5, *ones).expand(N, 5, *spatial)
        b = rep.repeat(G, *([1] * (1 + len(spatial))))
        c = objp.transpose(0, 1).reshape(N, 7, *ones).expand(N, 7, *spatial)
        return torch.cat((full, a, b, c), 1)
    leaves = [t.clone().double().requires_grad_(True) for t in (full, plane, rep, objp)]
    ref = torch_expr(*leaves)
    g = T("cat.g%s" % (spatial,), ref.shape)
    ref.backward(g.double())
    dl = [t.clone().to(DEV).requires_grad_(True) for t in (full, plane, rep, objp)]
    y = ops.cat_channels([(dl[0], "full"), (dl[1], "plane"), (dl[2], ("rep", G)), (dl[3], ("obj_plane", G))], N, spatial)
    assert torch.equal(y.detach().cpu(), ref.detach().float())
    y.backward(g.to(DEV))
    for got, want, what in zip(dl, leaves, ("full", "plane", "rep", "obj_plane")):
        _check(got.grad, want.grad, 2e-6, "cat d" + what)
    # per-object slices of a (B, G, C, *spatial) tensor, and a part without gradient
    obj = T("cat.obj%s" % (spatial,), (B, G, 2) + spatial)
    lo = obj.clone().double().requires_grad_(True)
    ref = torch.cat((lo.transpose(0, 1).reshape((N, 2) + spatial), full.double()), 1)
    g = T("cat.g2%s" % (spatial,), ref.shape)
    ref.backward(g.double())
    do = obj.clone().to(DEV).requires_grad_(True)
    y = ops.cat_channels([(do, ("obj", G)), (full.to(DEV), "full")], N, spatial)
    assert torch.equal(y.detach().cpu(), ref.detach().float())
    y.backward(g.to(DEV))
    _check(do.grad, lo.grad, 2e-6, "cat dobj")


@pytest.mark.parametrize("B,idf,Q,T_", [(3, 6, 16, 5), (4, 48, 4096, 12), (2, 96, 300, 20)])
def test_attention(B, idf, Q, T_):
    h = T("ath%d" % B, (B, idf, Q)).requires_grad_(True)
    src = T("ats%d" % B, (B, idf, T_), 0.3).requires_grad_(True)
    mask = torch.zeros(B, T_, dtype=torch.bool)
    for b in range(B):
        mask[b, max(1, T_ - 1 - b):] = True
    for mode in (0, 1):
        for t in (h, src):
            t.grad = None
        sc = torch.bmm(h.double().transpose(1, 2), src.double()).reshape(B * Q, T_)
        rows = (torch.arange(B * Q) % B) if mode == 0 else (torch.arange(B * Q) // Q)
        att = torch.softmax(sc.masked_fill(mask[rows], -float("inf")), 1).reshape(B, Q, T_).transpose(1, 2)
        wc = torch.bmm(src.double(), att)
        gw, ga = T("atgw%d" % B, wc.shape), T("atga%d" % B, att.shape)
        ((wc * gw.double()).sum() + (att * ga.double()).sum()).backward()
        hd, sd = (t.detach().to(DEV).requires_grad_(True) for t in (h, src))
        wcd, attd = ops.attention(hd, sd, mask.to(DEV), mode)
        ((wcd * gw.to(DEV)).sum() + (attd * ga.to(DEV)).sum()).backward()
        _check(wcd, wc, 5e-6, "wc"); _check(attd, att, 5e-6, "attn")
        _check(hd.grad, h.grad, 2e-5, "dh"); _check(sd.grad, src.grad, 2e-5, "dsrc")


def test_softmax_masked_and_scaled():
    x = T("smx", (3, 7, 5, 4)).requires_grad_(True)
    lens = torch.tensor([[5, 3, 1, 4]] * 21, dtype=torch.int32).reshape(3, 7, 4)
    ref = torch.zeros_like(x, dtype=torch.float64)
    xd64 = x.double()
    for k in range(4):
        n = int(lens[0, 0, k])
        ref = ref.clone()
        ref[:, :, :n, k] = torch.softmax(4.0 * xd64[:, :, :n, k], 2)
    g = T("smg", x.shape)
    ref.backward(g.double())
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops.softmax(xd, 2, 4.0, lens.to(DEV).contiguous())
    y.backward(g.to(DEV))
    _check(y, ref, 2e-6); _check(xd.grad, x.grad, 1e-5)
    y2 = ops.softmax(xd.detach(), 3)
    _check(y2, torch.softmax(x.detach().double(), 3), 2e-6)


def test_losses_bce_kl():
    p = torch.sigmoid(T("bcep", (16,), 2.0)).requires_grad_(True)
    for tgt in (0.0, 1.0):
        p.grad = None
        ref = O.bce(p.double(), torch.full((16,), tgt, dtype=torch.float64))
        (ref * 1.7).backward()
        pd = p.detach().to(DEV).requires_grad_(True)
        l = ops.bce(pd, tgt)
        (l * 1.7).backward()
        _check(l, ref, 1e-6); _check(pd.grad, p.grad, 1e-6)
    # torch's clamp of log at -100
    pd = torch.tensor([0.0, 1.0, 0.5], device=DEV)
    assert abs(float(ops.bce(pd, 1.0)) - (100 + 0 + np.log(2)) / 3) < 1e-4
    # saturated probabilities (a discriminator that has seen one batch for 30 steps): torch's backward is
    # (p - t) / max(p (1 - p), 1e-12), not the derivative of the clamped logs
    sat = torch.tensor([0.0, 1.0, 1e-13, 1.0 - 6e-8, 3e-20, 0.3], dtype=torch.float32)
    for tgt in (0.0, 1.0):
        pr = sat.clone().requires_grad_(True)
        F.binary_cross_entropy(pr, torch.full_like(pr, tgt)).backward()
        ph = sat.to(DEV).requires_grad_(True)
        ops.bce(ph, tgt).backward()
        assert torch.isfinite(ph.grad).all()
        np.testing.assert_allclose(ph.grad.cpu().numpy(), pr.grad.numpy(), rtol=1e-5, atol=0)
    mu = T("klm", (16, 100), 0.5).requires_grad_(True)
    lv = T("kll", (16, 100), 0.5).requires_grad_(True)
    ref = O.kl_loss(mu.double(), lv.double())
    ref.backward()
    md, ld = (t.detach().to(DEV).requires_grad_(True) for t in (mu, lv))
    l = ops.kl_loss(md, ld)
    l.backward()
    _check(l, ref, 1e-6); _check(md.grad, mu.grad, 1e-6); _check(ld.grad, lv.grad, 1e-6)


def test_pooling_and_resize():
    x = T("plx", (2, 5, 35, 35)).requires_grad_(True)
    for name, fn_ref, fn in (
            ("max", lambda v: F.max_pool2d(v, 3, 2), lambda v: ops.max_pool2d(v, 3, 2)),
            ("avg", lambda v: F.avg_pool2d(v, 3, 1, 1), lambda v: ops.avg_pool2d(v, 3, 1, 1)),
            ("avg8", lambda v: F.avg_pool2d(v[:, :, :8, :8], 8), lambda v: ops.avg_pool2d(v[:, :, :8, :8], 8)),
            ("bil", lambda v: F.interpolate(v, size=(61, 61), mode="bilinear", align_corners=False),
             lambda v: ops.bilinear_resize(v, 61, 61))):
        x.grad = None
        ref = fn_ref(x.double())
        g = T("plg" + name, ref.shape)
        ref.backward(g.double())
        xd = x.detach().to(DEV).requires_grad_(True)
        y = fn(xd)
        y.backward(g.to(DEV))
        _check(y, ref, 5e-6, name); _check(xd.grad, x.grad, 5e-6, name + " dx")


@pytest.mark.parametrize("shape,out", [((2, 3, 64, 64), (75, 75)), ((1, 2, 256, 256), (299, 299)), ((2, 2, 40, 24), (13, 9)),
                                       ((1, 1, 30, 30), (7, 200)), ((1, 2, 17, 17), (17, 17))])
def test_bilinear_resize_backward_gather(shape, out):
    """mogan_bilinear_bwd in gather form (one thread per input element, no atomics): up-scaling (the 256 -> 299 resize in front
    of the Inception trunk, model.py:256), down-scaling past the register path's candidate count, and identity, against
    autograd of F.interpolate in fp64."""
    x = T("bilg%s%s" % (shape, out), shape).requires_grad_(True)
    ref = F.interpolate(x.double(), size=out, mode="bilinear", align_corners=False)
    g = T("bilgg%s%s" % (shape, out), ref.shape)
    ref.backward(g.double())
    xd = x.detach().to(DEV).requires_grad_(True)
    y = ops.bilinear_resize(xd, out[0], out[1])
    y.backward(g.to(DEV))
    # (the source coordinate (o + 0.5) * scale - 0.5 is fp32 arithmetic here as in torch's own float kernels; the reference of
    # this test is fp64: 2e-5 covers the weight rounding at 256 -> 299)
    _check(y, ref, 2e-5, "resize"); _check(xd.grad, x.grad, 2e-5, "resize dx")
    y2 = ops.bilinear_resize(xd, out[0], out[1])
    xd.grad = None
    y2.backward(g.to(DEV))
    g1 = xd.grad.clone()
    xd.grad = None
    ops.bilinear_resize(xd, out[0], out[1]).backward(g.to(DEV))
    assert torch.equal(g1, xd.grad)                                      # deterministic (the scatter form was not)


@pytest.mark.parametrize("eps_mode", [0, 1])
def test_adam_ema(eps_mode):
    n = 10007
    p, g = T("adp", (n,)), T("adg", (n,), 0.01)
    net = {"w": p.clone().double().requires_grad_(True)}
    st = O.adam_state(net)
    ema = p.clone().double()
    pd, gd = p.to(DEV), g.to(DEV)
    pad = (-n) % 4
    m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV); emad = p.to(DEV).clone()
    state = torch.zeros(3, device=DEV)
    for step in range(1, 4):
        net["w"].grad = (g.double() * step)
        O.adam_step(net, st, 2e-4, eps_mode="torch2" if eps_mode == 0 else "torch041")
        ema.mul_(0.999).add_(net["w"].detach(), alpha=0.001)
        if step < 3:
            ops.adam_step(pd, gd * step, m, v, emad, 2e-4, 0.5, 0.999, 1e-8, step=step, eps_mode=eps_mode)
            state[0] = step
        else:   # device-resident step counter (hipGraph-friendly)
            ops.adam_step(pd, gd * step, m, v, emad, 2e-4, 0.5, 0.999, 1e-8, dev_state=state, eps_mode=eps_mode)
    assert float((pd.cpu().double() - net["w"].detach()).abs().max()) < 1e-6   # |p|~3: 2-3 ulp
    assert float((emad.cpu().double() - ema).abs().max()) < 1e-6


@pytest.mark.parametrize("B,C,hw,Tw,same_class", [(16, 256, 17, 12, False), (5, 32, 6, 7, True), (3, 16, 17, 18, False)])
def test_damsm_words_and_sentence_losses(B, C, hw, Tw, same_class):
    """mogan_damsm_words_fwd/bwd + mogan_damsm_ce_* + mogan_damsm_sent_* (miscc/losses.py:20-132, one func_attention per
    caption in the reference) against the fp64 oracle: both cross-entropies, the attention maps, the gradients w.r.t.
    the region features / the image code; ragged caption lengths, optional same-class mask, full benchmark size."""
    from mogan_amd.attngan.miscc import losses as L
    from mogan_amd.attngan.miscc.config import cfg
    cfg.TRAIN.SMOOTH.GAMMA1, cfg.TRAIN.SMOOTH.GAMMA2, cfg.TRAIN.SMOOTH.GAMMA3 = 4.0, 5.0, 10.0
    ocfg = O.Cfg(words_num=Tw)
    feat = T("damsm.feat%d" % B, (B, C, hw, hw))
    words = T("damsm.words%d" % B, (B, C, Tw))
    code, sent = T("damsm.code%d" % B, (B, C)), T("damsm.sent%d" % B, (B, C))
    lens = np.sort(np.random.RandomState(B).randint(2, Tw + 1, B))[::-1].copy()
    lens[0] = Tw
    class_ids = np.arange(B)
    if same_class:
        class_ids[2] = class_ids[0]                       # samples 0 and 2 share a class: masked out of each other's rows
    fd, cd = feat.double().requires_grad_(True), code.double().requires_grad_(True)
    w0, w1, att = O.words_loss(fd, words.double(), lens, ocfg, class_ids if same_class else None)
    s0, s1 = O.sent_loss(cd, sent.double(), ocfg, class_ids if same_class else None)
    (1.3 * w0 + 0.7 * w1 + 2.0 * s0 + 0.5 * s1).backward()
    fg, cg = feat.to(DEV).requires_grad_(True), code.to(DEV).requires_grad_(True)
    lab = torch.arange(B, device=DEV)
    lens_t = torch.from_numpy(lens.astype(np.int64))
    g0, g1, _ = L.words_loss(fg, words.to(DEV), lab, lens_t, class_ids, B)
    t0, t1 = L.sent_loss(cg, sent.to(DEV), lab, class_ids, B)
    ops.scalar_sum([g0, g1, t0, t1], [1.3, 0.7, 2.0, 0.5]).backward()
    torch.cuda.synchronize()
    for got, want, k in ((g0, w0, "w0"), (g1, w1, "w1"), (t0, s0, "s0"), (t1, s1, "s1")):
        np.testing.assert_allclose(float(got), float(want), rtol=2e-5, err_msg=k)
    assert rel_l2(fg.grad, fd.grad) <= 2e-5, rel_l2(fg.grad, fd.grad)
    assert rel_l2(cg.grad, cd.grad) <= 2e-5, rel_l2(cg.grad, cd.grad)
    with torch.no_grad():
        _, _, maps = L.words_loss(fg, words.to(DEV), None, lens_t, class_ids, B)
    for i in (0, B - 1):
        assert tuple(maps[i].shape) == (1, int(lens[i]), hw, hw)
        assert max_abs(maps[i], att[i]) <= 2e-6


@pytest.mark.parametrize("layer", [(8, 96, 32, 192, 4, 2, 1), (16, 384, 8, 768, 4, 2, 1), (4, 96, 64, 96, 3, 1, 1), (8, 192, 17, 64, 1, 1, 0)])
@pytest.mark.parametrize("kind", ["wide", "cancel"])
def test_fp32_products_on_the_bf16_pipe_hold_the_fp32_error_bound(layer, kind):
    """csrc/mogan_mma.h forms every fp32 product from exact 3-piece bf16 splits (6 partial products, dropped terms <=
    2^-23 |ab|).  Ordinary data is covered above; here: (wide) every value scaled by 2^U(-20,20) -- products of very
    different magnitude in one sum, rel-L2 against fp64 <= 1e-6 (measured 1e-7 .. 5e-7, the native fp32-MFMA build the
    same); (cancel) adjacent channels x, -x(1 + 1e-4 eps) with equal weights, so that the result is ~1e-4 of the terms:
    max |err| / sum |a||b| <= 2^-24 (one fp32 rounding of the term magnitude; measured <= 1.6e-8).  Forward (4x4 s2 through
    the direct / implicit-GEMM kernels, 3x3 through Winograd, 1x1 through the implicit GEMM) and data gradient."""
    B, Cin, H, Cout, k, s, p = layer
    g = torch.Generator().manual_seed(Cin * 7 + H)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k)) ** 0.5
    if kind == "wide":
        x = x * torch.exp2(torch.empty(x.shape).uniform_(-20, 20, generator=g))
        w = w * torch.exp2(torch.empty(w.shape).uniform_(-20, 20, generator=g))
    else:
        x[:, 1::2] = -x[:, 0::2] * (1 + 1e-4 * torch.randn(x[:, 1::2].shape, generator=g))
        w[:, 1::2] = w[:, 0::2]
    xd, wd = x.double().requires_grad_(True), w.double()
    yd = F.conv2d(xd, wd, None, s, p)
    dy = torch.randn(yd.shape, generator=g)
    yd.backward(dy.double())
    xa = x.double().abs().requires_grad_(True)
    ya = F.conv2d(xa, wd.abs(), None, s, p)
    ya.backward(dy.double().abs())
    y = ops.conv2d_forward(x.to(DEV), w.to(DEV), s, p, p, 0)
    dx = ops.conv2d_dgrad(dy.to(DEV), w.to(DEV), x.shape, s, p, p, 0)
    torch.cuda.synchronize()
    for got, want, scale, what in ((y, yd.detach(), ya.detach(), "fwd"), (dx, xd.grad, xa.grad, "dgrad")):
        err = (got.double().cpu() - want).abs()
        bound = float((err / scale.clamp_min(1e-300)).max())
        # the 3x3 layer runs the Winograd F(2x2,3x3) kernels, whose input / filter transforms add and subtract neighbouring
        # values: on wide-range data single elements lose up to ~3e-5 of the term magnitude (2.6e-5 here, 3.8e-5 in the native
        # fp32-MFMA build, 7.6e-7 through the direct kernel) -- a property of the transform, not of the split; rel-L2 below
        limit = 2.0 ** -14 if (k == 3 and kind == "wide") else 2.0 ** -19
        assert bound <= limit, "%s %s: max err / sum|a||b| = %.2e" % (kind, what, bound)
        if kind == "wide":
            assert rel_l2(got, want) <= 1e-6, "%s %s: rel-L2 %.2e" % (kind, what, rel_l2(got, want))
        elif what == "fwd":
            assert bound <= 2.0 ** -24, "cancel fwd: max err / sum|a||b| = %.2e" % bound


def test_fp32_products_on_the_bf16_pipe_input_domain():
    """The edges of the split form's input domain (include/mogan_hip.h, "Arithmetic"): (huge) |x| up to the largest bf16
    value 3.3895e38 is exact like everything else -- here 3.0e38 against weights of 1e-3, finite fp32 results; beyond it (up
    to FLT_MAX = 3.4028e38, and +-inf) the first piece rounds to infinity and the result is inf / NaN, never a silently wrong
    finite number; (tiny) inputs down to 2^-100 keep all three pieces normal bf16 numbers: rel-L2 <= 1e-6; below ~2^-110 the
    third (then the second) piece falls under the smallest normal bf16 and the result degrades gracefully towards the 8 / 16
    bits of the remaining pieces -- at 2^-118: rel-L2 <= 2^-12 -- on numbers whose squares are far below fp32's range."""
    g = torch.Generator().manual_seed(5)
    B, Cin, H, Cout = 2, 64, 8, 64
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.125
    run = lambda a, b: ops.conv2d_forward(a.to(DEV), b.to(DEV), 1, 0, 0, 0).cpu().double()
    ref = lambda a, b: F.conv2d(a.double(), b.double())
    # huge but representable first pieces
    xs = torch.sign(x) * 3.0e38
    ws = w * 1e-3 / 64
    y = run(xs, ws)
    assert torch.isfinite(y).all() and rel_l2(y, ref(xs, ws)) <= 1e-6
    # beyond the largest bf16: non-finite, not wrong-but-finite
    xs2 = xs.clone()
    xs2[0, 0, 0, 0] = 3.4e38
    y2 = run(xs2, ws)
    assert not torch.isfinite(y2[0, :, 0, 0]).any(), "a value above the bf16 range must poison its outputs"
    assert torch.isfinite(y2[1]).all()
    # tiny values
    y3 = run(x * 2.0 ** -100, w)
    assert rel_l2(y3, ref(x * 2.0 ** -100, w)) <= 1e-6
    y4 = run(x * 2.0 ** -118, w)
    assert rel_l2(y4, ref(x * 2.0 ** -118, w)) <= 2.0 ** -12


def test_native_fp32_mfma_build():
    """The second build variant of the same sources (-DMOGAN_X6=0: every MFMA kernel on the native v_mfma_f32_32x32x2_f32,
    libmogan_hip_f32.so, built by __graft_entry__.build()) is the reference point of the precision claims in DESIGN.md
    section 4a; it is not loaded by the product.  Kept under test here: the convolution / up-conv / bmm kernel tests of this
    file run once more in a child process whose MOGAN_LIB points at it."""
    import subprocess
    import sys
    so = os.path.join(ROOT, "multiple-objects-gan_amd", "libmogan_hip_f32.so")
    assert os.path.isfile(so), "libmogan_hip_f32.so is not built (python __graft_entry__.py)"
    code = ("import sys; sys.path.insert(0, %r); from helpers import load_pkg; load_pkg(); from mogan_amd.hip import lib; "
            "assert lib.load().mogan_mfma_form() == 1, 'not the native build'; import pytest; "
            "sys.exit(pytest.main([%r, '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', '-k', "
            "'(test_conv2d_fwd_dgrad_wgrad and (force0 or force1 or force5)) or test_upsample_conv3x3 or test_bmm or test_linear']))"
            % (os.path.join(ROOT, "tests"), os.path.abspath(__file__)))
    env = dict(os.environ, MOGAN_LIB=so)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])
    assert " passed" in out.stdout and "failed" not in out.stdout, out.stdout[-2000:]


PK_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad      (forward needs Cin % 32 == 0, the data gradient Cout % 32 == 0)
    (4, 64, 8, 8, 96, 4, 2, 1),        # down-convolution 8x8 -> 4x4, four parity classes in the data gradient
    (3, 96, 16, 16, 160, 4, 2, 1),     # ragged: N = 3*64 = 192 columns, 160 rows (the last m-tile half empty)
    (16, 128, 4, 4, 64, 3, 1, 1),      # 3x3 s1 on a 4x4 map (jointConv / the last D_NET256 layers), one class
    (5, 32, 6, 10, 32, 4, 2, 1),       # non-square map, one K-tile per tap, N = 5*15 = 75
    (2, 64, 8, 8, 64, 1, 1, 0),        # 1x1
    (15, 64, 4, 4, 32, 3, 1, 1),       # the "wrong pair" batch (B - 1 images): N = 240
]


@pytest.mark.parametrize("case", PK_CASES)
@pytest.mark.parametrize("force", [(-1, 0), (0, 1), (1, 3), (2, 2), (0, 5)])
def test_packed_weight_convolution(case, force):
    """csrc/mogan_pgemm.hip: forward and data gradient from the packed weight copies (mogan_pk_weight_pack,
    mogan_conv2d_fwd_pk, mogan_conv2d_dgrad_pk) against fp64, every tile shape, with and without K-splits; the weight
    gradient of the same autograd node takes the unpacked kernels.  The size heuristic of mogan_pk_conv_eligible is switched
    off (the hard constraints of the panel formats stay) so that small shapes reach the kernels."""
    B, Cin, H, W, Cout, k, s, pad = case
    ops.pk_debug_force(1, force[0], force[1])
    before = dict(ops.PK_STATS)
    try:
        x = T("pkx%s" % (case,), (B, Cin, H, W)).requires_grad_(True)
        w = T("pkw%s" % (case,), (Cout, Cin, k, k), 0.2).requires_grad_(True)
        ref = F.conv2d(x.double(), w.double(), None, s, pad)
        g = T("pkg%s" % (case,), ref.shape)
        ref.backward(g.double())
        xd = x.detach().to(DEV).requires_grad_(True)
        wd = w.detach().to(DEV).requires_grad_(True)
        ops.attach_packs(wd)
        y = ops.conv2d(xd, wd, None, s, pad, False)
        y.backward(g.to(DEV))
        torch.cuda.synchronize()
        assert ops.PK_STATS["fwd"] == before["fwd"] + 1 and ops.PK_STATS["dgrad"] == before["dgrad"] + 1, \
            "the packed path was not taken: %s -> %s" % (before, ops.PK_STATS)
        _check(y, ref, what="fwd")
        _check(xd.grad, x.grad, what="dgrad")
        _check(wd.grad, w.grad, what="wgrad")
        # a changed weight must be re-packed by its owner: same objects, new values
        with torch.no_grad():
            wd.mul_(-0.5)
        wd._mogan_pk.cell[0] += 1                       # what FlatAdam.touch() does: the copies are rebuilt at their next use
        y2 = ops.conv2d_forward(xd.detach(), wd, s, pad, pad, 0)
        torch.cuda.synchronize()
        _check(y2, -0.5 * ref.detach(), what="fwd after re-pack")
        with torch.no_grad():
            wd.mul_(-3.0)
        wd._mogan_pk.cell[0] += 1
        wd._mogan_pk.repack()                           # what FlatAdam.step() does: both copies now, from one read of w
        y3 = ops.conv2d_forward(xd.detach(), wd, s, pad, pad, 0)
        dx3 = ops.conv2d_dgrad(g.to(DEV), wd, xd.shape, s, pad, pad, 0)
        torch.cuda.synchronize()
        _check(y3, 1.5 * ref.detach(), what="fwd after repack()")
        _check(dx3, 1.5 * x.grad, what="dgrad after repack()")
    finally:
        ops.pk_debug_force(0, -1, 0)


@pytest.mark.parametrize("case", [(2, 96, 16, 32, 192), (3, 64, 12, 64, 96), (2, 32, 20, 96, 160), (1, 48, 9, 11, 64)])
def test_winograd_prepared_filter_images(case):
    """Round 6 (include/mogan_hip.h "Prepared filter images"): a weight with an owner (ops.attach_packs) keeps its Winograd filter
    images -- one per direction, built by mogan_wino_prep_group -- and its convolutions go through mogan_conv2d_fwd_wp /
    mogan_conv2d_dgrad_wp; without an owner the same convolution transforms the filters per call.  Both must give the SAME
    bits (same kernels, same image), a changed weight must be re-prepared by its owner (lazily at the next use after touch(),
    all images of a bucket in one launch after step()), and a geometry the Winograd kernels decline gets no image."""
    B, Cin, H, W, Cout = case
    x = T("wpx%s" % (case,), (B, Cin, H, W)).to(DEV)
    w0 = T("wpw%s" % (case,), (Cout, Cin, 3, 3), 0.2).to(DEV)
    g = T("wpg%s" % (case,), (B, Cout, H, W)).to(DEV)
    y_ref, dx_ref = ops.conv2d_forward(x, w0, 1, 1, 1, 0), ops.conv2d_dgrad(g, w0, x.shape, 1, 1, 1, 0)   # no owner: per-call transform
    w = w0.clone()
    pk = ops.attach_packs(w)
    before = ops.PK_STATS.get("wino_preps", 0)
    y, dx = ops.conv2d_forward(x, w, 1, 1, 1, 0), ops.conv2d_dgrad(g, w, x.shape, 1, 1, 1, 0)
    torch.cuda.synchronize()
    filled = H * W >= 0.7 * (-(-H // 4) * 4) * (-(-W // 32) * 32)
    takes = [bool(lib.load().mogan_wino_prep_bytes(B, Cin, H, W, Cout, 3, 3, 1, 1, 1, 0, d)) for d in (0, 1)]
    assert takes[0] == (filled and Cin % 16 == 0 and Cin >= 32 and Cout >= 64)
    assert takes[1] == (filled and Cout % 16 == 0 and Cout >= 32 and Cin >= 64)
    ntk = int(takes[0]) + int(takes[1])
    assert sorted(pk.wino) == [d for d in (0, 1) if takes[d]] and ops.PK_STATS.get("wino_preps", 0) == before + ntk
    assert torch.equal(y, y_ref) and torch.equal(dx, dx_ref)
    y_again = ops.conv2d_forward(x, w, 1, 1, 1, 0)                 # a current image is not rebuilt
    assert ops.PK_STATS.get("wino_preps", 0) == before + ntk and torch.equal(y_again, y_ref)
    takes = ntk > 0
    with torch.no_grad():
        w.mul_(-0.5)
    pk.cell[0] += 1                                               # FlatAdam.touch(): rebuilt at the next use
    y2 = ops.conv2d_forward(x, w, 1, 1, 1, 0)
    assert torch.equal(y2, ops.conv2d_forward(x, w0 * -0.5, 1, 1, 1, 0))
    with torch.no_grad():
        w.mul_(-3.0)
    pk.cell[0] += 1
    n0 = ops.PK_STATS.get("wino_preps", 0)
    ops.repack_all([pk])                                          # FlatAdam.step(): every image in use, one launch
    assert ops.PK_STATS.get("wino_preps", 0) == n0 + (1 if takes else 0)
    y3, dx3 = ops.conv2d_forward(x, w, 1, 1, 1, 0), ops.conv2d_dgrad(g, w, x.shape, 1, 1, 1, 0)
    assert ops.PK_STATS.get("wino_preps", 0) == n0 + (1 if takes else 0)
    assert torch.equal(y3, ops.conv2d_forward(x, w0 * 1.5, 1, 1, 1, 0)) and torch.equal(dx3, ops.conv2d_dgrad(g, w0 * 1.5, x.shape, 1, 1, 1, 0))
    _check(y3, 1.5 * F.conv2d(x.double().cpu(), w0.double().cpu(), None, 1, 1), what="fwd against fp64")


def test_winograd_prepared_images_of_many_weights_in_one_call():
    """mogan_wino_prep_group with more members than one launch holds (32): 20 weights of four shapes, both directions = 40 images"""
    shapes = [(192, 96), (96, 96), (64, 32), (160, 48)]
    ws, pks = [], []
    x = {ci: T("wgx%d" % ci, (2, ci, 8, 32)).to(DEV) for ci in (96, 32, 48)}
    for i in range(20):
        co, ci = shapes[i % 4]
        w = T("wgw%d" % i, (co, ci, 3, 3), 0.2).to(DEV)
        ws.append(w)
        pks.append(ops.attach_packs(w.clone()))
    for pk in pks:                                               # allocate the slots (lazily prepared, one by one)
        ci = pk.w.shape[1]
        ops.conv2d_forward(x[ci], pk.w, 1, 1, 1, 0)
        ops.conv2d_dgrad(torch.zeros(2, pk.w.shape[0], 8, 32, device=DEV), pk.w, x[ci].shape, 1, 1, 1, 0)
    with torch.no_grad():
        for pk in pks:
            pk.w.mul_(2.0)
            pk.cell[0] += 1
    n0 = ops.PK_STATS.get("wino_preps", 0)
    ops.repack_all(pks)
    assert ops.PK_STATS.get("wino_preps", 0) == n0 + 1
    for w, pk in zip(ws, pks):
        ci = w.shape[1]
        assert torch.equal(ops.conv2d_forward(x[ci], pk.w, 1, 1, 1, 0), ops.conv2d_forward(x[ci], w * 2.0, 1, 1, 1, 0))
        gy = T("wgg%d" % w.shape[0], (2, w.shape[0], 8, 32)).to(DEV)
        assert torch.equal(ops.conv2d_dgrad(gy, pk.w, x[ci].shape, 1, 1, 1, 0), ops.conv2d_dgrad(gy, w * 2.0, x[ci].shape, 1, 1, 1, 0))
    assert ops.PK_STATS.get("wino_preps", 0) == n0 + 1


@pytest.mark.parametrize("case", [(2, 96, 64, 192), (3, 64, 32, 96), (2, 192, 32, 384), (2, 96, 16, 96), (2, 40, 64, 72), (1, 64, 8, 64)])
def test_dconv2_prepared_filter_images(case):
    """Round 6, third session (include/mogan_hip.h mogan_conv_prep_bytes / _group): the 4x4 s2 p1 convolutions of the discriminators
    run on dconv2_fwd_kernel (forward over the space-to-depth image, data gradient by parity classes) with pre-split filter images
    that mogan_conv2d_fwd / _dgrad rebuild per call.  A weight with an owner keeps one image per direction instead (the same
    mechanism as the Winograd images): same bits as the per-call form, re-prepared by the owner after a change, both images of many
    weights in one launch per kind, and no image where the dispatch does not take that kernel."""
    B, Cin, H, Cout = case
    x = T("d2x%s" % (case,), (B, Cin, H, H)).to(DEV)
    w0 = T("d2w%s" % (case,), (Cout, Cin, 4, 4), 0.2).to(DEV)
    g = T("d2g%s" % (case,), (B, Cout, H // 2, H // 2)).to(DEV)
    y_ref, dx_ref = ops.conv2d_forward(x, w0, 2, 1, 1, 0), ops.conv2d_dgrad(g, w0, x.shape, 2, 1, 1, 0)     # no owner: per-call prep
    nb = [int(lib.load().mogan_conv_prep_bytes(B, Cin, H, H, Cout, 4, 4, 2, 1, 1, 0, d)) for d in (0, 1)]
    if lib.load().mogan_mfma_form() == 1:                      # the native-fp32 build has no pre-split filter images
        assert nb == [0, 0]
    else:
        # the forward takes dconv2 from 16-pixel output rows and 64 output channels on, the data gradient from 16-pixel parity grids
        assert bool(nb[0]) == (H // 2 >= 16 and Cin % 8 == 0 and Cout >= 64), nb
        assert bool(nb[1]) == (H // 2 >= 16 and Cout % 16 == 0 and Cin >= 64), nb
    w = w0.clone()
    pk = ops.attach_packs(w)
    before = ops.PK_STATS.get("wino_preps", 0)
    y, dx = ops.conv2d_forward(x, w, 2, 1, 1, 0), ops.conv2d_dgrad(g, w, x.shape, 2, 1, 1, 0)
    torch.cuda.synchronize()
    ntk = int(bool(nb[0])) + int(bool(nb[1]))
    assert sorted(pk.wino) == [d for d in (0, 1) if nb[d]] and ops.PK_STATS.get("wino_preps", 0) == before + ntk
    assert all(pk.wino[d][0].numel() >= nb[d] for d in pk.wino)
    assert torch.equal(y, y_ref) and torch.equal(dx, dx_ref)
    assert torch.equal(ops.conv2d_forward(x, w, 2, 1, 1, 0), y_ref) and ops.PK_STATS.get("wino_preps", 0) == before + ntk
    with torch.no_grad():
        w.mul_(-0.5)
    pk.cell[0] += 1                                               # FlatAdam.touch(): rebuilt at the next use
    assert torch.equal(ops.conv2d_forward(x, w, 2, 1, 1, 0), ops.conv2d_forward(x, w0 * -0.5, 2, 1, 1, 0))
    with torch.no_grad():
        w.mul_(-3.0)
    pk.cell[0] += 1
    n0 = ops.PK_STATS.get("wino_preps", 0)
    ops.repack_all([pk])                                          # FlatAdam.step(): every image in use, one launch
    assert ops.PK_STATS.get("wino_preps", 0) == n0 + (1 if ntk else 0)
    y3, dx3 = ops.conv2d_forward(x, w, 2, 1, 1, 0), ops.conv2d_dgrad(g, w, x.shape, 2, 1, 1, 0)
    assert ops.PK_STATS.get("wino_preps", 0) == n0 + (1 if ntk else 0)
    assert torch.equal(y3, ops.conv2d_forward(x, w0 * 1.5, 2, 1, 1, 0)) and torch.equal(dx3, ops.conv2d_dgrad(g, w0 * 1.5, x.shape, 2, 1, 1, 0))
    _check(y3, 1.5 * F.conv2d(x.double().cpu(), w0.double().cpu(), None, 2, 1), what="fwd against fp64")


@pytest.mark.parametrize("case", [(2, 96, 32, 96), (2, 192, 16, 96), (3, 64, 8, 128), (2, 48, 64, 24), (1, 20, 5, 12)])
def test_upconv_filters_kept_by_the_owner(case):
    """nearest-x2 upsample + conv3x3 runs as the transposed 4x4 s2 convolution with the virtual filters K = T w T^t
    (mogan_upconv3x3_*, which build K -- and, where dconv2_fwd_kernel takes the virtual convolution, its filter image -- per call).
    A weight with an owner keeps K and the images per weight version (mogan_upconv3x3_k4 + mogan_conv_prep_group behind the optimizer
    step) and calls mogan_conv2d_dgrad_wp / _fwd_wp on K: the same bits, no per-call transform launches."""
    B, Cin, H, Cout = case
    x = T("ukx%s" % (case,), (B, Cin, H, H)).to(DEV)
    w0 = T("ukw%s" % (case,), (Cout, Cin, 3, 3), 0.2).to(DEV)
    g = T("ukg%s" % (case,), (B, Cout, 2 * H, 2 * H)).to(DEV)
    assert ops.UPCONV4
    y_ref, dx_ref = ops.conv2d_forward(x, w0, 1, 1, 1, 1), ops.conv2d_dgrad(g, w0, x.shape, 1, 1, 1, 1)     # no owner: K per call
    w = w0.clone()
    pk = ops.attach_packs(w)
    k0 = ops.PK_STATS.get("k4_builds", 0)
    y, dx = ops.conv2d_forward(x, w, 1, 1, 1, 1), ops.conv2d_dgrad(g, w, x.shape, 1, 1, 1, 1)
    torch.cuda.synchronize()
    assert ops.PK_STATS.get("k4_builds", 0) == k0 + 1 and pk.k4 is not None        # one K for both directions
    assert torch.equal(y, y_ref) and torch.equal(dx, dx_ref)
    _check(y, F.conv2d(F.interpolate(x.double().cpu(), scale_factor=2, mode="nearest"), w0.double().cpu(), None, 1, 1), what="fwd against fp64")
    n0 = ops.PK_STATS.get("wino_preps", 0)
    assert torch.equal(ops.conv2d_forward(x, w, 1, 1, 1, 1), y_ref)               # current K and images are not rebuilt
    assert ops.PK_STATS.get("k4_builds", 0) == k0 + 1 and ops.PK_STATS.get("wino_preps", 0) == n0
    with torch.no_grad():
        w.mul_(-0.5)
    pk.cell[0] += 1                                               # FlatAdam.touch(): rebuilt at the next use
    assert torch.equal(ops.conv2d_dgrad(g, w, x.shape, 1, 1, 1, 1), ops.conv2d_dgrad(g, w0 * -0.5, x.shape, 1, 1, 1, 1))
    assert ops.PK_STATS.get("k4_builds", 0) == k0 + 2
    with torch.no_grad():
        w.mul_(-3.0)
    pk.cell[0] += 1
    ops.repack_all([pk])                                          # FlatAdam.step()
    k1, n1 = ops.PK_STATS.get("k4_builds", 0), ops.PK_STATS.get("wino_preps", 0)
    assert k1 == k0 + 3
    y3, dx3 = ops.conv2d_forward(x, w, 1, 1, 1, 1), ops.conv2d_dgrad(g, w, x.shape, 1, 1, 1, 1)
    assert ops.PK_STATS.get("k4_builds", 0) == k1 and ops.PK_STATS.get("wino_preps", 0) == n1
    assert torch.equal(y3, ops.conv2d_forward(x, w0 * 1.5, 1, 1, 1, 1)) and torch.equal(dx3, ops.conv2d_dgrad(g, w0 * 1.5, x.shape, 1, 1, 1, 1))


def test_prepared_images_of_both_kinds_in_one_call():
    """ops.repack_all over a bucket with 3x3 (Winograd) and 4x4 s2 (dconv2) weights: one mogan_conv_prep_group call, 36 + 36 images"""
    if lib.load().mogan_mfma_form() == 1:
        pytest.skip("the native-fp32 build has no prepared images")
    pks, refs, xs = [], [], {}
    for i in range(36):
        k = 3 if i % 2 == 0 else 4
        co, ci = [(192, 96), (96, 96), (128, 64)][i % 3]
        w = T("bkw%d" % i, (co, ci, k, k), 0.2).to(DEV)
        refs.append(w)
        pks.append(ops.attach_packs(w.clone()))
        xs[ci] = T("bkx%d" % ci, (2, ci, 32, 32)).to(DEV)

    def run(wt):
        co, ci, k = wt.shape[0], wt.shape[1], wt.shape[2]
        s = 1 if k == 3 else 2
        gy = T("bkg%d_%d" % (co, k), (2, co, 32 // s, 32 // s)).to(DEV)
        return ops.conv2d_forward(xs[ci], wt, s, 1, 1, 0), ops.conv2d_dgrad(gy, wt, xs[ci].shape, s, 1, 1, 0)

    for pk in pks:
        run(pk.w)                                                # allocate the slots
    assert all(sorted(pk.wino) == [0, 1] for pk in pks)
    with torch.no_grad():
        for pk in pks:
            pk.w.mul_(2.0)
            pk.cell[0] += 1
    n0 = ops.PK_STATS.get("wino_preps", 0)
    ops.repack_all(pks)
    assert ops.PK_STATS.get("wino_preps", 0) == n0 + 1
    for w, pk in zip(refs, pks):
        a, b = run(pk.w), run(w * 2.0)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert ops.PK_STATS.get("wino_preps", 0) == n0 + 1


PK_WGRAD_CASES = PK_CASES + [
    (4, 20, 9, 7, 50, 3, 2, 1),        # nothing aligned: Cin, Cout, the map and K = 4*5*4 = 80 output pixels (padded to 96)
    (33, 16, 4, 4, 40, 4, 1, 0),       # 1x1 outputs: K = 33
]


@pytest.mark.parametrize("case", PK_WGRAD_CASES)
@pytest.mark.parametrize("force", [(-1, 0), (0, 2), (1, 1), (2, 3)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_packed_weight_gradient(case, force, accumulate):
    """mogan_conv2d_wgrad_pk: dW = dY x im2col(x)^T on the packed-operand kernel (dY and the transposed im2col matrix packed per
    call) against fp64, every tile shape, with K-splits, written and accumulated."""
    B, Cin, H, W, Cout, k, s, pad = case
    ops.pk_debug_force(1, force[0], force[1])
    before = ops.PK_STATS["wgrad"]
    try:
        x = T("pwx%s" % (case,), (B, Cin, H, W))
        w = T("pww%s" % (case,), (Cout, Cin, k, k), 0.2).requires_grad_(True)
        ref = F.conv2d(x.double(), w.double(), None, s, pad)
        g = T("pwg%s" % (case,), ref.shape)
        ref.backward(g.double())
        base = T("pwb%s" % (case,), w.shape)
        out = base.clone().to(DEV) if accumulate else torch.full(w.shape, float("nan"), device=DEV)
        dw = ops.conv2d_wgrad(g.to(DEV), x.to(DEV), w.shape, s, pad, pad, 0, out=out, accumulate=accumulate)
        torch.cuda.synchronize()
        assert ops.PK_STATS["wgrad"] == before + 1, "the packed weight-gradient path was not taken"
        _check(dw, w.grad + (base.double() if accumulate else 0.0), what="wgrad")
    finally:
        ops.pk_debug_force(0, -1, 0)


def test_packed_weight_gradient_eligibility():
    e = lib.load().mogan_pk_wgrad_eligible
    ws = 256 << 20
    assert e(32, 1536, 8, 8, 3072, 4, 4, 2, 1, 1, ws) == 1 and e(32, 768, 16, 16, 1536, 4, 4, 2, 1, 1, ws) == 1
    assert e(32, 384, 16, 16, 384, 4, 4, 2, 1, 1, ws) == 0           # small dW: the implicit-GEMM kernel is ahead
    assert e(16, 96, 128, 128, 192, 4, 4, 2, 1, 1, ws) == 0          # wide map: the im2col matrix would be the large operand
    assert e(32, 1536, 8, 8, 3072, 4, 4, 2, 1, 1, 1 << 20) == 0      # operands do not fit the workspace
    assert e(4, 1536, 4, 4, 768, 3, 3, 1, 1, 1, ws) == 0             # K = 64 output pixels: not worth two packs


def test_packed_weight_eligibility():
    """mogan_pk_conv_eligible: the deep discriminator layers of the benchmark qualify, wide / thin / misaligned ones do not."""
    e = lib.load().mogan_pk_conv_eligible
    assert e(16, 1536, 8, 8, 3072, 4, 4, 2, 1, 1, 0) == 1 and e(16, 1536, 8, 8, 3072, 4, 4, 2, 1, 1, 1) == 1
    assert e(16, 3072, 4, 4, 1536, 3, 3, 1, 1, 1, 0) == 1 and e(15, 1024, 4, 4, 768, 3, 3, 1, 1, 1, 1) == 1
    assert e(16, 96, 128, 128, 192, 4, 4, 2, 1, 1, 0) == 0          # wide: 64x64 outputs
    assert e(16, 84, 16, 16, 192, 4, 4, 1, 1, 1, 0) == 0            # Cin % 32 != 0
    assert e(16, 768, 8, 8, 100, 4, 4, 2, 1, 1, 1) == 0             # data gradient: Cout % 32 != 0
    assert e(16, 64, 8, 8, 64, 4, 4, 2, 1, 1, 0) == 0               # K = 1024 but only 64 rows


DEEP_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, act
    (16, 64, 8, 8, 64, 4, 2, 1, ops.ACT_LRELU),      # 256 values per channel
    (16, 32, 16, 16, 96, 4, 2, 1, ops.ACT_LRELU),    # 1024
    (3, 64, 4, 4, 96, 3, 1, 1, ops.ACT_RELU),        # 48: ragged half-waves
    (32, 32, 16, 16, 32, 4, 2, 1, ops.ACT_NONE),     # 2048: the largest map the tail kernels take
    (15, 64, 4, 4, 32, 3, 1, 1, ops.ACT_LRELU),      # the "wrong pair" batch
]


@pytest.mark.parametrize("case", DEEP_CASES)
@pytest.mark.parametrize("split", [0, 1, 3])
def test_deep_block_conv_bn_act(case, split):
    """csrc/mogan_pgemm.hip deep block (packed-weight GEMM + one tail kernel each way) = conv2d -> BatchNorm2d(train) ->
    LeakyReLU / ReLU (model.py:575-613) against torch in fp64: output, running statistics, and the gradients of the input,
    the filter, gamma and beta; K-split slabs summed by the tail kernel (split 3) and the direct store (split 1)."""
    import torch.nn as nn
    from mogan_amd.attngan.model_base import FusedSeq, HipBatchNorm2d, HipConv2d
    B, Cin, H, W, Cout, k, s, pad, act = case
    ops.pk_debug_force(1, -1, split)
    before = dict(ops.DEEP_STATS)
    try:
        mods = [HipConv2d(Cin, Cout, k, s, pad, bias=False), HipBatchNorm2d(Cout)]
        if act == ops.ACT_LRELU:
            mods.append(nn.LeakyReLU(0.2))
        elif act == ops.ACT_RELU:
            mods.append(nn.ReLU())
        seq = FusedSeq(*mods)
        with torch.no_grad():
            seq[0].weight.copy_(T("dbw%s" % (case,), (Cout, Cin, k, k), 0.1))
            seq[1].weight.copy_(T("dbg%s" % (case,), (Cout,), 0.3, 1.0))
            seq[1].bias.copy_(T("dbb%s" % (case,), (Cout,), 0.2))
        x = T("dbx%s" % (case,), (B, Cin, H, W))
        # fp64 reference
        xd = x.double().requires_grad_(True)
        wd = seq[0].weight.detach().double().requires_grad_(True)
        gd = seq[1].weight.detach().double().requires_grad_(True)
        bd = seq[1].bias.detach().double().requires_grad_(True)
        rm, rv = torch.zeros(Cout, dtype=torch.float64), torch.ones(Cout, dtype=torch.float64)
        yd = F.conv2d(xd, wd, None, s, pad)
        zd = F.batch_norm(yd, rm, rv, gd, bd, True, 0.1, 1e-5)
        if act == ops.ACT_LRELU:
            zd = F.leaky_relu(zd, 0.2)
        elif act == ops.ACT_RELU:
            zd = F.relu(zd)
        gz = T("dbgz%s" % (case,), zd.shape)
        zd.backward(gz.double())
        # the fused path
        seq = seq.to(DEV).train()
        ops.attach_packs(seq[0].weight)
        xg = x.to(DEV).requires_grad_(True)
        z = seq(xg)
        assert ops.DEEP_STATS["fwd"] == before["fwd"] + 1, "the deep block was not taken"
        z.backward(gz.to(DEV))
        torch.cuda.synchronize()
        assert ops.DEEP_STATS["bwd"] == before["bwd"] + 1
        assert max_abs(z, zd) <= 2e-5
        assert max_abs(seq[1].running_mean, rm) <= 1e-6 and max_abs(seq[1].running_var, rv) <= 1e-5
        assert rel_l2(xg.grad, xd.grad) <= 5e-6, rel_l2(xg.grad, xd.grad)
        assert rel_l2(seq[0].weight.grad, wd.grad) <= 5e-6
        assert max_abs(seq[1].weight.grad, gd.grad) <= 2e-4 * max(1.0, float(gd.grad.abs().max()))
        assert max_abs(seq[1].bias.grad, bd.grad) <= 2e-4 * max(1.0, float(bd.grad.abs().max()))
    finally:
        ops.pk_debug_force(0, -1, 0)


@pytest.mark.parametrize("case", [(16, 64, 8, 8, 64, 4, 2, 1, ops.ACT_LRELU), (16, 32, 16, 16, 96, 4, 2, 1, ops.ACT_LRELU),
                                  (5, 64, 4, 4, 96, 3, 1, 1, ops.ACT_RELU), (16, 64, 4, 4, 32, 3, 1, 1, ops.ACT_NONE)])
@pytest.mark.parametrize("split", [0, 3])
def test_deep_block_two_groups(case, split):
    """The deep block on a [real; fake] batch (groups = 2; miscc/losses.py:136-174 makes two calls): ONE convolution over 2B
    images, BatchNorm statistics per half, running statistics updated half after half -- against fp64 torch making the two
    calls, and against two calls of the one-group deep block (same kernels up to the K-split of the GEMM)."""
    import torch.nn as nn
    from mogan_amd.attngan.model_base import FusedSeq, HipBatchNorm2d, HipConv2d
    B, Cin, H, W, Cout, k, s, pad, act = case
    ops.pk_debug_force(1, -1, split)
    before = dict(ops.DEEP_STATS)
    try:
        def make():
            mods = [HipConv2d(Cin, Cout, k, s, pad, bias=False), HipBatchNorm2d(Cout)]
            if act == ops.ACT_LRELU:
                mods.append(nn.LeakyReLU(0.2))
            elif act == ops.ACT_RELU:
                mods.append(nn.ReLU())
            seq = FusedSeq(*mods)
            with torch.no_grad():
                seq[0].weight.copy_(T("d2w%s" % (case,), (Cout, Cin, k, k), 0.1))
                seq[1].weight.copy_(T("d2g%s" % (case,), (Cout,), 0.3, 1.0))
                seq[1].bias.copy_(T("d2b%s" % (case,), (Cout,), 0.2))
            return seq
        x = torch.cat([T("d2xa%s" % (case,), (B, Cin, H, W)), T("d2xb%s" % (case,), (B, Cin, H, W), 0.7, 0.4)])
        seq = make()
        xd = x.double().requires_grad_(True)
        wd = seq[0].weight.detach().double().requires_grad_(True)
        gd = seq[1].weight.detach().double().requires_grad_(True)
        bd = seq[1].bias.detach().double().requires_grad_(True)
        rm, rv = torch.zeros(Cout, dtype=torch.float64), torch.ones(Cout, dtype=torch.float64)
        outs = []
        for g in range(2):
            zd = F.batch_norm(F.conv2d(xd[g * B:(g + 1) * B], wd, None, s, pad), rm, rv, gd, bd, True, 0.1, 1e-5)
            outs.append(F.leaky_relu(zd, 0.2) if act == ops.ACT_LRELU else (F.relu(zd) if act == ops.ACT_RELU else zd))
        zd = torch.cat(outs)
        gz = T("d2gz%s" % (case,), zd.shape)
        zd.backward(gz.double())
        # one pass, two groups
        seq = seq.to(DEV).train()
        ops.attach_packs(seq[0].weight)
        xg = x.to(DEV).requires_grad_(True)
        z = seq(xg, groups=2)
        assert ops.DEEP_STATS["fwd"] == before["fwd"] + 1, "the grouped deep block was not taken"
        assert getattr(z, "_mogan_panel", None) is not None
        z.backward(gz.to(DEV))
        torch.cuda.synchronize()
        assert ops.DEEP_STATS["bwd"] == before["bwd"] + 1
        assert max_abs(z, zd) <= 2e-5
        assert max_abs(seq[1].running_mean, rm) <= 1e-6 and max_abs(seq[1].running_var, rv) <= 1e-5
        assert int(seq[1].num_batches_tracked) == 2
        assert rel_l2(xg.grad, xd.grad) <= 5e-6, rel_l2(xg.grad, xd.grad)
        assert rel_l2(seq[0].weight.grad, wd.grad) <= 5e-6
        assert max_abs(seq[1].weight.grad, gd.grad) <= 2e-4 * max(1.0, float(gd.grad.abs().max()))
        assert max_abs(seq[1].bias.grad, bd.grad) <= 2e-4 * max(1.0, float(bd.grad.abs().max()))
        # ... and the two calls of the one-group block
        seq2 = make().to(DEV).train()
        ops.attach_packs(seq2[0].weight)
        x2 = x.to(DEV).requires_grad_(True)
        z2 = torch.cat([seq2(x2[:B]), seq2(x2[B:])])
        z2.backward(gz.to(DEV))
        torch.cuda.synchronize()
        assert max_abs(z, z2.double().cpu()) <= 1e-5
        assert max_abs(seq[1].running_mean, seq2[1].running_mean.double().cpu()) <= 1e-7
        assert max_abs(seq[1].running_var, seq2[1].running_var.double().cpu()) <= 1e-6
        assert rel_l2(xg.grad, x2.grad.double().cpu()) <= 2e-6
        assert rel_l2(seq[0].weight.grad, seq2[0].weight.grad.double().cpu()) <= 2e-6
        # the next deep block takes the 2B panel: same result as from the plain tensor
        nxt = FusedSeq(HipConv2d(Cout, 64, 3, 1, 1, bias=False), HipBatchNorm2d(64), nn.LeakyReLU(0.2)).to(DEV).train()
        ops.attach_packs(nxt[0].weight)
        if Cout % 32 == 0 and z.shape[2] * z.shape[3] <= 64:
            hits = ops.DEEP_STATS["panel_hits"]
            a1 = nxt(z, groups=2)
            assert ops.DEEP_STATS["panel_hits"] == hits + 1
            a2 = nxt(z.detach().clone(), groups=2)
            torch.cuda.synchronize()
            assert torch.equal(a1, a2)
    finally:
        ops.pk_debug_force(0, -1, 0)


def test_deep_blocks_hand_over_their_pixel_panel():
    """Two deep blocks in a row: the second takes the first one's pixel panel (no activation pack), and gets the same result
    as from the plain tensor."""
    import torch.nn as nn
    from mogan_amd.attngan.model_base import FusedSeq, HipBatchNorm2d, HipConv2d
    ops.pk_debug_force(1, -1, 0)
    try:
        torch.manual_seed(11)
        a = FusedSeq(HipConv2d(64, 96, 4, 2, 1, bias=False), HipBatchNorm2d(96), nn.LeakyReLU(0.2)).to(DEV).train()
        b = FusedSeq(HipConv2d(96, 64, 3, 1, 1, bias=False), HipBatchNorm2d(64), nn.LeakyReLU(0.2)).to(DEV).train()
        for m in (a, b):
            ops.attach_packs(m[0].weight)
        x = torch.randn(8, 64, 8, 8, device=DEV)
        hits = ops.DEEP_STATS["panel_hits"]
        h = a(x)
        z1 = b(h)
        assert ops.DEEP_STATS["panel_hits"] == hits + 1, "the second block did not take the first one's panel"
        z2 = b(h.detach().clone())                       # same values, no panel attached
        assert ops.DEEP_STATS["panel_hits"] == hits + 1
        torch.cuda.synchronize()
        assert torch.equal(z1, z2)
    finally:
        ops.pk_debug_force(0, -1, 0)


# ---------------------------------------------------------------------------------------------------------------------------------
# Frozen trunk on pixel panels: mogan_pk_group (grouped packed-operand GEMM) + mogan_panel_tail_group, straight through the C ABI
def _panel_decode(panel, Q, Cp):
    """pixel panel bytes -> (Q, Cp) fp32: the three bf16 pieces of every value add up to it exactly (csrc/mogan_mma.h)"""
    p = panel.view(torch.bfloat16).view(Q, Cp // 32, 3, 32).float()
    return (p[:, :, 0] + p[:, :, 1] + p[:, :, 2]).reshape(Q, Cp)


def _tail_member(B, n, H, W, srcs, dst=None, dst_bs=0, panel=None, Cp=0, c0=0, scale=None, shift=None, relu=0, box=0, add=None,
                 add_bs=0, mask=None, mask_bs=0):
    a = lib.TailArgs()
    for j, (t, bs, slab, ns) in enumerate(srcs):
        a.src[j], a.src_bs[j], a.src_slab[j], a.src_nsplit[j] = t, bs, slab, ns
    a.nsrc, a.B, a.n, a.H, a.W, a.relu, a.box = len(srcs), B, n, H, W, relu, box
    if dst is not None:
        a.dst, a.dst_bs = dst, dst_bs
    if panel is not None:
        a.panel, a.CGp, a.cg0 = panel, Cp // 32, c0 // 32
    if scale is not None:
        a.scale, a.shift = scale, shift
    if add is not None:
        a.add, a.add_bs = add, add_bs
    if mask is not None:
        a.mask, a.mask_bs = mask, mask_bs
    return a


def _run_tail(members):
    import ctypes
    arr = (lib.TailArgs * len(members))(*members)
    lib.call("mogan_panel_tail_group", len(members), ctypes.cast(arr, ctypes.c_void_p), lib.stream_ptr())


PANEL_CONVS = [  # Cin, c0 (slice offset in a wider panel), Cp_total, H, W, Cout, KH, KW, stride, ph, pw
    (48, 32, 128, 9, 11, 64, 5, 5, 1, 2, 2), (64, 0, 64, 17, 17, 40, 1, 7, 1, 0, 3), (96, 64, 192, 17, 17, 96, 7, 1, 1, 3, 0),
    (160, 0, 160, 8, 8, 200, 3, 3, 1, 1, 1), (288, 0, 288, 13, 13, 72, 3, 3, 2, 0, 0), (192, 0, 224, 8, 8, 136, 1, 1, 1, 0, 0)]


@pytest.mark.parametrize("nsplit", [1, 3])
def test_panel_group_gemm_forward_and_tail(nsplit):
    """All PANEL_CONVS as ONE grouped launch + ONE tail launch: input slices inside wider panels (channel offset, a slice of 48
    channels padded to 64 with zero filters), 1x7 / 7x1 / 5x5 / stride-2 geometries, K-split slabs; y = relu(scale * conv + shift)
    against fp64, the fp32 copy and the decoded output panel (exact split, zero pad channels)."""
    import ctypes
    B = 3
    L = lib.load()
    keep, pk, tails, refs = [], [], [], []
    for i, (Cin, c0, Cp, H, W, Cout, KH, KW, s, ph, pw) in enumerate(PANEL_CONVS):
        x = T("pgx%d" % i, (B, Cp, H, W)).to(DEV)
        w = T("pgw%d" % i, (Cout, Cin, KH, KW), 0.2).to(DEV)
        sc, sh = T("pgs%d" % i, (Cout,), 0.5, 1.0).to(DEV), T("pgh%d" % i, (Cout,), 0.3).to(DEV)
        Q = B * H * W
        xp = torch.empty(Q * Cp * 6, dtype=torch.uint8, device=DEV)
        _run_tail([_tail_member(B, Cp, H, W, [(x.data_ptr(), Cp * H * W, 0, 1)], panel=xp.data_ptr(), Cp=Cp)])
        assert torch.equal(_panel_decode(xp, Q, Cp), x.permute(0, 2, 3, 1).reshape(Q, Cp)), "panel split is not exact"
        cin_p = (Cin + 31) // 32 * 32
        wpad = torch.zeros(Cout, cin_p, KH, KW, device=DEV)
        wpad[:, :Cin] = w
        wpk = torch.empty(L.mogan_pk_weight_bytes(Cout, cin_p, KH, KW, 1, 0), dtype=torch.uint8, device=DEV)
        lib.call("mogan_pk_weight_pack", wpad.data_ptr(), wpk.data_ptr(), Cout, cin_p, KH, KW, 1, ph, pw, 0, lib.stream_ptr())
        OH, OW = (H + 2 * ph - KH) // s + 1, (W + 2 * pw - KW) // s + 1
        raw = torch.full((nsplit, B, Cout, OH, OW), float("nan"), device=DEV)
        a = lib.PkArgs()
        a.wpk, a.panel, a.raw = wpk.data_ptr(), xp.data_ptr(), raw.data_ptr()
        a.B, a.M, a.Cp, a.CGp, a.cg0, a.PH, a.PW, a.outH, a.outW = B, Cout, cin_p, Cp // 32, c0 // 32, H, W, OH, OW
        a.KH, a.KW, a.stride, a.ph, a.pw, a.dgrad, a.nsplit = KH, KW, s, ph, pw, 0, nsplit
        pk.append(a)
        Cop = (Cout + 31) // 32 * 32 + 32                         # output panel: the slice sits behind one foreign group
        y32 = torch.zeros(B, Cout, OH, OW, device=DEV)
        yp = torch.full((B * OH * OW * Cop * 6,), 0x7f, dtype=torch.uint8, device=DEV)
        keep += [x, xp, wpad, wpk, raw, sc, sh, y32, yp]
        tails.append((raw, Cout, OH, OW, sc, sh, y32, yp, Cop))
        # channels of the slice beyond Cin (48 -> 64) hold data in the panel: the zero filters must silence them
        refs.append(torch.relu(F.conv2d(x[:, c0:c0 + Cin].double().cpu(), w.double().cpu(), None, s, (ph, pw))
                               * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1)))
    arr = (lib.PkArgs * len(pk))(*pk)
    lib.call("mogan_pk_group", len(pk), ctypes.cast(arr, ctypes.c_void_p), lib.stream_ptr())
    members = []
    for a, (raw, Cout, OH, OW, sc, sh, y32, yp, Cop) in zip(arr, tails):
        assert 1 <= a.nsplit <= nsplit
        members.append(_tail_member(B, Cout, OH, OW, [(raw.data_ptr(), Cout * OH * OW, B * Cout * OH * OW, a.nsplit)],
                                    dst=y32.data_ptr(), dst_bs=Cout * OH * OW, panel=yp.data_ptr(), Cp=Cop, c0=32,
                                    scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1))
    _run_tail(members)
    torch.cuda.synchronize()
    for (raw, Cout, OH, OW, sc, sh, y32, yp, Cop), ref, case in zip(tails, refs, PANEL_CONVS):
        _check(y32, ref, what="panel conv %s" % (case,))
        dec = _panel_decode(yp, B * OH * OW, Cop)
        n_up = (Cout + 31) // 32 * 32
        assert torch.equal(dec[:, 32:32 + Cout], y32.permute(0, 2, 3, 1).reshape(-1, Cout)), case
        assert float(dec[:, 32 + Cout:32 + n_up].abs().max() if n_up > Cout else 0.0) == 0.0, "pad channels of the slice must be zero"
        assert torch.equal(yp.view(-1, Cop // 32, 192)[:, 0], torch.full_like(yp.view(-1, Cop // 32, 192)[:, 0], 0x7f)), \
            "the foreign channel group in front of the slice was touched"


@pytest.mark.parametrize("case", [c for c in PANEL_CONVS if c[8] == 1])
def test_panel_group_gemm_data_gradient_and_tail(case):
    """stride-1 data gradient from the dY panel (two members that add into one gradient = two K ranges of one sum), + the gradient's
    other contributor, ReLU mask of the input: against the fp64 transposed convolution"""
    import ctypes
    Cin, _, _, H, W, Cout, KH, KW, s, ph, pw = case
    B = 2
    L = lib.load()
    x = torch.relu(T("pdx", (B, Cin, H, W))).to(DEV)                      # the layer input (a ReLU output: exact zeros)
    other = T("pdo", (B, Cin, H, W)).to(DEV)
    coutp = (Cout + 31) // 32 * 32
    raws, keep, ref = [], [], other.double().cpu()
    pk = []
    for j in range(2):
        w = T("pdw%d" % j, (Cout, Cin, KH, KW), 0.2).to(DEV)
        dy = T("pdy%d" % j, (B, Cout, H, W)).to(DEV)
        Q = B * H * W
        dyp = torch.empty(Q * coutp * 6, dtype=torch.uint8, device=DEV)
        _run_tail([_tail_member(B, Cout, H, W, [(dy.data_ptr(), Cout * H * W, 0, 1)], panel=dyp.data_ptr(), Cp=coutp)])
        wpad = torch.zeros(coutp, Cin, KH, KW, device=DEV)
        wpad[:Cout] = w
        wpk = torch.empty(L.mogan_pk_weight_bytes(coutp, Cin, KH, KW, 1, 1), dtype=torch.uint8, device=DEV)
        lib.call("mogan_pk_weight_pack", wpad.data_ptr(), wpk.data_ptr(), coutp, Cin, KH, KW, 1, ph, pw, 1, lib.stream_ptr())
        raw = torch.full((2, B, Cin, H, W), float("nan"), device=DEV)
        a = lib.PkArgs()
        a.wpk, a.panel, a.raw = wpk.data_ptr(), dyp.data_ptr(), raw.data_ptr()
        a.B, a.M, a.Cp, a.CGp, a.cg0, a.PH, a.PW, a.outH, a.outW = B, Cin, coutp, coutp // 32, 0, H, W, H, W
        a.KH, a.KW, a.stride, a.ph, a.pw, a.dgrad, a.nsplit = KH, KW, 1, ph, pw, 1, 2
        pk.append(a)
        raws.append(raw)
        keep += [w, dy, dyp, wpad, wpk]
        ref = ref + F.conv_transpose2d(dy.double().cpu(), w.double().cpu(), None, 1, (ph, pw))
    ref = ref * (x.double().cpu() > 0)
    arr = (lib.PkArgs * 2)(*pk)
    lib.call("mogan_pk_group", 2, ctypes.cast(arr, ctypes.c_void_p), lib.stream_ptr())
    dx = torch.empty(B, Cin, H, W, device=DEV)
    cinp = (Cin + 31) // 32 * 32
    dxp = torch.empty(B * H * W * cinp * 6, dtype=torch.uint8, device=DEV)
    srcs = [(r.data_ptr(), Cin * H * W, B * Cin * H * W, a.nsplit) for r, a in zip(raws, arr)]
    _run_tail([_tail_member(B, Cin, H, W, srcs, dst=dx.data_ptr(), dst_bs=Cin * H * W, panel=dxp.data_ptr(), Cp=cinp,
                            add=other.data_ptr(), add_bs=Cin * H * W, mask=x.data_ptr(), mask_bs=Cin * H * W)])
    torch.cuda.synchronize()
    _check(dx, ref, what="panel dgrad %s" % (case,))
    assert torch.equal(_panel_decode(dxp, B * H * W, cinp)[:, :Cin], dx.permute(0, 2, 3, 1).reshape(-1, Cin))


def test_panel_tail_box_filter():
    """the pool branch's 3x3 mean (zero padding, divisor 9 = F.avg_pool2d(x, 3, 1, 1)) over the sum of two slabs, then affine + ReLU"""
    B, C, H, W = 3, 40, 8, 17
    slabs = T("ptb", (2, B, C, H, W)).to(DEV)
    sc, sh = T("pts", (C,), 0.5, 1.0).to(DEV), T("pth", (C,), 0.3).to(DEV)
    y = torch.empty(B, C, H, W, device=DEV)
    _run_tail([_tail_member(B, C, H, W, [(slabs.data_ptr(), C * H * W, B * C * H * W, 2)], dst=y.data_ptr(), dst_bs=C * H * W,
                            scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1, box=1)])
    ref = torch.relu(F.avg_pool2d(slabs.double().cpu().sum(0), 3, 1, 1) * sc.double().cpu().view(1, -1, 1, 1)
                     + sh.double().cpu().view(1, -1, 1, 1))
    assert max_abs(y.cpu().double(), ref) <= 2e-6


@pytest.mark.parametrize("case", [(16, 12, [12, 12, 11, 10, 9, 9, 8, 8, 7, 7, 6, 6, 5, 5, 5, 5]), (6, 18, [15, 11, 9, 9, 6, 1]),
                                  (1, 12, [12]), (3, 32, [32, 20, 2])])
def test_text_encoder_as_one_launch(case):
    """csrc/mogan_lstm.hip (mogan_lstm_encoder_fwd): RNN_ENCODER.forward in eval mode -- embedding, packed bidirectional LSTM,
    unpacking, the transposes (model.py:183-204) -- as one launch against the stock nn.Embedding / nn.LSTM path of the same module
    on the device (MIOpen) and against torch on the CPU in fp64: zero / non-zero initial state, T_max < T, a one-word caption."""
    from mogan_amd.attngan import model
    from mogan_amd.attngan.miscc.config import cfg
    B, Tw, lens = case
    cfg.RNN_TYPE = 'LSTM'
    torch.manual_seed(5 + B)
    enc = model.RNN_ENCODER(300, nhidden=256).to(DEV).eval()
    cap = torch.zeros(B, Tw, dtype=torch.int64)
    for i, n in enumerate(lens):
        cap[i, :n] = torch.randint(1, 300, (n,))
    cap = cap.to(DEV)
    for zero_state in (True, False):
        hid = enc.init_hidden(B)
        if not zero_state:
            hid = tuple(T("lstmh%d_%d" % (k, B), tuple(h.shape), 0.5).to(DEV) for k, h in enumerate(hid))
        with torch.no_grad():
            n0 = ops.PK_STATS.get("lstm_fused", 0)
            w1, s1 = enc(cap, torch.tensor(lens), hid)
            assert ops.PK_STATS.get("lstm_fused", 0) == n0 + 1
            model.RNN_ENCODER.FUSED = False
            try:
                w0, s0 = enc(cap, torch.tensor(lens), hid)
            finally:
                model.RNN_ENCODER.FUSED = True
            assert ops.PK_STATS.get("lstm_fused", 0) == n0 + 1
            enc64 = model.RNN_ENCODER(300, nhidden=256).double().eval()
            enc64.load_state_dict({k: v.double().cpu() for k, v in enc.state_dict().items()})
            w64, s64 = enc64(cap.cpu(), torch.tensor(lens), tuple(h.double().cpu() for h in hid))
        assert tuple(w1.shape) == tuple(w0.shape) == (B, 256, max(lens)) and tuple(s1.shape) == tuple(s0.shape) == (B, 256)
        for got, ref in ((w1, w64), (s1, s64), (w0, w64), (s0, s64)):
            assert float((got.double().cpu() - ref).abs().max()) <= 5e-6
        assert float((w1 - w0).abs().max()) <= 5e-6 and float((s1 - s0).abs().max()) <= 5e-6
        for i, n in enumerate(lens):                                   # exact zeros behind every caption's end
            assert float(w1[i, :, n:].abs().max()) == 0.0 if n < max(lens) else True
    with torch.enable_grad():                                          # gradients asked for: the stock modules
        n0 = ops.PK_STATS.get("lstm_fused", 0)
        enc(cap, torch.tensor(lens), enc.init_hidden(B))
        assert ops.PK_STATS.get("lstm_fused", 0) == n0
