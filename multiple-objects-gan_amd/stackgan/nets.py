"""StackGAN-style generators / discriminators with the object pathway, shared by the three sibling
trees of the reference that use them:

    coco-stackgan   code/coco/stackgan/model.py      (STAGE1_G/D :147-307, STAGE2_G/D :311-537)
    clevr           code/clevr/model.py              (STAGE1_G/D :113-260)
    multi-mnist     code/multi-mnist/model.py        (STAGE1_G/D :113-257)

The three files differ only in widths, the label dimension, whether a caption embedding conditions
the generator and a few hard-coded constants; `Variant` carries exactly those differences and the
per-tree `model.py` modules bind it to their own global `cfg`, so the class names, constructor
signatures, forward signatures, return structure and state_dict keys are the reference's.

Everything runs on libmogan_hip.so (FusedSeq -> fused conv / BN+act launches); there is no CPU path.
"""
import os
from dataclasses import dataclass

import torch
import torch.nn as nn

from ..attngan.model_base import FusedSeq, HipBatchNorm1d, HipBatchNorm2d, HipConv2d, HipLinear
from ..hip import ops


@dataclass(frozen=True)
class Variant:
    name: str
    img_ch: int            # 3 (coco, clevr) / 1 (multi-mnist)
    label_dim: int         # one-hot width of a bbox label: 81 / 13 / 10
    max_objects: int       # 3 / 4 / 3
    text: bool             # caption embedding -> CA_NET -> c_code (coco only)
    label_net: bool        # generator embeds the bbox label with `self.label` (coco, clevr); mnist feeds the one-hot
    fixed_ef: int = 0      # mnist hard-codes ef_dim = 10 (multi-mnist/model.py:117,197)
    bbox_cdim: int = 0     # mnist hard-codes BBOX_NET c_dim = 128 (multi-mnist/model.py:83)


COCO = Variant("coco-stackgan", 3, 81, 3, True, True)
CLEVR = Variant("clevr", 3, 13, 4, False, True)
MNIST = Variant("multi-mnist", 1, 10, 3, False, False, fixed_ef=10, bbox_cdim=128)


def conv3x3(in_planes, out_planes, stride=1):
    return HipConv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv4x4(in_planes, out_planes, stride, bias=False):
    return HipConv2d(in_planes, out_planes, kernel_size=4, stride=stride, padding=1, bias=bias)


def upBlock(in_planes, out_planes):
    """nearest x2 -> conv3x3 -> BN -> ReLU (S/model.py:16-22): upsample fused into the conv gather,
    BN statistics, one BN+ReLU apply."""
    return FusedSeq(nn.Upsample(scale_factor=2, mode='nearest'), conv3x3(in_planes, out_planes),
                    HipBatchNorm2d(out_planes), nn.ReLU(True))


class ResBlock(nn.Module):
    """S/model.py:25-41: conv-BN-ReLU-conv-BN, += x, ReLU (the add rides on the second BN apply)."""

    def __init__(self, channel_num):
        super(ResBlock, self).__init__()
        self.block = FusedSeq(conv3x3(channel_num, channel_num), HipBatchNorm2d(channel_num), nn.ReLU(True),
                              conv3x3(channel_num, channel_num), HipBatchNorm2d(channel_num))
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return ops.act(self.block(x, residual=x), ops.ACT_RELU, 0.0)


class CA_NET(nn.Module):
    """S/model.py:44-72: Linear(t_dim -> 2c)+ReLU -> (mu, logvar); c = eps*exp(logvar/2)+mu.
    `eps` may be injected (parity tests, graph capture); otherwise drawn here like the reference."""

    def __init__(self, cfg):
        super(CA_NET, self).__init__()
        self.t_dim = cfg.TEXT.DIMENSION
        self.c_dim = cfg.GAN.CONDITION_DIM
        self.fc = HipLinear(self.t_dim, self.c_dim * 2, bias=True)
        self.relu = nn.ReLU()

    def encode(self, text_embedding):
        x = ops.act(self.fc(text_embedding), ops.ACT_RELU, 0.0)
        return x[:, :self.c_dim], x[:, self.c_dim:]

    def reparametrize(self, mu, logvar, eps=None):
        if eps is None:
            eps = torch.randn_like(mu)
        return ops.reparam(mu, logvar, eps)

    def forward(self, text_embedding, eps=None):
        mu, logvar = self.encode(text_embedding)
        return self.reparametrize(mu, logvar, eps), mu, logvar


class D_GET_LOGITS(nn.Module):
    """S/model.py:75-104, C/model.py:44-71, M/model.py:43-70.  Raw logits (the losses use
    BCEWithLogits).  `cond_dim` is the width of the conditioning vector concatenated at 4x4
    (coco: nef; clevr: 13; mnist: 10); `uncond_in` reproduces clevr's hard-coded 109-channel head."""

    def __init__(self, ndf, nef, bcondition=True, cond_dim=None, uncond_in=None):
        super(D_GET_LOGITS, self).__init__()
        self.df_dim, self.ef_dim, self.bcondition = ndf, nef, bcondition
        self.cond_dim = nef if cond_dim is None else cond_dim
        if bcondition:
            self.outlogits = FusedSeq(conv3x3(ndf * 8 + self.cond_dim, ndf * 8), HipBatchNorm2d(ndf * 8),
                                      nn.LeakyReLU(0.2, inplace=True),
                                      HipConv2d(ndf * 8, 1, kernel_size=4, stride=4))
        else:
            self.outlogits = FusedSeq(HipConv2d(ndf * 8 if uncond_in is None else uncond_in, 1,
                                                kernel_size=4, stride=4))

    def forward(self, h_code, c_code=None):
        if self.bcondition and c_code is not None:
            h_code = ops.cat_channels([(h_code, "full"), (c_code.reshape(-1, self.cond_dim), "plane")], h_code.shape[0],
                                      tuple(h_code.shape[2:]))
        return self.outlogits(h_code).view(-1)


# The reference's python loops over the objects as ONE batch of K*B samples (object-major) with per-object BatchNorm statistics
# (FusedSeq(..., groups=K): SURVEY F11), as in attngan/model.py; MOGAN_OBJ_BATCH=0 = the literal loops (same results; A/B switch)
BATCH_OBJECTS = True     # (module attribute: False = the literal per-object loops; tests compare both)


def _objects_first(t, K):
    """(B, >= K, ...) per-object tensor -> (K*B, ...), the objects as the slow batch index"""
    v = t[:, :K].transpose(0, 1)
    return v.reshape((K * t.shape[0],) + tuple(t.shape[2:]))


def _repeat_batch(t, K):
    """(B, ...) -> (K*B, ...): the same B samples once per object"""
    return t.unsqueeze(0).expand((K,) + tuple(t.shape)).reshape((K * t.shape[0],) + tuple(t.shape[1:]))


def _stn_objects(x, theta, K, out_hw, align, plane=False):
    """the per-object transformer calls of a loop as one object-major batch of K*B samples without materialised inputs
    (hip/ops.stn_shared; see attngan/model.py:_stn_objects): x is (K*B, C, H, W), (B, C, H, W) -- one image batch shared by the
    objects -- or, plane, (K*B, C): a label vector standing for its repetition over the plane; theta (B, >= K, 2, 3)"""
    Bp = theta.shape[0]
    if theta.shape[1] != K:
        theta = theta[:, :K].contiguous()
    in_hw = out_hw if plane else tuple(x.shape[2:])
    return ops.stn_shared(x, theta, K * Bp, in_hw, out_hw, align, plane=plane, theta_G=K)


def _obj(t, K):
    """(B, >= K, ...) per-object tensor as a part of ops.cat_channels"""
    return t if t.shape[1] == K else t[:, :K].contiguous()


def _tile(vec, size):
    """(B,C) -> (B,C,size,size) contiguous (label replicated spatially before the STN paste)."""
    B, C = vec.shape
    return vec.reshape(B, C, 1, 1).expand(B, C, size, size)


class BBOX_NET(nn.Module):
    """Layout encoder: every object's label vector is painted into its bbox on a 16x16 canvas (STN
    with the inverse matrix), the canvases are summed and three stride-2 convs reduce the layout to
    c/8 x 2 x 2 (S/model.py:114-142, C/model.py:80-111, M/model.py:80-110)."""

    def __init__(self, cfg, in_dim, c_dim):
        super(BBOX_NET, self).__init__()
        self.cfg, self.in_dim, self.c_dim = cfg, in_dim, c_dim
        c = c_dim
        self.encode = FusedSeq(
            conv3x3(in_dim, c // 2, stride=2), nn.LeakyReLU(0.2, inplace=True),
            conv3x3(c // 2, c // 4, stride=2), HipBatchNorm2d(c // 4), nn.LeakyReLU(0.2, inplace=True),
            conv3x3(c // 4, c // 8, stride=2), HipBatchNorm2d(c // 8), nn.LeakyReLU(0.2, inplace=True))
        self.out_dim = (c // 8) * 4

    def forward(self, labels, transf_matr_inv, max_objects):
        B, K = labels.shape[0], max_objects
        if BATCH_OBJECTS:
            lab = _stn_objects(_objects_first(labels, K), transf_matr_inv, K, (16, 16), bool(self.cfg.STN_ALIGN_CORNERS), plane=True)
            return self.encode(ops.group_sum(lab, K)).view(B, -1)
        layout = None
        for idx in range(max_objects):
            lab = ops.stn(_tile(labels[:, idx], 16), transf_matr_inv[:, idx], (B, self.in_dim, 16, 16),
                          bool(self.cfg.STN_ALIGN_CORNERS))
            layout = lab if layout is None else ops.add(layout, lab)
        return self.encode(layout).view(B, -1)


# ------------------------------------------------------------------------------------- stage I
class STAGE1_G(nn.Module):
    """64x64 generator: object pathway (label -> 4x4 -> two upBlocks -> STN paste at 16x16, summed over
    objects; one BN call per object), layout encoding, global pathway, concat at 16x16."""

    def __init__(self, cfg, variant):
        super(STAGE1_G, self).__init__()
        self.cfg, self.variant = cfg, variant
        self.gf_dim = cfg.GAN.GF_DIM * 8
        self.ef_dim = variant.fixed_ef or cfg.GAN.CONDITION_DIM
        self.z_dim = cfg.Z_DIM
        self.define_module()

    def _align(self):
        return bool(self.cfg.STN_ALIGN_CORNERS)

    def define_module(self):
        v, ngf = self.variant, self.gf_dim
        ninput = self.z_dim + (self.ef_dim if v.text else 0)
        linput = (self.ef_dim if v.text else 0) + v.label_dim
        if v.text:
            self.ca_net = CA_NET(self.cfg)
        if self.cfg.USE_BBOX_LAYOUT:
            self.bbox_net = BBOX_NET(self.cfg, self.ef_dim, v.bbox_cdim or self.cfg.GAN.CONDITION_DIM)
            ninput += self.bbox_net.out_dim
        self.fc = FusedSeq(HipLinear(ninput, ngf * 4 * 4, bias=False), HipBatchNorm1d(ngf * 4 * 4), nn.ReLU(True))
        self.label = FusedSeq(HipLinear(linput, self.ef_dim, bias=False), HipBatchNorm1d(self.ef_dim),
                              nn.ReLU(True))
        self.local1 = upBlock(self.ef_dim, ngf // 2)
        self.local2 = upBlock(ngf // 2, ngf // 4)
        self.upsample1 = upBlock(ngf, ngf // 2)
        self.upsample2 = upBlock(ngf // 2, ngf // 4)
        self.upsample3 = upBlock(ngf // 2, ngf // 8)
        self.upsample4 = upBlock(ngf // 8, ngf // 16)
        self.img = FusedSeq(conv3x3(ngf // 16, v.img_ch), nn.Tanh())

    def generate(self, c_code, noise, transf_matrices_inv, label_one_hot, max_objects):
        """-> (fake_img, local_labels (B,K,ef))"""
        v, B, K = self.variant, noise.shape[0], max_objects
        if BATCH_OBJECTS:
            oh = [(_obj(label_one_hot, K), ("obj", K))]
            if v.label_net:                                  # (c_code repeated for the K objects) | one-hot, object-major
                lab = self.label(ops.cat_channels(oh if c_code is None else [(c_code, ("rep", K))] + oh, K * B), groups=K)
            else:
                lab = ops.cat_channels(oh, K * B)            # the one-hot labels themselves, object-major
            h = self.local2(self.local1(ops.cat_channels([(lab, "plane")], K * B, (4, 4)), groups=K), groups=K)
            h = _stn_objects(h, transf_matrices_inv, K, tuple(h.shape[2:]), self._align())
            canvas = ops.group_sum(h, K)
            local_labels = lab.view(K, B, -1).transpose(0, 1)
        else:
            labels, canvas = [], None
            for idx in range(max_objects):
                if v.label_net:
                    src = label_one_hot[:, idx] if c_code is None else torch.cat((c_code, label_one_hot[:, idx]), 1)
                    lab = self.label(src)
                else:
                    lab = label_one_hot[:, idx]
                labels.append(lab)
                h = self.local2(self.local1(_tile(lab, 4)))
                h = ops.stn(h, transf_matrices_inv[:, idx], tuple(h.shape), self._align())
                canvas = h if canvas is None else ops.add(canvas, h)
            local_labels = torch.stack(labels, 1)
        parts = [noise] + ([c_code] if c_code is not None else [])
        if self.cfg.USE_BBOX_LAYOUT:
            parts.append(self.bbox_net(local_labels, transf_matrices_inv, max_objects))
        z_c_code = parts[0] if len(parts) == 1 else ops.cat_channels([(t, "full") for t in parts], B)
        h_code = self.fc(z_c_code).view(-1, self.gf_dim, 4, 4)
        h_code = self.upsample2(self.upsample1(h_code))
        h_code = ops.cat_channels([(h_code, "full"), (canvas, "full")], B, tuple(h_code.shape[2:]))
        h_code = self.upsample4(self.upsample3(h_code))
        return self.img(h_code), local_labels


class STAGE1_D(nn.Module):
    """64x64 discriminator: per object STN-crop the image to 16x16, concat the tiled label, conv4x4 s1
    (-> 15x15) + BN + LeakyReLU, STN-paste back to 16x16, sum; concat with the global conv features at
    16x16 (S/model.py:238-307, C/model.py:194-260, M/model.py:192-257)."""

    def __init__(self, cfg, variant, cond_dim=None, uncond_in=None):
        super(STAGE1_D, self).__init__()
        self.cfg, self.variant = cfg, variant
        self.df_dim = cfg.GAN.DF_DIM
        self.ef_dim = variant.fixed_ef or cfg.GAN.CONDITION_DIM
        ndf, v = self.df_dim, variant
        self.local = FusedSeq(conv4x4(v.img_ch + v.label_dim, ndf * 2, 1), HipBatchNorm2d(ndf * 2),
                              nn.LeakyReLU(0.2, inplace=True))
        self.act = nn.LeakyReLU(0.2, inplace=True)
        self.conv1 = conv4x4(v.img_ch, ndf, 2)
        self.conv2 = conv4x4(ndf, ndf * 2, 2)
        self.bn2 = HipBatchNorm2d(ndf * 2)
        self.conv3 = conv4x4(ndf * 4, ndf * 4, 2)
        self.bn3 = HipBatchNorm2d(ndf * 4)
        self.conv4 = conv4x4(ndf * 4, ndf * 8, 2)
        self.bn4 = HipBatchNorm2d(ndf * 8)
        self.get_cond_logits = D_GET_LOGITS(ndf, self.ef_dim, True, cond_dim, uncond_in)
        self.get_uncond_logits = None

    def _align(self):
        return bool(self.cfg.STN_ALIGN_CORNERS)

    def _encode_img(self, image, label, transf_matrices, transf_matrices_inv, max_objects):
        B, v, ndf = image.shape[0], self.variant, self.df_dim
        canvas = self._object_canvas(image, label, transf_matrices, transf_matrices_inv, max_objects, 16)
        h = ops.conv2d_lrelu(image, self.conv1.weight, self.conv1.stride[0], self.conv1.padding, 0.2)
        h = self.bn2.fused(self.conv2(h), ops.ACT_LRELU, 0.2)
        h = ops.cat_channels([(h, "full"), (canvas, "full")], B, tuple(h.shape[2:]))
        h = self.bn3.fused(self.conv3(h), ops.ACT_LRELU, 0.2)
        return self.bn4.fused(self.conv4(h), ops.ACT_LRELU, 0.2)

    def _object_canvas(self, image, label, transf_matrices, transf_matrices_inv, K, size):
        """per object: STN crop + tiled label -> self.local -> STN paste; summed over the objects"""
        B, ndf = image.shape[0], self.df_dim
        if BATCH_OBJECTS:
            # every object's crop reads the one image batch; the label rides into the concat as a code repeated over the plane,
            # straight from the loader's (B, K, label_dim) layout
            crop = _stn_objects(image, transf_matrices, K, (size, size), self._align())
            h = self.local(ops.cat_channels([(crop, "full"), (_obj(label, K), ("obj_plane", K))], K * B, (size, size)), groups=K)
            h = _stn_objects(h, transf_matrices_inv, K, (size, size), self._align())
            return ops.group_sum(h, K)
        canvas = None
        for idx in range(K):
            crop = ops.stn(image, transf_matrices[:, idx], (B, image.shape[1], size, size), self._align())
            h = self.local(torch.cat((crop, _tile(label[:, idx], size)), 1))
            h = ops.stn(h, transf_matrices_inv[:, idx], (B, ndf * 2, size, size), self._align())
            canvas = h if canvas is None else ops.add(canvas, h)
        return canvas

    def forward(self, image, label, transf_matrices, transf_matrices_inv, max_objects=None):
        return self._encode_img(image, label, transf_matrices, transf_matrices_inv,
                                max_objects or self.variant.max_objects)


# ------------------------------------------------------------------------------------- stage II (coco)
class STAGE2_G(nn.Module):
    """256x256 refinement generator (S/model.py:311-442): frozen STAGE1_G -> encoder to 16x16 -> joint with
    the caption code and the painted label layout -> residual blocks -> second object pathway (STN crop of
    the 16x16 features + label -> two upBlocks -> STN paste at 64x64) -> concat at 64x64 -> 256x256."""

    def __init__(self, cfg, variant, stage1_g):
        super(STAGE2_G, self).__init__()
        self.cfg, self.variant = cfg, variant
        self.gf_dim = cfg.GAN.GF_DIM
        self.ef_dim = cfg.GAN.CONDITION_DIM
        self.z_dim = cfg.Z_DIM
        self.STAGE1_G = stage1_g
        for p in self.STAGE1_G.parameters():
            p.requires_grad = False
        self.define_module()

    def _align(self):
        return bool(self.cfg.STN_ALIGN_CORNERS)

    def define_module(self):
        ngf, ef, v = self.gf_dim, self.ef_dim, self.variant
        self.ca_net = CA_NET(self.cfg)
        self.label = FusedSeq(HipLinear(ef + v.label_dim, ef, bias=False), HipBatchNorm1d(ef), nn.ReLU(True))
        # S/model.py:340 hard-codes 768 feature channels (= 4*ngf at GF_DIM 192) for the cropped patch
        self.local1 = upBlock(ef + 768, ngf * 2)
        self.local2 = upBlock(ngf * 2, ngf)
        self.encoder = FusedSeq(conv3x3(v.img_ch, ngf), nn.ReLU(True),
                                conv4x4(ngf, ngf * 2, 2), HipBatchNorm2d(ngf * 2), nn.ReLU(True),
                                conv4x4(ngf * 2, ngf * 4, 2), HipBatchNorm2d(ngf * 4), nn.ReLU(True))
        joint_in = (ef * 2 if self.cfg.USE_BBOX_LAYOUT else ef) + ngf * 4
        self.hr_joint = FusedSeq(conv3x3(joint_in, ngf * 4), HipBatchNorm2d(ngf * 4), nn.ReLU(True))
        self.residual = nn.Sequential(*[ResBlock(ngf * 4) for _ in range(self.cfg.GAN.R_NUM)])
        self.upsample1 = upBlock(ngf * 4, ngf * 2)
        self.upsample2 = upBlock(ngf * 2, ngf)
        self.upsample3 = upBlock(ngf * 2, ngf // 2)
        self.upsample4 = upBlock(ngf // 2, ngf // 4)
        self.img = FusedSeq(conv3x3(ngf // 4, v.img_ch), nn.Tanh())

    def forward(self, text_embedding, noise, transf_matrices_inv, transf_matrices_s2, transf_matrices_inv_s2,
                label_one_hot, max_objects=3, eps=None, eps_s1=None):
        B, ef = noise.shape[0], self.ef_dim
        with torch.no_grad():      # the reference detaches stage-I's output and froze its parameters
            _, stage1_img, _, _, _ = self.STAGE1_G(text_embedding, noise, transf_matrices_inv, label_one_hot,
                                                   eps=eps_s1)
        stage1_img = stage1_img.detach()
        encoded_img = self.encoder(stage1_img)
        c_code, mu, logvar = self.ca_net(text_embedding, eps)
        if BATCH_OBJECTS:
            return self._forward_batched(stage1_img, encoded_img, c_code, mu, logvar, transf_matrices_inv, transf_matrices_s2,
                                         transf_matrices_inv_s2, label_one_hot, max_objects)
        labels = [self.label(torch.cat((c_code, label_one_hot[:, idx]), 1)) for idx in range(max_objects)] \
            if self.cfg.USE_BBOX_LAYOUT else None
        parts = [encoded_img, _tile(c_code, 16)]
        if self.cfg.USE_BBOX_LAYOUT:
            layout = None
            for idx in range(max_objects):
                lab = ops.stn(_tile(labels[idx], 16), transf_matrices_inv[:, idx], (B, ef, 16, 16), self._align())
                layout = lab if layout is None else ops.add(layout, lab)
            parts.append(layout)
        h_code = self.residual(self.hr_joint(torch.cat(parts, 1)))
        canvas = None
        if labels is None:
            labels = []
        for idx in range(max_objects):
            if not self.cfg.USE_BBOX_LAYOUT:
                labels.append(self.label(torch.cat((c_code, label_one_hot[:, idx]), 1)))
            patch = ops.stn(h_code, transf_matrices_s2[:, idx], (B, h_code.shape[1], 16, 16), self._align())
            h = self.local2(self.local1(torch.cat((patch, _tile(labels[idx], 16)), 1)))
            h = ops.stn(h, transf_matrices_inv_s2[:, idx], (B, self.gf_dim, 64, 64), self._align())
            canvas = h if canvas is None else ops.add(canvas, h)
        h_code = self.upsample2(self.upsample1(h_code))
        h_code = torch.cat((h_code, canvas), 1)
        h_code = self.upsample4(self.upsample3(h_code))
        return stage1_img, self.img(h_code), mu, logvar, torch.stack(labels, 1)

    def _forward_batched(self, stage1_img, encoded_img, c_code, mu, logvar, transf_matrices_inv, transf_matrices_s2,
                         transf_matrices_inv_s2, label_one_hot, K):
        """the two object loops of S/model.py:380-420 as batches of K*B samples (see BATCH_OBJECTS)"""
        B, ef = c_code.shape[0], self.ef_dim
        lab = self.label(ops.cat_channels([(c_code, ("rep", K)), (_obj(label_one_hot, K), ("obj", K))], K * B), groups=K)   # (K*B, ef)
        parts = [(encoded_img, "full"), (c_code, "plane")]
        if self.cfg.USE_BBOX_LAYOUT:
            lay = _stn_objects(lab, transf_matrices_inv, K, (16, 16), self._align(), plane=True)
            parts.append((ops.group_sum(lay, K), "full"))
        h_code = self.residual(self.hr_joint(ops.cat_channels(parts, B, (16, 16))))
        patch = _stn_objects(h_code, transf_matrices_s2, K, (16, 16), self._align())
        h = self.local2(self.local1(ops.cat_channels([(patch, "full"), (lab, "plane")], K * B, (16, 16)), groups=K), groups=K)
        h = _stn_objects(h, transf_matrices_inv_s2, K, (64, 64), self._align())
        canvas = ops.group_sum(h, K)
        h_code = self.upsample2(self.upsample1(h_code))
        h_code = ops.cat_channels([(h_code, "full"), (canvas, "full")], B, tuple(h_code.shape[2:]))
        h_code = self.upsample4(self.upsample3(h_code))
        return stage1_img, self.img(h_code), mu, logvar, lab.view(K, B, ef).transpose(0, 1)


class STAGE2_D(nn.Module):
    """256x256 discriminator (S/model.py:445-537): object pathway at 32x32 with two conv4x4 s1 layers
    (32 -> 31 -> 30, pasted back to 32x32), concat after conv3, six stride-2 convs + two 3x3."""

    def __init__(self, cfg, variant):
        super(STAGE2_D, self).__init__()
        self.cfg, self.variant = cfg, variant
        self.df_dim = cfg.GAN.DF_DIM
        self.ef_dim = cfg.GAN.CONDITION_DIM
        ndf, nef, v = self.df_dim, self.ef_dim, variant
        self.local = FusedSeq(conv4x4(v.img_ch + v.label_dim, ndf * 2, 1), HipBatchNorm2d(ndf * 2),
                              nn.LeakyReLU(0.2, inplace=True),
                              conv4x4(ndf * 2, ndf * 2, 1), HipBatchNorm2d(ndf * 2),
                              nn.LeakyReLU(0.2, inplace=True))
        self.act = nn.LeakyReLU(0.2, inplace=True)
        self.conv1 = conv4x4(v.img_ch, ndf, 2)
        widths = [(ndf, ndf * 2), (ndf * 2, ndf * 4), (ndf * 6, ndf * 8), (ndf * 8, ndf * 16), (ndf * 16, ndf * 32)]
        for i, (cin, cout) in enumerate(widths):
            setattr(self, "conv%d" % (i + 2), conv4x4(cin, cout, 2))
            setattr(self, "bn%d" % (i + 2), HipBatchNorm2d(cout))
        self.conv7 = conv3x3(ndf * 32, ndf * 16)
        self.bn7 = HipBatchNorm2d(ndf * 16)
        self.conv8 = conv3x3(ndf * 16, ndf * 8)
        self.bn8 = HipBatchNorm2d(ndf * 8)
        self.get_cond_logits = D_GET_LOGITS(ndf, nef, bcondition=True)
        self.get_uncond_logits = D_GET_LOGITS(ndf, nef, bcondition=False)

    def _align(self):
        return bool(self.cfg.STN_ALIGN_CORNERS)

    def _encode_img(self, image, label, transf_matrices, transf_matrices_inv, max_objects):
        B, ndf = image.shape[0], self.df_dim
        canvas = STAGE1_D._object_canvas(self, image, label, transf_matrices, transf_matrices_inv, max_objects, 32)
        h = ops.conv2d_lrelu(image, self.conv1.weight, self.conv1.stride[0], self.conv1.padding, 0.2)
        h = self.bn2.fused(self.conv2(h), ops.ACT_LRELU, 0.2)
        h = self.bn3.fused(self.conv3(h), ops.ACT_LRELU, 0.2)
        h = ops.cat_channels([(h, "full"), (canvas, "full")], B, tuple(h.shape[2:]))
        for i in range(4, 9):
            h = getattr(self, "bn%d" % i).fused(getattr(self, "conv%d" % i)(h), ops.ACT_LRELU, 0.2)
        return h

    def forward(self, image, label, transf_matrices, transf_matrices_inv, max_objects=3):
        return self._encode_img(image, label, transf_matrices, transf_matrices_inv, max_objects)
