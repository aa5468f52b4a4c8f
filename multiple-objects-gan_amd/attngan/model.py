"""Host-side mirror of the reference's coco-attngan networks (code/coco/attngan/model.py) on the
MI355X kernels of libmogan_hip.so.

Same class surface, constructor arguments, forward signatures, return structure and -- because
attribute names and nn.Sequential indices are kept -- the same state_dict keys, so reference
checkpoints load and `netG.apply(weights_init)` works (parameter holders are subclasses of
nn.Conv2d / nn.BatchNorm* / nn.Linear whose class names still contain 'Conv' / 'BatchNorm' /
'Linear', which miscc/utils.py:321-331 dispatches on).  What differs is underneath: every
nn.Sequential is a `FusedSeq` that walks its children and issues fused HIP launches
(upsample+conv, BN+GLU/LeakyReLU/ReLU, BN+residual), and there is no CPU path.
"""
import os

import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from ..hip import ops
from .miscc.config import cfg
from .model_base import FusedSeq, GLU, HipBatchNorm1d, HipBatchNorm2d, HipConv2d, HipLinear
from .GlobalAttention import GlobalAttentionGeneral as ATT_NET
from . import inception

MAX_OBJECTS = 3
# The reference's python loops over the objects as ONE batch of MAX_OBJECTS*B samples (object-major) with per-object BatchNorm
# statistics (FusedSeq(..., groups=G)); 0 = the literal loops (same results; A/B switch)
BATCH_OBJECTS = True     # (module attribute: False = the reference's literal per-object loops; tests compare both)


def stn(image, transformation_matrix, size):
    """model.py:17-21; align_corners is explicit (cfg.STN_ALIGN_CORNERS)."""
    return ops.stn(image, transformation_matrix, size, bool(cfg.STN_ALIGN_CORNERS))


def _objects_first(t, G):
    """(B, >= G, ...) per-object tensor -> (G, B, ...): the objects of the reference's loops as the slow batch index"""
    return t[:, :G].transpose(0, 1)


def _stn_objects(x, theta, G, out_hw, plane=False):
    """stn(x_g, theta[:, g], ...) for every object g as one object-major batch of G*B' samples (model.py:109-111, 402-404, 663-671):
    x is (G*B', C, H, W) -- one map per (object, image) --, or (B', C, H, W) -- the same image for every object --, or, plane,
    (G*B', C) -- a label vector the reference repeats over the plane first.  theta (B', G, 2, 3) stays in the loader's layout:
    the kernel does the object-major lookup, and a shared / constant source is never materialised."""
    Bp = theta.shape[0]
    N = G * Bp
    if theta.shape[1] != G:
        theta = theta[:, :G].contiguous()
    in_hw = out_hw if plane else tuple(x.shape[2:])
    return ops.stn_shared(x, theta, N, in_hw, out_hw, bool(cfg.STN_ALIGN_CORNERS), plane=plane, theta_G=G)


def _sum_objects(h, G):
    """(G*B, ...) object-major -> (B, ...): h_0 + h_1 + ... in the loop's order (model.py:113,406,671)"""
    return ops.group_sum(h, G)


def conv1x1(in_planes, out_planes, bias=False):
    return HipConv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=bias)


def conv3x3(in_planes, out_planes, stride=1):
    return HipConv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def upBlock(in_planes, out_planes):
    """nearest x2 -> conv3x3 -> BN -> GLU (model.py:48-55) = one conv launch (upsample fused into the
    im2col gather) + BN statistics + one BN+GLU apply."""
    return FusedSeq(nn.Upsample(scale_factor=2, mode='nearest'),
                    conv3x3(in_planes, out_planes * 2),
                    HipBatchNorm2d(out_planes * 2),
                    GLU())


def Block3x3_relu(in_planes, out_planes):
    return FusedSeq(conv3x3(in_planes, out_planes * 2), HipBatchNorm2d(out_planes * 2), GLU())


class ResBlock(nn.Module):
    """model.py:67-81; the `out += residual` is fused into the second BN apply."""

    def __init__(self, channel_num):
        super(ResBlock, self).__init__()
        self.block = FusedSeq(conv3x3(channel_num, channel_num * 2), HipBatchNorm2d(channel_num * 2), GLU(),
                              conv3x3(channel_num, channel_num), HipBatchNorm2d(channel_num))

    def forward(self, x):
        return self.block(x, residual=x)


class BBOX_NET(nn.Module):
    """model.py:84-116."""

    def __init__(self):
        super(BBOX_NET, self).__init__()
        self.c_dim = cfg.GAN.CONDITION_DIM
        c = self.c_dim
        self.encode = FusedSeq(
            conv3x3(c, c // 2, stride=2), nn.LeakyReLU(0.2, inplace=True),
            conv3x3(c // 2, c // 4, stride=2), HipBatchNorm2d(c // 4), nn.LeakyReLU(0.2, inplace=True),
            conv3x3(c // 4, c // 8, stride=2), HipBatchNorm2d(c // 8), nn.LeakyReLU(0.2, inplace=True))

    def forward(self, labels, transf_matr_inv):
        B, G = labels.shape[0], MAX_OBJECTS
        if not BATCH_OBJECTS:
            label_layout = None
            for idx in range(G):
                lab = labels[:, idx].reshape(B, self.c_dim, 1, 1).expand(B, self.c_dim, 16, 16)
                lab = stn(lab, transf_matr_inv[:, idx], (B, self.c_dim, 16, 16))
                label_layout = lab if label_layout is None else ops.add(label_layout, lab)
            return self.encode(label_layout).view(B, -1)
        # the reference's loop over the objects (model.py:105-114) as ONE batch of G*B samples, object-major; the label vector is
        # the transformer's constant source (no 16 x 16 copy of it)
        lab = _objects_first(labels, G).reshape(G * B, self.c_dim)
        lab = _stn_objects(lab, transf_matr_inv, G, (16, 16), plane=True)
        label_layout = _sum_objects(lab, G)
        return self.encode(label_layout).view(B, -1)


# --------------------------------------------------------------------------- text / image encoders
class RNN_ENCODER(nn.Module):
    """model.py:120-204.  Frozen front-end that runs once per step without gradients.  The parameters live in the stock
    nn.Embedding / nn.LSTM modules (state_dict keys of the reference); in eval mode on the device the forward is ONE launch
    (csrc/mogan_lstm.hip: 119 launches and 0.8 ms of host time per call on MIOpen's LSTM), anything else -- training mode,
    gradients, a GRU, other sizes -- takes the stock modules."""

    def __init__(self, ntoken, ninput=300, drop_prob=0.5, nhidden=128, nlayers=1, bidirectional=True):
        super(RNN_ENCODER, self).__init__()
        self.n_steps = cfg.TEXT.WORDS_NUM
        self.ntoken, self.ninput, self.drop_prob = ntoken, ninput, drop_prob
        self.nlayers, self.bidirectional, self.rnn_type = nlayers, bidirectional, cfg.RNN_TYPE
        self.num_directions = 2 if bidirectional else 1
        self.nhidden = nhidden // self.num_directions
        self.encoder = nn.Embedding(self.ntoken, self.ninput)
        self.drop = nn.Dropout(self.drop_prob)
        rnn = {'LSTM': nn.LSTM, 'GRU': nn.GRU}.get(self.rnn_type)
        if rnn is None:
            raise NotImplementedError
        self.rnn = rnn(self.ninput, self.nhidden, self.nlayers, batch_first=True, dropout=self.drop_prob,
                       bidirectional=self.bidirectional)
        self.encoder.weight.data.uniform_(-0.1, 0.1)

    def init_hidden(self, bsz):
        weight = next(self.parameters()).data
        shape = (self.nlayers * self.num_directions, bsz, self.nhidden)
        if self.rnn_type == 'LSTM':
            return (weight.new_zeros(shape), weight.new_zeros(shape))
        return weight.new_zeros(shape)

    FUSED = True      # eval mode on the device: embedding + bi-LSTM as one launch (hip/ops.lstm_encoder_forward)

    def forward(self, captions, cap_lens, hidden, mask=None):
        lens = cap_lens.data.tolist() if torch.is_tensor(cap_lens) else list(cap_lens)
        if (self.FUSED and captions.is_cuda and not self.training and not torch.is_grad_enabled() and self.rnn_type == 'LSTM'
                and self.nlayers == 1 and self.bidirectional):
            from ..hip import ops
            out = ops.lstm_encoder_forward(captions, lens, self.encoder.weight, self.rnn, hidden[0], hidden[1])
            if out is not None:                     # (B, 2H, T_max) = the reference's output.transpose(1, 2), and (B, 2H)
                return out
        emb = self.drop(self.encoder(captions))
        emb = pack_padded_sequence(emb, lens, batch_first=True)
        output, hidden = self.rnn(emb, hidden)
        output = pad_packed_sequence(output, batch_first=True)[0]
        words_emb = output.transpose(1, 2)
        sent_emb = (hidden[0] if self.rnn_type == 'LSTM' else hidden).transpose(0, 1).contiguous()
        return words_emb, sent_emb.view(-1, self.nhidden * self.num_directions)


class CNN_ENCODER(nn.Module):
    """model.py:207-313: bilinear resize to 299x299, Inception-v3 trunk through Mixed_7c, 1x1 projection
    of the 17x17x768 map (`emb_features`) and a linear projection of the pooled 2048 code
    (`emb_cnn_code`).  The reference downloads ImageNet weights in __init__ (model.py:215-217); here
    `pretrained=False` random-initialises the trunk (no network on the GPU box) and checkpoints are
    loaded through load_state_dict with the same (torchvision) key names."""

    def __init__(self, nef, pretrained=False):
        super(CNN_ENCODER, self).__init__()
        self.nef = nef if cfg.TRAIN.FLAG else 256
        if pretrained:
            raise RuntimeError("no network access: load the DAMSM image_encoder checkpoint instead")
        for name, make in inception.TRUNK:
            setattr(self, name, make())
        inception.init_trunk(self)
        self.emb_features = HipConv2d(768, self.nef, kernel_size=1, stride=1, padding=0, bias=False)
        self.emb_cnn_code = HipLinear(2048, self.nef)
        self.init_trainable_weights()

    def init_trainable_weights(self):
        self.emb_features.weight.data.uniform_(-0.1, 0.1)
        self.emb_cnn_code.weight.data.uniform_(-0.1, 0.1)

    def _frozen(self):
        return inception.FAST_TRUNK and not self.training and not any(p.requires_grad for p in self.parameters())

    def forward(self, x):
        x = ops.bilinear_resize(x, 299, 299)
        if self._frozen():
            # the train step's case (trainer.py:62-66): explicit forward/backward over the folded, grouped trunk
            features, x = inception.frozen_trunk(self, x)
            x = ops.avg_pool2d(x, 8).view(x.size(0), -1)
            return self.emb_features(features), self.emb_cnn_code(x)
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x)))
        x = ops.max_pool2d(x, 3, 2)
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x))
        x = ops.max_pool2d(x, 3, 2)
        for name in ("Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
            x = getattr(self, name)(x)
        features = x                                                   # 17 x 17 x 768
        for name in ("Mixed_7a", "Mixed_7b", "Mixed_7c"):
            x = getattr(self, name)(x)
        x = ops.avg_pool2d(x, 8).view(x.size(0), -1)                   # 2048
        return self.emb_features(features), self.emb_cnn_code(x)


# --------------------------------------------------------------------------- generator
class CA_NET(nn.Module):
    """model.py:317-345.  eps ~ N(0,1) is drawn inside forward in the reference; `eps` may be
    injected for parity tests / graph capture."""

    def __init__(self):
        super(CA_NET, self).__init__()
        self.t_dim = cfg.TEXT.EMBEDDING_DIM
        self.c_dim = cfg.GAN.CONDITION_DIM
        self.fc = HipLinear(self.t_dim, self.c_dim * 4, bias=True)
        self.relu = GLU()

    def encode(self, text_embedding):
        x = self.relu(self.fc(text_embedding))
        return x[:, :self.c_dim], x[:, self.c_dim:]

    def reparametrize(self, mu, logvar, eps=None):
        if eps is None:
            eps = torch.randn_like(mu)
        return ops.reparam(mu, logvar, eps)

    def forward(self, text_embedding, eps=None):
        mu, logvar = self.encode(text_embedding)
        return self.reparametrize(mu, logvar, eps), mu, logvar


class INIT_STAGE_G(nn.Module):
    """model.py:348-422: object pathway (x3, one BN call per object: SURVEY.md F11), BBOX_NET,
    global pathway, concat at 16x16."""

    def __init__(self, ngf, ncf):
        super(INIT_STAGE_G, self).__init__()
        self.gf_dim = ngf
        self.in_dim = cfg.GAN.Z_DIM + ncf
        self.define_module()

    def define_module(self):
        nz, ngf = self.in_dim, self.gf_dim
        linput = 100 + 81
        self.ef_dim = 100
        self.bbox_net = BBOX_NET()
        nz += 48
        self.fc = FusedSeq(HipLinear(nz, ngf * 4 * 4 * 2, bias=False), HipBatchNorm1d(ngf * 4 * 4 * 2), GLU())
        self.label = FusedSeq(HipLinear(linput, self.ef_dim, bias=False), HipBatchNorm1d(self.ef_dim),
                              nn.ReLU(True))
        self.local1 = upBlock(self.ef_dim, ngf // 2)
        self.local2 = upBlock(ngf // 2, ngf // 4)
        self.upsample1 = upBlock(ngf, ngf // 2)
        self.upsample2 = upBlock(ngf // 2, ngf // 4)
        self.upsample3 = upBlock(ngf // 2, ngf // 8)
        self.upsample4 = upBlock(ngf // 8, ngf // 16)

    def forward(self, z_code, c_code, transf_matrices_inv, label_one_hot):
        B, G = z_code.shape[0], MAX_OBJECTS
        if not BATCH_OBJECTS:
            return self._forward_looped(z_code, c_code, transf_matrices_inv, label_one_hot)
        # the object loop of model.py:395-407 as ONE batch of G*B samples (object-major); every BatchNorm inside still sees one
        # object's B samples per "call" (groups=G: own statistics, running statistics updated object after object)
        # (every concat / repeat of the loop body is one ops.cat_channels launch: c_code repeated for the G objects next to
        # label_one_hot[:, g], the label code repeated over the 4 x 4 plane, ...)
        onehot = label_one_hot if label_one_hot.shape[1] == G else label_one_hot[:, :G]
        lab = self.label(ops.cat_channels([(c_code, ("rep", G)), (onehot, ("obj", G))], G * B), groups=G)
        h = ops.cat_channels([(lab, "plane")], G * B, (4, 4))
        h = self.local2(self.local1(h, groups=G), groups=G)
        h = _stn_objects(h, transf_matrices_inv, G, tuple(h.shape[2:]))
        h_code_locals = _sum_objects(h, G)
        bbox_code = self.bbox_net(lab.view(G, B, self.ef_dim).transpose(0, 1), transf_matrices_inv)
        out_code = self.fc(ops.cat_channels([(c_code, "full"), (z_code, "full"), (bbox_code, "full")], B))
        out_code = self.upsample2(self.upsample1(out_code.view(-1, self.gf_dim, 4, 4)))
        out_code = ops.cat_channels([(out_code, "full"), (h_code_locals, "full")], B, tuple(out_code.shape[2:]))
        return self.upsample4(self.upsample3(out_code))

    def _forward_looped(self, z_code, c_code, transf_matrices_inv, label_one_hot):
        """the literal loop of model.py:395-407 (MOGAN_OBJ_BATCH=0)"""
        B = z_code.shape[0]
        local_labels, h_code_locals = [], None
        for idx in range(MAX_OBJECTS):
            lab = self.label(torch.cat((c_code, label_one_hot[:, idx]), 1))
            local_labels.append(lab)
            h = lab.view(B, self.ef_dim, 1, 1).expand(B, self.ef_dim, 4, 4)
            h = self.local2(self.local1(h))
            h = stn(h, transf_matrices_inv[:, idx], h.shape)
            h_code_locals = h if h_code_locals is None else ops.add(h_code_locals, h)
        bbox_code = self.bbox_net(torch.stack(local_labels, 1), transf_matrices_inv)
        out_code = self.fc(torch.cat((c_code, z_code, bbox_code), 1)).view(-1, self.gf_dim, 4, 4)
        out_code = self.upsample2(self.upsample1(out_code))
        out_code = torch.cat((out_code, h_code_locals), 1)
        return self.upsample4(self.upsample3(out_code))


class NEXT_STAGE_G(nn.Module):
    """model.py:425-461."""

    def __init__(self, ngf, nef, ncf):
        super(NEXT_STAGE_G, self).__init__()
        self.gf_dim, self.ef_dim, self.cf_dim = ngf, nef, ncf
        self.num_residual = cfg.GAN.R_NUM
        self.att = ATT_NET(ngf, self.ef_dim)
        self.residual = nn.Sequential(*[ResBlock(ngf * 2) for _ in range(cfg.GAN.R_NUM)])
        self.upsample = upBlock(ngf * 2, ngf)

    def forward(self, h_code, c_code, word_embs, mask):
        self.att.applyMask(mask)
        c_code, att = self.att(h_code, word_embs)
        out_code = self.residual(ops.cat_channels([(h_code, "full"), (c_code, "full")], h_code.shape[0],
                                                  tuple(h_code.shape[2:])))
        return self.upsample(out_code), att


class GET_IMAGE_G(nn.Module):
    """model.py:464-475."""

    def __init__(self, ngf):
        super(GET_IMAGE_G, self).__init__()
        self.gf_dim = ngf
        self.img = FusedSeq(conv3x3(ngf, 3), nn.Tanh())

    def forward(self, h_code):
        return self.img(h_code)


class G_NET(nn.Module):
    """model.py:478-528."""

    def __init__(self):
        super(G_NET, self).__init__()
        ngf, nef, ncf = cfg.GAN.GF_DIM, cfg.TEXT.EMBEDDING_DIM, cfg.GAN.CONDITION_DIM
        self.ca_net = CA_NET()
        if cfg.TREE.BRANCH_NUM > 0:
            self.h_net1 = INIT_STAGE_G(ngf * 16, ncf)
            self.img_net1 = GET_IMAGE_G(ngf)
        if cfg.TREE.BRANCH_NUM > 1:
            self.h_net2 = NEXT_STAGE_G(ngf, nef, ncf)
            self.img_net2 = GET_IMAGE_G(ngf)
        if cfg.TREE.BRANCH_NUM > 2:
            self.h_net3 = NEXT_STAGE_G(ngf, nef, ncf)
            self.img_net3 = GET_IMAGE_G(ngf)

    def forward(self, z_code, sent_emb, word_embs, mask, transf_matrices_inv, label_one_hot, eps=None):
        fake_imgs, att_maps = [], []
        c_code, mu, logvar = self.ca_net(sent_emb, eps)
        if cfg.TREE.BRANCH_NUM > 0:
            h_code = self.h_net1(z_code, c_code, transf_matrices_inv, label_one_hot)
            fake_imgs.append(self.img_net1(h_code))
        if cfg.TREE.BRANCH_NUM > 1:
            h_code, att1 = self.h_net2(h_code, c_code, word_embs, mask)
            fake_imgs.append(self.img_net2(h_code))
            if att1 is not None:
                att_maps.append(att1)
        if cfg.TREE.BRANCH_NUM > 2:
            h_code, att2 = self.h_net3(h_code, c_code, word_embs, mask)
            fake_imgs.append(self.img_net3(h_code))
            if att2 is not None:
                att_maps.append(att2)
        return fake_imgs, att_maps, mu, logvar


# --------------------------------------------------------------------------- discriminators
def Block3x3_leakRelu(in_planes, out_planes):
    return FusedSeq(conv3x3(in_planes, out_planes), HipBatchNorm2d(out_planes), nn.LeakyReLU(0.2, inplace=True))


def downBlock(in_planes, out_planes):
    return FusedSeq(HipConv2d(in_planes, out_planes, 4, 2, 1, bias=False), HipBatchNorm2d(out_planes),
                    nn.LeakyReLU(0.2, inplace=True))


def encode_image_by_16times(ndf):
    return FusedSeq(
        HipConv2d(3, ndf, 4, 2, 1, bias=False), nn.LeakyReLU(0.2, inplace=True),
        HipConv2d(ndf, ndf * 2, 4, 2, 1, bias=False), HipBatchNorm2d(ndf * 2), nn.LeakyReLU(0.2, inplace=True),
        HipConv2d(ndf * 2, ndf * 4, 4, 2, 1, bias=False), HipBatchNorm2d(ndf * 4), nn.LeakyReLU(0.2, inplace=True),
        HipConv2d(ndf * 4, ndf * 8, 4, 2, 1, bias=False), HipBatchNorm2d(ndf * 8), nn.LeakyReLU(0.2, inplace=True))


class D_GET_LOGITS(nn.Module):
    """model.py:616-642."""

    def __init__(self, ndf, nef, bcondition=False):
        super(D_GET_LOGITS, self).__init__()
        self.df_dim, self.ef_dim, self.bcondition = ndf, nef, bcondition
        if self.bcondition:
            self.jointConv = Block3x3_leakRelu(ndf * 8 + nef, ndf * 8)
        self.outlogits = FusedSeq(HipConv2d(ndf * 8, 1, kernel_size=4, stride=4), nn.Sigmoid())

    def forward(self, h_code, c_code=None):
        if self.bcondition and c_code is not None:
            # model.py:632-634: c_code repeated over the 4 x 4 map next to h_code, one launch
            h_code = self.jointConv(ops.cat_channels([(h_code, "full"), (c_code.reshape(-1, self.ef_dim), "plane")],
                                                     h_code.shape[0], tuple(h_code.shape[2:])))
        return self.outlogits(h_code).view(-1)


class _D_BASE(nn.Module):
    def _logit_heads(self, b_jcu):
        ndf, nef = cfg.GAN.DF_DIM, cfg.TEXT.EMBEDDING_DIM
        self.UNCOND_DNET = D_GET_LOGITS(ndf, nef, bcondition=False) if b_jcu else None
        self.COND_DNET = D_GET_LOGITS(ndf, nef, bcondition=True)


class D_NET64(_D_BASE):
    """model.py:646-711: object pathway (stn crop -> cat one-hot -> conv4x4 s1 -> BN -> LeakyReLU -> stn
    paste, x3 with per-call BN statistics) + global pathway."""

    def __init__(self, b_jcu=True):
        super(D_NET64, self).__init__()
        self._logit_heads(b_jcu)
        ndf = cfg.GAN.DF_DIM
        self.act = nn.LeakyReLU(0.2, inplace=True)
        self.conv1 = HipConv2d(3, ndf, 4, 2, 1, bias=False)
        self.conv2 = HipConv2d(ndf, ndf * 2, 4, 2, 1, bias=False)
        self.bn2 = HipBatchNorm2d(ndf * 2)
        self.conv3 = HipConv2d(ndf * 4, ndf * 4, 4, 2, 1, bias=False)
        self.bn3 = HipBatchNorm2d(ndf * 4)
        self.conv4 = HipConv2d(ndf * 4, ndf * 8, 4, 2, 1, bias=False)
        self.bn4 = HipBatchNorm2d(ndf * 8)
        self.local = FusedSeq(HipConv2d(3 + 81, ndf * 2, 4, 1, 1, bias=False), HipBatchNorm2d(ndf * 2),
                              nn.LeakyReLU(0.2, inplace=True))

    def forward(self, image, label, transf_matrices, transf_matrices_inv):
        B, G = image.shape[0], MAX_OBJECTS
        if not BATCH_OBJECTS:
            h_code_locals = None
            for idx in range(G):
                lab = label[:, idx].reshape(B, 81, 1, 1).expand(B, 81, 16, 16)
                h = stn(image, transf_matrices[:, idx], (B, image.shape[1], 16, 16))
                h = self.local(torch.cat((h, lab), 1))
                h = stn(h, transf_matrices_inv[:, idx], (B, h.shape[1], 16, 16))
                h_code_locals = h if h_code_locals is None else ops.add(h_code_locals, h)
            return self._trunk(image, h_code_locals)
        # the object loop of model.py:662-672 as ONE batch of G*B samples (object-major), BatchNorm per object (groups=G)
        # every object's crop reads the ONE image batch (no G-fold copy); the one-hot label rides into the concat as a code
        # repeated over the 16 x 16 plane, straight from the loader's (B, G, 81) layout
        onehot = label if label.shape[1] == G else label[:, :G]
        h = _stn_objects(image, transf_matrices, G, (16, 16))
        h = self.local(ops.cat_channels([(h, "full"), (onehot, ("obj_plane", G))], G * B, (16, 16)), groups=G)
        h = _stn_objects(h, transf_matrices_inv, G, (16, 16))
        return self._trunk(image, _sum_objects(h, G))

    def _trunk(self, image, h_code_locals):
        h = ops.conv2d_lrelu(image, self.conv1.weight, self.conv1.stride[0], self.conv1.padding, 0.2)
        h = self.bn2.fused(self.conv2(h), ops.ACT_LRELU, 0.2)
        h = ops.cat_channels([(h, "full"), (h_code_locals, "full")], h.shape[0], tuple(h.shape[2:]))
        h = self.bn3.fused(self.conv3(h), ops.ACT_LRELU, 0.2)
        return self.bn4.fused(self.conv4(h), ops.ACT_LRELU, 0.2)


class D_NET128(_D_BASE):
    """model.py:715-734."""

    def __init__(self, b_jcu=True):
        super(D_NET128, self).__init__()
        ndf = cfg.GAN.DF_DIM
        self.img_code_s16 = encode_image_by_16times(ndf)
        self.img_code_s32 = downBlock(ndf * 8, ndf * 16)
        self.img_code_s32_1 = Block3x3_leakRelu(ndf * 16, ndf * 8)
        self._logit_heads(b_jcu)

    PAIRED = True           # forward(x, groups=2): x = [real; fake], one pass with per-half BatchNorm statistics (losses.py)

    def forward(self, x_var, groups=1):
        x = self.img_code_s32(self.img_code_s16(x_var, groups=groups), groups=groups)
        return self.img_code_s32_1(x, groups=groups)


class D_NET256(_D_BASE):
    """model.py:738-760."""

    def __init__(self, b_jcu=True):
        super(D_NET256, self).__init__()
        ndf = cfg.GAN.DF_DIM
        self.img_code_s16 = encode_image_by_16times(ndf)
        self.img_code_s32 = downBlock(ndf * 8, ndf * 16)
        self.img_code_s64 = downBlock(ndf * 16, ndf * 32)
        self.img_code_s64_1 = Block3x3_leakRelu(ndf * 32, ndf * 16)
        self.img_code_s64_2 = Block3x3_leakRelu(ndf * 16, ndf * 8)
        self._logit_heads(b_jcu)

    PAIRED = True

    def forward(self, x_var, groups=1):
        x = self.img_code_s32(self.img_code_s16(x_var, groups=groups), groups=groups)
        x = self.img_code_s64(x, groups=groups)
        return self.img_code_s64_2(self.img_code_s64_1(x, groups=groups), groups=groups)
