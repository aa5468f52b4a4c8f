"""CPU oracle for the StackGAN-family train steps (coco-stackgan stage I/II, clevr, multi-mnist):
a functional torch-CPU (fp32 or fp64) restatement.  TEST INFRASTRUCTURE ONLY -- imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / reported CPU baseline;
never by the product package.

Pinned against tests/golden/stackgan_*.npz, which tests/golden/make_golden_stackgan.py captured from
the reference's own python run on torch-CPU in the build container:
    S = /root/reference/code/coco/stackgan   (model.py, miscc/utils.py, trainer.py:188-231)
    C = /root/reference/code/clevr           (model.py, miscc/utils.py, trainer.py:127-157)
    M = /root/reference/code/multi-mnist     (model.py, miscc/utils.py, trainer.py:131-160)
As for AttnGAN, the arithmetic of conv/BN/grid_sample lives in PyTorch (torch==0.4.1 pinned by the
reference, not vendored); `align_corners` and the Adam epsilon placement are explicit.

Every network is a plain dict {state_dict key -> tensor} with the reference's key names.
"""
import torch
import torch.nn.functional as F

from . import attngan_oracle as A
from .attngan_oracle import (adam_state, adam_step, bn, conv, from_state_dict, kl_loss, lrelu,  # noqa: F401
                             parameters, stn, zero_grad)


class SCfg:
    """The cfg fields the path reads + the constants the three model.py files hard-code."""

    def __init__(self, tree, gf_dim=None, df_dim=None, cond_dim=None, z_dim=100, text_dim=1024, r_num=2,
                 use_bbox_layout=True, kl_coeff=2.0, lr_g=2e-4, lr_d=2e-4, stage=1):
        assert tree in ("coco", "clevr", "mnist")
        dflt = {"coco": (192, 96, 128), "clevr": (96, 48, 16), "mnist": (128, 64, 128)}[tree]
        self.tree, self.stage = tree, stage
        self.gf_dim = dflt[0] if gf_dim is None else gf_dim
        self.df_dim = dflt[1] if df_dim is None else df_dim
        self.cond_dim = dflt[2] if cond_dim is None else cond_dim
        self.z_dim, self.text_dim, self.r_num = z_dim, text_dim, r_num
        self.use_bbox_layout, self.kl_coeff, self.lr_g, self.lr_d = use_bbox_layout, kl_coeff, lr_g, lr_d
        self.text = tree == "coco"
        self.img_ch = 1 if tree == "mnist" else 3                     # M/model.py:155,205,212
        self.label_dim = {"coco": 81, "clevr": 13, "mnist": 10}[tree]  # S:166,252 C:123,204 M:117
        self.max_objects = 4 if tree == "clevr" else 3
        self.ef_dim = 10 if tree == "mnist" else self.cond_dim         # M/model.py:117
        self.bbox_in = self.ef_dim                                     # M/model.py:86: conv3x3(10, ...)
        self.bbox_c = 128 if tree == "mnist" else self.cond_dim        # M/model.py:83
        self.d_cond = {"coco": self.cond_dim, "clevr": 13, "mnist": 10}[tree]


def relu_up(net, pre, x):
    """upBlock (S/model.py:16-22): nearest x2 -> conv3x3 (.1) -> BN (.2) -> ReLU."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.relu(bn(net, pre + ".2", conv(net, pre + ".1", x)))


def res_block(net, pre, x):
    """ResBlock (S/model.py:25-41)."""
    h = F.relu(bn(net, pre + ".block.1", conv(net, pre + ".block.0", x)))
    h = bn(net, pre + ".block.4", conv(net, pre + ".block.3", h))
    return F.relu(h + x)


def tile(v, size):
    return v.reshape(v.shape[0], -1, 1, 1).repeat(1, 1, size, size)


def ca_net(net, pre, text, eps, cfg):
    """CA_NET (S/model.py:44-72): ReLU, eps injected."""
    x = F.relu(F.linear(text, net[pre + ".fc.weight"], net[pre + ".fc.bias"]))
    mu, logvar = x[:, :cfg.cond_dim], x[:, cfg.cond_dim:]
    return eps * torch.exp(0.5 * logvar) + mu, mu, logvar


def bbox_net(net, pre, labels, tmi, cfg, K):
    """BBOX_NET (S/model.py:114-142, C:80-111, M:80-110)."""
    B = labels.shape[0]
    layout = labels.new_zeros(B, cfg.bbox_in, 16, 16)
    for k in range(K):
        lab = tile(labels[:, k], 16)
        layout = layout + stn(lab, tmi[:, k], lab.shape)
    h = lrelu(conv(net, pre + ".encode.0", layout, 2, 1))
    h = lrelu(bn(net, pre + ".encode.3", conv(net, pre + ".encode.2", h, 2, 1)))
    h = lrelu(bn(net, pre + ".encode.6", conv(net, pre + ".encode.5", h, 2, 1)))
    return h.reshape(B, -1)


def stage1_g(net, cfg, z, tmi, onehot, text=None, eps=None, pre="", K=None):
    """STAGE1_G.forward (S/model.py:201-235, C:158-192, M:158-190) -> (img, mu, logvar, local_labels)."""
    K = K or cfg.max_objects
    B, ngf = z.shape[0], cfg.gf_dim * 8
    c = mu = logvar = None
    if cfg.text:
        c, mu, logvar = ca_net(net, pre + "ca_net", text, eps, cfg)
    labels = []
    canvas = z.new_zeros(B, ngf // 4, 16, 16)
    for k in range(K):
        if cfg.tree == "mnist":          # M/model.py:163: the one-hot itself; `self.label` is never applied
            lab = onehot[:, k]
        else:
            src = onehot[:, k] if c is None else torch.cat((c, onehot[:, k]), 1)
            lab = F.relu(bn(net, pre + "label.1", F.linear(src, net[pre + "label.0.weight"])))
        labels.append(lab)
        h = relu_up(net, pre + "local1", tile(lab, 4))
        h = relu_up(net, pre + "local2", h)
        canvas = canvas + stn(h, tmi[:, k], h.shape)
    local_labels = torch.stack(labels, 1)
    parts = [z] + ([c] if c is not None else [])
    if cfg.use_bbox_layout:
        parts.append(bbox_net(net, pre + "bbox_net", local_labels, tmi, cfg, K))
    h = F.linear(torch.cat(parts, 1), net[pre + "fc.0.weight"])
    h = F.relu(bn(net, pre + "fc.1", h)).reshape(B, ngf, 4, 4)
    h = relu_up(net, pre + "upsample1", h)
    h = relu_up(net, pre + "upsample2", h)
    h = torch.cat((h, canvas), 1)
    h = relu_up(net, pre + "upsample3", h)
    h = relu_up(net, pre + "upsample4", h)
    return torch.tanh(conv(net, pre + "img.0", h)), mu, logvar, local_labels


def stage1_d(net, cfg, image, label, tm, tmi, K=None):
    """STAGE1_D._encode_img (S/model.py:266-302, C:226-255, M:223-252)."""
    K = K or cfg.max_objects
    B, ndf = image.shape[0], cfg.df_dim
    canvas = image.new_zeros(B, ndf * 2, 16, 16)
    for k in range(K):
        h = stn(image, tm[:, k], (B, image.shape[1], 16, 16))
        h = torch.cat((h, tile(label[:, k], 16)), 1)
        h = lrelu(bn(net, "local.1", conv(net, "local.0", h, 1, 1)))
        canvas = canvas + stn(h, tmi[:, k], (B, ndf * 2, 16, 16))
    h = lrelu(conv(net, "conv1", image, 2, 1))
    h = lrelu(bn(net, "bn2", conv(net, "conv2", h, 2, 1)))
    h = torch.cat((h, canvas), 1)
    h = lrelu(bn(net, "bn3", conv(net, "conv3", h, 2, 1)))
    return lrelu(bn(net, "bn4", conv(net, "conv4", h, 2, 1)))


def stage2_g(net, cfg, text, z, tmi, tm_s2, tmi_s2, onehot, eps, eps_s1, K=3):
    """STAGE2_G.forward (S/model.py:371-442) -> (stage1_img, img, mu, logvar, local_labels)."""
    B, ngf, ef = z.shape[0], cfg.gf_dim, cfg.cond_dim
    with torch.no_grad():
        s1_img = stage1_g(net, cfg, z, tmi, onehot, text, eps_s1, pre="STAGE1_G.", K=K)[0]
    h = F.relu(conv(net, "encoder.0", s1_img))
    h = F.relu(bn(net, "encoder.3", conv(net, "encoder.2", h, 2, 1)))
    enc = F.relu(bn(net, "encoder.6", conv(net, "encoder.5", h, 2, 1)))
    c, mu, logvar = ca_net(net, "ca_net", text, eps, cfg)
    labels = []
    parts = [enc, tile(c, 16)]
    if cfg.use_bbox_layout:
        layout = z.new_zeros(B, ef, 16, 16)
        for k in range(K):
            lab = F.relu(bn(net, "label.1", F.linear(torch.cat((c, onehot[:, k]), 1), net["label.0.weight"])))
            labels.append(lab)
            layout = layout + stn(tile(lab, 16), tmi[:, k], (B, ef, 16, 16))
        parts.append(layout)
    h = F.relu(bn(net, "hr_joint.1", conv(net, "hr_joint.0", torch.cat(parts, 1))))
    for r in range(cfg.r_num):
        h = res_block(net, "residual.%d" % r, h)
    canvas = z.new_zeros(B, ngf, 64, 64)
    for k in range(K):
        if not cfg.use_bbox_layout:
            labels.append(F.relu(bn(net, "label.1",
                                    F.linear(torch.cat((c, onehot[:, k]), 1), net["label.0.weight"]))))
        patch = stn(h, tm_s2[:, k], (B, h.shape[1], 16, 16))
        x = relu_up(net, "local1", torch.cat((patch, tile(labels[k], 16)), 1))
        x = relu_up(net, "local2", x)
        canvas = canvas + stn(x, tmi_s2[:, k], (B, ngf, 64, 64))
    h = relu_up(net, "upsample1", h)
    h = relu_up(net, "upsample2", h)
    h = torch.cat((h, canvas), 1)
    h = relu_up(net, "upsample3", h)
    h = relu_up(net, "upsample4", h)
    return s1_img, torch.tanh(conv(net, "img.0", h)), mu, logvar, torch.stack(labels, 1)


def stage2_d(net, cfg, image, label, tm, tmi, K=3):
    """STAGE2_D._encode_img (S/model.py:482-532)."""
    B, ndf = image.shape[0], cfg.df_dim
    canvas = image.new_zeros(B, ndf * 2, 32, 32)
    for k in range(K):
        h = stn(image, tm[:, k], (B, image.shape[1], 32, 32))
        h = torch.cat((h, tile(label[:, k], 32)), 1)
        h = lrelu(bn(net, "local.1", conv(net, "local.0", h, 1, 1)))
        h = lrelu(bn(net, "local.4", conv(net, "local.3", h, 1, 1)))
        canvas = canvas + stn(h, tmi[:, k], (B, ndf * 2, 32, 32))
    h = lrelu(conv(net, "conv1", image, 2, 1))
    h = lrelu(bn(net, "bn2", conv(net, "conv2", h, 2, 1)))
    h = lrelu(bn(net, "bn3", conv(net, "conv3", h, 2, 1)))
    h = torch.cat((h, canvas), 1)
    for i in (4, 5, 6):
        h = lrelu(bn(net, "bn%d" % i, conv(net, "conv%d" % i, h, 2, 1)))
    for i in (7, 8):
        h = lrelu(bn(net, "bn%d" % i, conv(net, "conv%d" % i, h, 1, 1)))
    return h


def cond_logits(net, cfg, h, c, pre="get_cond_logits"):
    """D_GET_LOGITS (S/model.py:75-104): raw logits."""
    x = torch.cat((h, c.reshape(-1, cfg.d_cond, 1, 1).repeat(1, 1, 4, 4)), 1)
    x = lrelu(bn(net, pre + ".outlogits.1", conv(net, pre + ".outlogits.0", x)))
    return F.conv2d(x, net[pre + ".outlogits.3.weight"], net[pre + ".outlogits.3.bias"], 4).reshape(-1)


def uncond_logits(net, h, pre="get_uncond_logits"):
    return F.conv2d(h, net[pre + ".outlogits.0.weight"], net[pre + ".outlogits.0.bias"], 4).reshape(-1)


def condition(cfg, onehot, mu):
    """what the logits head is conditioned on (S/miscc/utils.py:75; C/miscc/utils.py:98-99; M:78)."""
    if cfg.text:
        return mu.detach()
    c = onehot.sum(1)
    return c.clamp(min=0) if cfg.tree == "clevr" else c


def _bce(logits, target):
    return F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, target))


def d_features(net, cfg, image, label, tm, tmi):
    return (stage2_d if cfg.stage == 2 else stage1_d)(net, cfg, image, label, tm, tmi)


def discriminator_loss(net, cfg, real, fake, label, tm, tmi, cond):
    """compute_discriminator_loss (S/miscc/utils.py:68-107, C:91-128, M:71-107)."""
    B = real.shape[0]
    fake, cond = fake.detach(), cond.detach()
    real_f = d_features(net, cfg, real, label, tm, tmi)
    fake_f = d_features(net, cfg, fake, label, tm, tmi)
    e_real = _bce(cond_logits(net, cfg, real_f, cond), 1.0)
    e_wrong = _bce(cond_logits(net, cfg, real_f[:B - 1], cond[1:]), 0.0)
    e_fake = _bce(cond_logits(net, cfg, fake_f, cond), 0.0)
    if cfg.stage == 2:
        u_real = _bce(uncond_logits(net, real_f), 1.0)
        u_fake = _bce(uncond_logits(net, fake_f), 0.0)
        err = (e_real + u_real) / 2. + (e_fake + e_wrong + u_fake) / 3.
        return err, (e_real + u_real) / 2., e_wrong, (e_fake + u_fake) / 2.
    return e_real + (e_fake + e_wrong) * 0.5, e_real, e_wrong, e_fake


def generator_loss(net, cfg, fake, label, tm, tmi, cond):
    """compute_generator_loss (S/miscc/utils.py:110-125, C:131-144, M:110-123)."""
    f = d_features(net, cfg, fake, label, tm, tmi)
    err = _bce(cond_logits(net, cfg, f, cond.detach()), 1.0)
    if cfg.stage == 2:
        err = err + _bce(uncond_logits(net, f), 1.0)
    return err


class TrainState:
    def __init__(self, net_g, net_d, cfg):
        self.g, self.d, self.cfg = net_g, net_d, cfg
        self.opt_g, self.opt_d = adam_state(net_g), adam_state(net_d)


def generate(st, b):
    cfg = st.cfg
    if cfg.stage == 2:
        _, fake, mu, logvar, _ = stage2_g(st.g, cfg, b["txt_embedding"], b["z"], b["tmi"], b["tm_s2"], b["tmi_s2"],
                                          b["label_one_hot"], b["eps"], b["eps_s1"])
    else:
        fake, mu, logvar, _ = stage1_g(st.g, cfg, b["z"], b["tmi"], b["label_one_hot"], b.get("txt_embedding"),
                                       b.get("eps"))
    return fake, mu, logvar


def train_step(st, b):
    """S/trainer.py:188-231 (C/trainer.py:127-157, M/trainer.py:131-160): G fwd once; D: zero_grad, loss,
    backward, Adam; G: zero_grad, loss through the updated D (+ KL * coeff), backward, Adam."""
    cfg = st.cfg
    fake, mu, logvar = generate(st, b)
    tm, tmi = (b["tm_s2"], b["tmi_s2"]) if cfg.stage == 2 else (b["tm"], b["tmi"])
    cond = condition(cfg, b["label_one_hot"], mu)
    zero_grad(st.d)
    err_d, e_real, e_wrong, e_fake = discriminator_loss(st.d, cfg, b["real_imgs"], fake, b["label_one_hot"], tm, tmi,
                                                        cond)
    err_d.backward()
    adam_step(st.d, st.opt_d, cfg.lr_d)
    zero_grad(st.g)
    err_g = generator_loss(st.d, cfg, fake, b["label_one_hot"], tm, tmi, cond)
    logs = dict(errD=float(err_d.detach()), errD_real=float(e_real.detach()), errD_wrong=float(e_wrong.detach()),
                errD_fake=float(e_fake.detach()), errG=float(err_g.detach()), fake=fake.detach())
    total = err_g
    if cfg.text:
        kl = kl_loss(mu, logvar)
        total = err_g + kl * cfg.kl_coeff
        logs["kl"] = float(kl.detach())
    total.backward()
    zero_grad(st.d)          # D gradients produced by the G backward are discarded (next netD.zero_grad())
    adam_step(st.g, st.opt_g, cfg.lr_g)
    return logs


# ------------------------------------------------------------- state_dict layouts (key -> shape)
_bn = A._bn


def _up(spec, pre, cin, cout):
    spec[pre + ".1.weight"] = (cout, cin, 3, 3)
    _bn(spec, pre + ".2", cout)


def _bbox_spec(s, pre, cfg):
    c = cfg.bbox_c
    s[pre + ".encode.0.weight"] = (c // 2, cfg.bbox_in, 3, 3)
    s[pre + ".encode.2.weight"] = (c // 4, c // 2, 3, 3)
    _bn(s, pre + ".encode.3", c // 4)
    s[pre + ".encode.5.weight"] = (c // 8, c // 4, 3, 3)
    _bn(s, pre + ".encode.6", c // 8)


def stage1_g_spec(cfg, pre=""):
    """Key names/shapes/order of STAGE1_G().state_dict() (S/model.py:156-199, C:121-156, M:121-156)."""
    s, ngf, ef = {}, cfg.gf_dim * 8, cfg.ef_dim
    ninput = cfg.z_dim + (ef if cfg.text else 0)
    if cfg.text:
        s[pre + "ca_net.fc.weight"], s[pre + "ca_net.fc.bias"] = (ef * 2, cfg.text_dim), (ef * 2,)
    if cfg.use_bbox_layout:
        _bbox_spec(s, pre + "bbox_net", cfg)
        ninput += (cfg.bbox_c // 8) * 4
    s[pre + "fc.0.weight"] = (ngf * 16, ninput)
    _bn(s, pre + "fc.1", ngf * 16)
    s[pre + "label.0.weight"] = (ef, (ef if cfg.text else 0) + cfg.label_dim)
    _bn(s, pre + "label.1", ef)
    _up(s, pre + "local1", ef, ngf // 2)
    _up(s, pre + "local2", ngf // 2, ngf // 4)
    _up(s, pre + "upsample1", ngf, ngf // 2)
    _up(s, pre + "upsample2", ngf // 2, ngf // 4)
    _up(s, pre + "upsample3", ngf // 2, ngf // 8)
    _up(s, pre + "upsample4", ngf // 8, ngf // 16)
    s[pre + "img.0.weight"] = (cfg.img_ch, ngf // 16, 3, 3)
    return s


def _logits_spec(s, pre, ndf, cond):
    s[pre + ".outlogits.0.weight"] = (ndf * 8, ndf * 8 + cond, 3, 3)
    _bn(s, pre + ".outlogits.1", ndf * 8)
    s[pre + ".outlogits.3.weight"] = (1, ndf * 8, 4, 4)
    s[pre + ".outlogits.3.bias"] = (1,)


def stage1_d_spec(cfg):
    """STAGE1_D().state_dict() (S/model.py:245-264, C:201-224, M:199-221)."""
    s, ndf = {}, cfg.df_dim
    s["local.0.weight"] = (ndf * 2, cfg.img_ch + cfg.label_dim, 4, 4)
    _bn(s, "local.1", ndf * 2)
    s["conv1.weight"] = (ndf, cfg.img_ch, 4, 4)
    s["conv2.weight"] = (ndf * 2, ndf, 4, 4)
    _bn(s, "bn2", ndf * 2)
    s["conv3.weight"] = (ndf * 4, ndf * 4, 4, 4)
    _bn(s, "bn3", ndf * 4)
    s["conv4.weight"] = (ndf * 8, ndf * 4, 4, 4)
    _bn(s, "bn4", ndf * 8)
    _logits_spec(s, "get_cond_logits", ndf, cfg.d_cond)
    return s


def stage2_g_spec(cfg):
    """STAGE2_G(STAGE1_G()).state_dict() (S/model.py:312-369)."""
    s, ngf, ef = {}, cfg.gf_dim, cfg.cond_dim
    s.update(stage1_g_spec(cfg, "STAGE1_G."))
    s["ca_net.fc.weight"], s["ca_net.fc.bias"] = (ef * 2, cfg.text_dim), (ef * 2,)
    s["label.0.weight"] = (ef, ef + cfg.label_dim)
    _bn(s, "label.1", ef)
    _up(s, "local1", ef + 768, ngf * 2)
    _up(s, "local2", ngf * 2, ngf)
    s["encoder.0.weight"] = (ngf, 3, 3, 3)
    s["encoder.2.weight"] = (ngf * 2, ngf, 4, 4)
    _bn(s, "encoder.3", ngf * 2)
    s["encoder.5.weight"] = (ngf * 4, ngf * 2, 4, 4)
    _bn(s, "encoder.6", ngf * 4)
    s["hr_joint.0.weight"] = (ngf * 4, (ef * 2 if cfg.use_bbox_layout else ef) + ngf * 4, 3, 3)
    _bn(s, "hr_joint.1", ngf * 4)
    for r in range(cfg.r_num):
        q = "residual.%d.block" % r
        s[q + ".0.weight"] = (ngf * 4, ngf * 4, 3, 3)
        _bn(s, q + ".1", ngf * 4)
        s[q + ".3.weight"] = (ngf * 4, ngf * 4, 3, 3)
        _bn(s, q + ".4", ngf * 4)
    _up(s, "upsample1", ngf * 4, ngf * 2)
    _up(s, "upsample2", ngf * 2, ngf)
    _up(s, "upsample3", ngf * 2, ngf // 2)
    _up(s, "upsample4", ngf // 2, ngf // 4)
    s["img.0.weight"] = (3, ngf // 4, 3, 3)
    return s


def stage2_d_spec(cfg):
    """STAGE2_D().state_dict() (S/model.py:452-480)."""
    s, ndf = {}, cfg.df_dim
    s["local.0.weight"] = (ndf * 2, 3 + cfg.label_dim, 4, 4)
    _bn(s, "local.1", ndf * 2)
    s["local.3.weight"] = (ndf * 2, ndf * 2, 4, 4)
    _bn(s, "local.4", ndf * 2)
    s["conv1.weight"] = (ndf, 3, 4, 4)
    for i, (cin, cout) in enumerate([(ndf, ndf * 2), (ndf * 2, ndf * 4), (ndf * 6, ndf * 8), (ndf * 8, ndf * 16),
                                     (ndf * 16, ndf * 32)]):
        s["conv%d.weight" % (i + 2)] = (cout, cin, 4, 4)
        _bn(s, "bn%d" % (i + 2), cout)
    s["conv7.weight"] = (ndf * 16, ndf * 32, 3, 3)
    _bn(s, "bn7", ndf * 16)
    s["conv8.weight"] = (ndf * 8, ndf * 16, 3, 3)
    _bn(s, "bn8", ndf * 8)
    _logits_spec(s, "get_cond_logits", ndf, cfg.cond_dim)
    s["get_uncond_logits.outlogits.0.weight"] = (1, ndf * 8, 4, 4)
    s["get_uncond_logits.outlogits.0.bias"] = (1,)
    return s


def init_state_dict(spec, seed=0):
    """weights_init of the family (S/miscc/utils.py:129-139): conv/linear weights ~ N(0, 0.02), BN gamma ~
    N(1, 0.02), biases 0."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in spec.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shp)
        elif k.endswith("running_mean") or k.endswith("bias"):
            sd[k] = torch.zeros(shp)
        elif len(shp) == 1:
            sd[k] = 1.0 + 0.02 * torch.randn(shp, generator=gen)
        else:
            sd[k] = 0.02 * torch.randn(shp, generator=gen)
    return sd
