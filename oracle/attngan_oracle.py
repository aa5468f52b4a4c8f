"""CPU oracle: a functional torch-CPU (fp32 or fp64) restatement of the reference's AttnGAN
G+D train step.  TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the checker / reported CPU baseline; never by the product
package (multiple-objects-gan_amd/), which must fail loudly when libmogan_hip.so is missing.

Pinned against tests/golden/*.npz, which tests/golden/make_golden.py captured from the
reference's own python (code/coco/attngan/...) run on torch-CPU in the build container.
The arithmetic of conv/BN/grid_sample/softmax itself lives in PyTorch (reference pins
torch==0.4.1, requirements.txt:29; not vendored): the two known version-dependent semantics
are explicit flags here -- `align_corners` (default False = the reference as run
on torch>=1.3) and the Adam epsilon placement (`adam_step`).

Every network is a plain dict {state_dict key -> tensor} using the reference's key names.
All citations are relative to /root/reference/code/coco/attngan/.
"""
import math

import torch
import torch.nn.functional as F

MAX_OBJECTS = 3          # model.py:14
ALIGN_CORNERS = False


class Cfg:
    """The cfg fields the path reads (miscc/config.py:9-64 + cfg/coco_train.yml)."""
    def __init__(self, gf_dim=48, df_dim=96, z_dim=100, cond_dim=100, emb_dim=256, r_num=3,
                 words_num=12, gamma1=4.0, gamma2=5.0, gamma3=10.0, lam=50.0, branch_num=3,
                 lr_g=2e-4, lr_d=2e-4):
        self.gf_dim, self.df_dim, self.z_dim, self.cond_dim = gf_dim, df_dim, z_dim, cond_dim
        self.emb_dim, self.r_num, self.words_num = emb_dim, r_num, words_num
        self.gamma1, self.gamma2, self.gamma3, self.lam = gamma1, gamma2, gamma3, lam
        self.branch_num, self.lr_g, self.lr_d = branch_num, lr_g, lr_d


def from_state_dict(sd, dtype=torch.float32, requires_grad=True):
    """Clone a state_dict into an oracle net: float params become autograd leaves, BN
    running buffers stay plain tensors (updated in place by `bn`)."""
    net = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point():
            t = t.to(dtype)
            buf = k.endswith("running_mean") or k.endswith("running_var")
            if requires_grad and not buf:
                t.requires_grad_(True)
        net[k] = t
    return net


def parameters(net):
    return [(k, v) for k, v in net.items() if torch.is_tensor(v) and v.requires_grad]


# ---------------------------------------------------------------------------- primitives
def stn(x, theta, size, align_corners=None):
    """model.py:17-21: affine_grid + grid_sample (bilinear, zero padding)."""
    ac = ALIGN_CORNERS if align_corners is None else align_corners
    grid = F.affine_grid(theta.to(x.dtype), list(size), align_corners=ac)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=ac)


def glu(x):
    """model.py:24-32."""
    c = x.shape[1] // 2
    return x[:, :c] * torch.sigmoid(x[:, c:])


BN_TRAINING = True       # module switch: False = netG.eval() (running statistics), used by the sampling-path tests


def bn(net, key, x, training=None, eps=1e-5):
    """nn.BatchNorm1d/2d in train mode: batch statistics, running stats momentum 0.1
    (unbiased variance into running_var), num_batches_tracked += 1; eval mode: running statistics."""
    training = BN_TRAINING if training is None else training
    rm, rv = net[key + ".running_mean"], net[key + ".running_var"]
    y = F.batch_norm(x, rm, rv, net[key + ".weight"], net[key + ".bias"], training, 0.1, eps)
    if training and (key + ".num_batches_tracked") in net:
        net[key + ".num_batches_tracked"] += 1
    return y


def conv(net, key, x, stride=1, pad=1):
    return F.conv2d(x, net[key + ".weight"], net.get(key + ".bias"), stride, pad)


def up_block(net, pre, x):
    """model.py:48-55: nearest x2 -> conv3x3 -> BN -> GLU (Sequential indices 0..3)."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return glu(bn(net, pre + ".2", conv(net, pre + ".1", x)))


def res_block(net, pre, x):
    """model.py:67-81."""
    h = glu(bn(net, pre + ".block.1", conv(net, pre + ".block.0", x)))
    h = bn(net, pre + ".block.4", conv(net, pre + ".block.3", h))
    return h + x


LRELU_MASKS = None      # test hook, see lrelu
LRELU_FLIPS = None      # test hook: list of (decisions that differ from the oracle's own sign, decisions, max |x| among them)


def lrelu(x):
    """nn.LeakyReLU(0.2).  Test hook: when LRELU_MASKS is a list of boolean tensors, the sign decision of each call is
    taken from the next entry (the decisions the checked implementation made) instead of from x itself -- the gradient
    of the SAME piecewise-linear function; pre-activations within fp32 noise of the kink otherwise flip a few of the
    millions of decisions per layer and move the weight gradients below by ~1e-3."""
    if LRELU_MASKS is not None:
        m = LRELU_MASKS.pop(0)
        assert m.shape == x.shape, (tuple(m.shape), tuple(x.shape))
        if LRELU_FLIPS is not None:      # an imposed mask must only differ where x is within rounding of the kink
            diff = m != (x.detach() > 0)
            n = int(diff.sum())
            LRELU_FLIPS.append((n, x.numel(), float(x.detach()[diff].abs().max()) if n else 0.0))
        return torch.where(m, x, 0.2 * x)
    return F.leaky_relu(x, 0.2)


# ------------------------------------------------------------------------------ generator
def ca_net(net, pre, sent_emb, eps, cfg):
    """model.py:317-345; eps ~ N(0,1) is drawn inside forward there, injected here."""
    x = glu(F.linear(sent_emb, net[pre + ".fc.weight"], net[pre + ".fc.bias"]))
    mu, logvar = x[:, :cfg.cond_dim], x[:, cfg.cond_dim:]
    c = eps * torch.exp(0.5 * logvar) + mu
    return c, mu, logvar


def bbox_net(net, pre, labels, tmi, cfg):
    """model.py:84-116."""
    B = labels.shape[0]
    layout = labels.new_zeros(B, cfg.cond_dim, 16, 16)
    for k in range(MAX_OBJECTS):
        lab = labels[:, k].reshape(B, -1, 1, 1).repeat(1, 1, 16, 16)
        layout = layout + stn(lab, tmi[:, k], lab.shape)
    h = lrelu(conv(net, pre + ".encode.0", layout, 2, 1))
    h = lrelu(bn(net, pre + ".encode.3", conv(net, pre + ".encode.2", h, 2, 1)))
    h = lrelu(bn(net, pre + ".encode.6", conv(net, pre + ".encode.5", h, 2, 1)))
    return h.reshape(B, -1)


def init_stage_g(net, pre, z, c, tmi, onehot, cfg):
    """model.py:348-422 (object pathway x3 with per-call BN statistics: SURVEY.md F11)."""
    B, ngf = z.shape[0], cfg.gf_dim * 16
    local_labels = []
    h_locals = z.new_zeros(B, ngf // 4, 16, 16)
    for k in range(MAX_OBJECTS):
        lab = F.linear(torch.cat((c, onehot[:, k]), 1), net[pre + ".label.0.weight"])
        lab = F.relu(bn(net, pre + ".label.1", lab))
        local_labels.append(lab)
        h = lab.reshape(B, -1, 1, 1).repeat(1, 1, 4, 4)
        h = up_block(net, pre + ".local1", h)
        h = up_block(net, pre + ".local2", h)
        h_locals = h_locals + stn(h, tmi[:, k], h.shape)
    local_labels = torch.stack(local_labels, 1)
    code = bbox_net(net, pre + ".bbox_net", local_labels, tmi, cfg)
    h = F.linear(torch.cat((c, z, code), 1), net[pre + ".fc.0.weight"])
    h = glu(bn(net, pre + ".fc.1", h)).reshape(B, ngf, 4, 4)
    h = up_block(net, pre + ".upsample1", h)
    h = up_block(net, pre + ".upsample2", h)
    h = torch.cat((h, h_locals), 1)
    h = up_block(net, pre + ".upsample3", h)
    return up_block(net, pre + ".upsample4", h)


def global_attention(net, pre, h, context, mask):
    """GlobalAttention.py:72-123, including the mask-indexing behaviour of line 104-108:
    `mask.repeat(queryL, 1)` is laid over rows ordered b*queryL+q, so row r is masked with
    mask[r mod B]."""
    B, idf, ih, iw = h.shape
    Q, T = ih * iw, context.shape[2]
    target = h.reshape(B, idf, Q).transpose(1, 2)                       # B,Q,idf
    src = F.conv2d(context.unsqueeze(3), net[pre + ".conv_context.weight"]).squeeze(3)  # B,idf,T
    attn = torch.bmm(target, src).reshape(B * Q, T)
    if mask is not None:
        rows = torch.arange(B * Q) % B
        attn = attn.masked_fill(mask[rows], -float("inf"))
    attn = torch.softmax(attn, 1).reshape(B, Q, T).transpose(1, 2)      # B,T,Q
    wc = torch.bmm(src, attn)
    return wc.reshape(B, idf, ih, iw), attn.reshape(B, T, ih, iw)


def next_stage_g(net, pre, h, words, mask, cfg):
    """model.py:425-461."""
    c, att = global_attention(net, pre + ".att", h, words, mask)
    x = torch.cat((h, c), 1)
    for r in range(cfg.r_num):
        x = res_block(net, "%s.residual.%d" % (pre, r), x)
    return up_block(net, pre + ".upsample", x), att


def get_image(net, pre, h):
    """model.py:464-475."""
    return torch.tanh(conv(net, pre + ".img.0", h))


def g_net(net, cfg, z, sent_emb, words, mask, tmi, onehot, eps):
    """model.py:478-528."""
    c, mu, logvar = ca_net(net, "ca_net", sent_emb, eps, cfg)
    h1 = init_stage_g(net, "h_net1", z, c, tmi, onehot, cfg)
    imgs, atts, hs = [get_image(net, "img_net1", h1)], [], [h1]
    h = h1
    for s in (2, 3)[:cfg.branch_num - 1]:
        h, att = next_stage_g(net, "h_net%d" % s, h, words, mask, cfg)
        imgs.append(get_image(net, "img_net%d" % s, h))
        atts.append(att)
        hs.append(h)
    return imgs, atts, mu, logvar, hs


# -------------------------------------------------------------------------- discriminators
def down(net, pre, x):
    """downBlock / Block3x3_leakRelu (model.py:575-591): conv(.0) + BN(.1) + LeakyReLU."""
    k = net[pre + ".0.weight"].shape[-1]
    s, p = (2, 1) if k == 4 else (1, 1)
    return lrelu(bn(net, pre + ".1", conv(net, pre + ".0", x, s, p)))


def encode16(net, pre, x):
    """encode_image_by_16times (model.py:595-613)."""
    x = lrelu(conv(net, pre + ".0", x, 2, 1))
    for c, b in ((2, 3), (5, 6), (8, 9)):
        x = lrelu(bn(net, "%s.%d" % (pre, b), conv(net, "%s.%d" % (pre, c), x, 2, 1)))
    return x


def d_net64(net, image, label, tm, tmi, cfg):
    """model.py:646-711."""
    B = image.shape[0]
    h_locals = image.new_zeros(B, cfg.df_dim * 2, 16, 16)
    for k in range(MAX_OBJECTS):
        lab = label[:, k].reshape(B, 81, 1, 1).repeat(1, 1, 16, 16)
        h = stn(image, tm[:, k], (B, image.shape[1], 16, 16))
        h = torch.cat((h, lab), 1)
        h = lrelu(bn(net, "local.1", conv(net, "local.0", h, 1, 1)))
        h_locals = h_locals + stn(h, tmi[:, k], (B, h.shape[1], 16, 16))
    h = lrelu(conv(net, "conv1", image, 2, 1))
    h = lrelu(bn(net, "bn2", conv(net, "conv2", h, 2, 1)))
    h = torch.cat((h, h_locals), 1)
    h = lrelu(bn(net, "bn3", conv(net, "conv3", h, 2, 1)))
    return lrelu(bn(net, "bn4", conv(net, "conv4", h, 2, 1)))


def d_net128(net, x):
    """model.py:715-734."""
    x = encode16(net, "img_code_s16", x)
    x = down(net, "img_code_s32", x)
    return down(net, "img_code_s32_1", x)


def d_net256(net, x):
    """model.py:738-760."""
    x = encode16(net, "img_code_s16", x)
    x = down(net, "img_code_s32", x)
    x = down(net, "img_code_s64", x)
    x = down(net, "img_code_s64_1", x)
    return down(net, "img_code_s64_2", x)


def d_features(i, net, img, batch, cfg):
    if i == 0:
        return d_net64(net, img, batch["label_one_hot"], batch["tm"], batch["tmi"], cfg)
    return d_net128(net, img) if i == 1 else d_net256(net, img)


def d_logits(net, pre, h, c=None):
    """D_GET_LOGITS (model.py:616-642); conv 4x4 s4 has a bias; sigmoid output."""
    if c is not None:
        cc = c.reshape(c.shape[0], -1, 1, 1).repeat(1, 1, 4, 4)
        h = down(net, pre + ".jointConv", torch.cat((h, cc), 1))
    out = F.conv2d(h, net[pre + ".outlogits.0.weight"], net[pre + ".outlogits.0.bias"], 4)
    return torch.sigmoid(out).reshape(-1)


# ---------------------------------------------------------------------------------- losses
def bce(p, target):
    """nn.BCELoss (mean), as the reference calls it (miscc/losses.py:158-168): torch clamps the logs at -100 in the forward
    pass and differentiates (p - t) / max(p (1 - p), 1e-12) in the backward pass.  (A composition of torch.log and
    torch.clamp has the same forward but a NaN gradient, 0 * inf, wherever a saturated discriminator outputs exactly 0 or
    1 -- which it does after a few dozen steps on one synthetic batch, bench.py's parity leg.)"""
    return F.binary_cross_entropy(p, target.to(p.dtype).expand_as(p))


def discriminator_loss(i, net, real, fake, cond, batch, cfg):
    """miscc/losses.py:136-174. D(real) and D(fake.detach()) are separate calls (separate BN
    batch statistics); wrong pairs = real[:B-1] vs cond[1:]."""
    B = real.shape[0]
    ones, zeros = real.new_ones(B), real.new_zeros(B)
    rf = d_features(i, net, real, batch, cfg)
    ff = d_features(i, net, fake.detach(), batch, cfg)
    c_real = bce(d_logits(net, "COND_DNET", rf, cond), ones)
    c_fake = bce(d_logits(net, "COND_DNET", ff, cond), zeros)
    c_wrong = bce(d_logits(net, "COND_DNET", rf[:B - 1], cond[1:B]), zeros[1:B])
    u_real = bce(d_logits(net, "UNCOND_DNET", rf), ones)
    u_fake = bce(d_logits(net, "UNCOND_DNET", ff), zeros)
    return (u_real + c_real) / 2.0 + (u_fake + c_fake + c_wrong) / 3.0


def func_attention(query, context, gamma1):
    """GlobalAttention.py:31-69."""
    B, T = query.shape[0], query.shape[2]
    ih, iw = context.shape[2], context.shape[3]
    S = ih * iw
    ctx = context.reshape(B, -1, S)
    attn = torch.bmm(ctx.transpose(1, 2), query)                 # B,S,T
    attn = torch.softmax(attn.reshape(B * S, T), 1).reshape(B, S, T)
    attn = attn.transpose(1, 2).reshape(B * T, S) * gamma1
    attn = torch.softmax(attn, 1).reshape(B, T, S)
    wc = torch.bmm(ctx, attn.transpose(1, 2))                    # B,C,T
    return wc, attn.reshape(B, T, ih, iw)


def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """miscc/losses.py:11-17."""
    w12 = (x1 * x2).sum(dim)
    return w12 / (x1.norm(2, dim) * x2.norm(2, dim)).clamp(min=eps)


def words_loss(feat, words, cap_lens, cfg, class_ids=None):
    """miscc/losses.py:62-132. class_ids mask is a no-op for COCO (ids unique per sample)
    but is honoured when given."""
    B = feat.shape[0]
    sims, att_maps = [], []
    lens = [int(v) for v in cap_lens]
    for i in range(B):
        n = lens[i]
        word = words[i, :, :n].unsqueeze(0).repeat(B, 1, 1)
        wc, attn = func_attention(word, feat, cfg.gamma1)
        att_maps.append(attn[i].unsqueeze(0))
        w = word.transpose(1, 2).reshape(B * n, -1)
        c = wc.transpose(1, 2).reshape(B * n, -1)
        row = cosine_similarity(w, c).reshape(B, n)
        row = torch.log(torch.exp(row * cfg.gamma2).sum(1, keepdim=True))
        sims.append(row)
    sim = torch.cat(sims, 1) * cfg.gamma3
    if class_ids is not None:
        ids = torch.as_tensor(class_ids)
        m = (ids[:, None] == ids[None, :]) & ~torch.eye(B, dtype=torch.bool)
        sim = sim.masked_fill(m, -float("inf"))
    labels = torch.arange(B)
    return F.cross_entropy(sim, labels), F.cross_entropy(sim.t(), labels), att_maps


def sent_loss(code, sent, cfg, class_ids=None, eps=1e-8):
    """miscc/losses.py:20-59."""
    B = code.shape[0]
    n0 = code.norm(2, 1, keepdim=True)
    n1 = sent.norm(2, 1, keepdim=True)
    s = code @ sent.t() / (n0 @ n1.t()).clamp(min=eps) * cfg.gamma3
    if class_ids is not None:
        ids = torch.as_tensor(class_ids)
        m = (ids[:, None] == ids[None, :]) & ~torch.eye(B, dtype=torch.bool)
        s = s.masked_fill(m, -float("inf"))
    labels = torch.arange(B)
    return F.cross_entropy(s, labels), F.cross_entropy(s.t(), labels)


def generator_loss(nets_d, image_encoder, fakes, batch, cfg):
    """miscc/losses.py:177-226."""
    B = fakes[0].shape[0]
    ones = fakes[0].new_ones(B)
    total, logs = 0.0, {}
    for i, net in enumerate(nets_d):
        f = d_features(i, net, fakes[i], batch, cfg)
        g = bce(d_logits(net, "COND_DNET", f, batch["sent_emb"]), ones) \
            + bce(d_logits(net, "UNCOND_DNET", f), ones)
        total = total + g
        logs["g_loss%d" % i] = g
        if i == len(nets_d) - 1:
            feat, code = image_encoder(fakes[i])
            w0, w1, _ = words_loss(feat, batch["words_embs"], batch["cap_lens"], cfg,
                                   batch.get("class_ids"))
            s0, s1 = sent_loss(code, batch["sent_emb"], cfg, batch.get("class_ids"))
            logs["w_loss"], logs["s_loss"] = (w0 + w1) * cfg.lam, (s0 + s1) * cfg.lam
            total = total + logs["w_loss"] + logs["s_loss"]
    return total, logs


def kl_loss(mu, logvar):
    """miscc/losses.py:230-234."""
    return -0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp())


# ------------------------------------------------------------------------------ optimizer
def adam_state(net):
    return {"step": 0, "m": {k: torch.zeros_like(v) for k, v in parameters(net)},
            "v": {k: torch.zeros_like(v) for k, v in parameters(net)}}


def adam_step(net, st, lr, beta1=0.5, beta2=0.999, eps=1e-8, eps_mode="torch2"):
    """torch.optim.Adam as the reference configures it (trainer.py:137-148).
    eps_mode 'torch2': denom = sqrt(v)/sqrt(1-b2^t) + eps (the oracle-as-run-here);
    'torch041': denom = sqrt(v) + eps, step = lr*sqrt(1-b2^t)/(1-b1^t) (the pinned version)."""
    st["step"] += 1
    t = st["step"]
    bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
    with torch.no_grad():
        for k, p in parameters(net):
            if p.grad is None:
                continue
            g, m, v = p.grad, st["m"][k], st["v"][k]
            m.mul_(beta1).add_(g, alpha=1 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            if eps_mode == "torch2":
                denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(m, denom, value=-lr / bc1)
            else:
                denom = v.sqrt().add_(eps)
                p.addcdiv_(m, denom, value=-lr * math.sqrt(bc2) / bc1)


def zero_grad(net):
    for _, p in parameters(net):
        p.grad = None


# ------------------------------------------------------------------------------ text encoder
def _lstm_direction(x, lens, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of a one-layer LSTM over right-padded sequences x (B,T,I), what nn.LSTM does with a
    PackedSequence (model.py:183-188): every sample is advanced over ITS OWN valid steps only (forward 0..len-1, reverse
    len-1..0), outputs beyond len stay zero, the returned hidden state is the one after the sample's last step.
    Gate order of the stacked weights: input, forget, cell, output."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = x.new_zeros(B, T, H)
    lens = torch.as_tensor(lens)
    for step in range(T):
        t = lens - 1 - step if reverse else torch.full_like(lens, step)
        live = (t >= 0) & (t < lens)                                   # samples that still have a token at this step
        idx = t.clamp(min=0)
        xt = x[torch.arange(B), idx]
        gates = xt @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, g, o = gates.chunk(4, 1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        m = live.unsqueeze(1)
        c = torch.where(m, c_new, c)
        h = torch.where(m, h_new, h)
        out[torch.arange(B)[live], idx[live]] = h_new[live]
    return out, h


def rnn_encoder(net, captions, cap_lens):
    """RNN_ENCODER.forward in eval mode (model.py:176-204; dropout inactive): embedding -> bidirectional one-layer LSTM
    over the packed captions -> words_emb (B, 2H, max_len), sent_emb (B, 2H) = [last forward h | last reverse h].
    `net`: state_dict of the module (keys encoder.weight, rnn.weight_ih_l0, ..., rnn.bias_hh_l0_reverse)."""
    emb = net["encoder.weight"][captions]                              # (B,T,ninput)
    lens = [int(v) for v in cap_lens]
    emb = emb[:, :max(lens)]
    of, hf = _lstm_direction(emb, lens, net["rnn.weight_ih_l0"], net["rnn.weight_hh_l0"], net["rnn.bias_ih_l0"],
                             net["rnn.bias_hh_l0"], False)
    ob, hb = _lstm_direction(emb, lens, net["rnn.weight_ih_l0_reverse"], net["rnn.weight_hh_l0_reverse"],
                             net["rnn.bias_ih_l0_reverse"], net["rnn.bias_hh_l0_reverse"], True)
    return torch.cat([of, ob], 2).transpose(1, 2), torch.cat([hf, hb], 1)


# ------------------------------------------------------------------------------ train step
class TrainState:
    def __init__(self, net_g, nets_d, cfg):
        self.g, self.ds, self.cfg = net_g, nets_d, cfg
        self.opt_g = adam_state(net_g)
        self.opt_ds = [adam_state(d) for d in nets_d]
        self.ema = [p.detach().clone() for _, p in parameters(net_g)]   # copy_G_params


def train_step(st, batch, image_encoder):
    """One iteration of the loop in trainer.py:281-342 (text embeddings precomputed):
    G fwd once -> for each D: zero_grad, discriminator_loss, backward, Adam -> generator_loss
    + KL through the *updated* Ds, backward, Adam -> EMA(0.999)."""
    cfg = st.cfg
    fakes, _, mu, logvar, _ = g_net(st.g, cfg, batch["z"], batch["sent_emb"], batch["words_embs"],
                                    batch["mask"], batch["tmi"], batch["label_one_hot"], batch["eps"])
    logs = {}
    for i, d in enumerate(st.ds):
        zero_grad(d)
        err = discriminator_loss(i, d, batch["imgs"][i], fakes[i], batch["sent_emb"], batch, cfg)
        err.backward()
        adam_step(d, st.opt_ds[i], cfg.lr_d)
        logs["errD%d" % i] = float(err.detach())
    zero_grad(st.g)
    err_g, glogs = generator_loss(st.ds, image_encoder, fakes, batch, cfg)
    kl = kl_loss(mu, logvar)
    err_g = err_g + kl
    err_g.backward()
    adam_step(st.g, st.opt_g, cfg.lr_g)
    with torch.no_grad():
        for (_, p), a in zip(parameters(st.g), st.ema):
            a.mul_(0.999).add_(p, alpha=0.001)
    logs.update(errG=float(err_g.detach()), kl=float(kl.detach()), fake64=fakes[0].detach(), fake_last=fakes[-1].detach())
    logs.update({k: float(v.detach()) for k, v in glogs.items()})
    return logs


# ------------------------------------------------------------- state_dict layouts (key -> shape)
def _bn(spec, key, c):
    for s, shp in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)),
                   ("running_var", (c,)), ("num_batches_tracked", ())):
        spec["%s.%s" % (key, s)] = shp


def _up(spec, pre, cin, cout):
    spec[pre + ".1.weight"] = (cout * 2, cin, 3, 3)
    _bn(spec, pre + ".2", cout * 2)


def g_net_spec(cfg):
    """Key names/shapes/order of G_NET().state_dict() (model.py:478-495)."""
    s, ngf, c = {}, cfg.gf_dim * 16, cfg.cond_dim
    s["ca_net.fc.weight"], s["ca_net.fc.bias"] = (c * 4, cfg.emb_dim), (c * 4,)
    p = "h_net1"
    s[p + ".bbox_net.encode.0.weight"] = (c // 2, c, 3, 3)
    s[p + ".bbox_net.encode.2.weight"] = (c // 4, c // 2, 3, 3)
    _bn(s, p + ".bbox_net.encode.3", c // 4)
    s[p + ".bbox_net.encode.5.weight"] = (c // 8, c // 4, 3, 3)
    _bn(s, p + ".bbox_net.encode.6", c // 8)
    s[p + ".fc.0.weight"] = (ngf * 32, cfg.z_dim + c + 48)
    _bn(s, p + ".fc.1", ngf * 32)
    s[p + ".label.0.weight"] = (100, 181)
    _bn(s, p + ".label.1", 100)
    _up(s, p + ".local1", 100, ngf // 2)
    _up(s, p + ".local2", ngf // 2, ngf // 4)
    _up(s, p + ".upsample1", ngf, ngf // 2)
    _up(s, p + ".upsample2", ngf // 2, ngf // 4)
    _up(s, p + ".upsample3", ngf // 2, ngf // 8)
    _up(s, p + ".upsample4", ngf // 8, ngf // 16)
    s["img_net1.img.0.weight"] = (3, cfg.gf_dim, 3, 3)
    g = cfg.gf_dim
    for st in (2, 3)[:cfg.branch_num - 1]:
        p = "h_net%d" % st
        s[p + ".att.conv_context.weight"] = (g, cfg.emb_dim, 1, 1)
        for r in range(cfg.r_num):
            q = "%s.residual.%d.block" % (p, r)
            s[q + ".0.weight"] = (g * 4, g * 2, 3, 3)
            _bn(s, q + ".1", g * 4)
            s[q + ".3.weight"] = (g * 2, g * 2, 3, 3)
            _bn(s, q + ".4", g * 2)
        _up(s, p + ".upsample", g * 2, g)
        s["img_net%d.img.0.weight" % st] = (3, g, 3, 3)
    return s


def _logits(spec, ndf, nef):
    spec["UNCOND_DNET.outlogits.0.weight"] = (1, ndf * 8, 4, 4)
    spec["UNCOND_DNET.outlogits.0.bias"] = (1,)
    spec["COND_DNET.jointConv.0.weight"] = (ndf * 8, ndf * 8 + nef, 3, 3)
    _bn(spec, "COND_DNET.jointConv.1", ndf * 8)
    spec["COND_DNET.outlogits.0.weight"] = (1, ndf * 8, 4, 4)
    spec["COND_DNET.outlogits.0.bias"] = (1,)


def _enc16(spec, pre, ndf):
    spec[pre + ".0.weight"] = (ndf, 3, 4, 4)
    for c, b, m in ((2, 3, 1), (5, 6, 2), (8, 9, 4)):
        spec["%s.%d.weight" % (pre, c)] = (ndf * m * 2, ndf * m, 4, 4)
        _bn(spec, "%s.%d" % (pre, b), ndf * m * 2)


def _blk(spec, pre, cin, cout, k):
    spec[pre + ".0.weight"] = (cout, cin, k, k)
    _bn(spec, pre + ".1", cout)


def d_net_spec(i, cfg):
    """Key names/shapes/order of D_NET64/128/256().state_dict() (model.py:646-760)."""
    s, ndf, nef = {}, cfg.df_dim, cfg.emb_dim
    if i == 0:
        _logits(s, ndf, nef)
        s["conv1.weight"] = (ndf, 3, 4, 4)
        s["conv2.weight"] = (ndf * 2, ndf, 4, 4)
        _bn(s, "bn2", ndf * 2)
        s["conv3.weight"] = (ndf * 4, ndf * 4, 4, 4)
        _bn(s, "bn3", ndf * 4)
        s["conv4.weight"] = (ndf * 8, ndf * 4, 4, 4)
        _bn(s, "bn4", ndf * 8)
        s["local.0.weight"] = (ndf * 2, 84, 4, 4)
        _bn(s, "local.1", ndf * 2)
        return s
    _enc16(s, "img_code_s16", ndf)
    _blk(s, "img_code_s32", ndf * 8, ndf * 16, 4)
    if i == 1:
        _blk(s, "img_code_s32_1", ndf * 16, ndf * 8, 3)
    else:
        _blk(s, "img_code_s64", ndf * 16, ndf * 32, 4)
        _blk(s, "img_code_s64_1", ndf * 32, ndf * 16, 3)
        _blk(s, "img_code_s64_2", ndf * 16, ndf * 8, 3)
    _logits(s, ndf, nef)
    return s


def init_state_dict(spec, seed=0, gain=1.0):
    """weights_init-like random init for standalone oracle runs (miscc/utils.py:321-331):
    orthogonal conv/linear weights, BN gamma ~ N(1, 0.02), beta 0."""
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
            w = torch.randn(shp[0], int(torch.tensor(shp[1:]).prod()), generator=gen)
            rows, cols = w.shape
            q, r = torch.linalg.qr(w.t() if rows < cols else w)
            q = q * torch.sign(torch.diagonal(r)).unsqueeze(0)
            sd[k] = (gain * (q.t() if rows < cols else q)).reshape(shp).contiguous()
    return sd
