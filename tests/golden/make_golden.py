#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/*.npz by running the REFERENCE's own
python (code/coco/attngan/{model,GlobalAttention}.py, miscc/{losses,utils}.py) on CPU in the
build container through tests/golden/ref_shim.py.   Usage:  python tests/golden/make_golden.py

Fixtures are data only: inputs are regenerated deterministically by the tests
(tests/helpers.det_array / det_fill_state, mogan_amd.attngan.synthetic.make_batch), the
.npz files hold the reference's outputs (full tensors when small, `probe` summaries
otherwise).  The train-step loop body restates code/coco/attngan/trainer.py:281-342
(trainer.py itself is py2-only) while calling the reference's own model/loss functions.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import ref_shim                                   # noqa: E402
from helpers import det_array, det_fill_state, probe  # noqa: E402
from standin import StandInEncoder                # noqa: E402
import mogan_loader                               # noqa: E402
mogan_loader.load()
from mogan_amd.attngan import synthetic           # noqa: E402

torch.set_num_threads(4)
SMALL = dict(GF_DIM=4, DF_DIM=4, EMBEDDING_DIM=16, WORDS_NUM=5, R_NUM=2)


def T(name, shape, scale=1.0, shift=0.0):
    return torch.from_numpy(det_array(name, shape, scale, shift))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-14s %7.1f KB  (%d arrays)" % (name, os.path.getsize(path) / 1024, len(out)))


def grads_probe(module, prefix):
    return {prefix + k.replace(".", "__"): probe(p.grad) for k, p in module.named_parameters()
            if p.grad is not None}


def state_probe(module, prefix):
    return {prefix + k.replace(".", "__"): probe(v.float()) for k, v in module.state_dict().items()}


# ------------------------------------------------------------------------------------------
def gen_theta(ns):
    bbox = np.array([[0.1, 0.2, 0.3, 0.4], [0.0, 0.0, 1.0, 1.0], [0.45, 0.05, 0.5, 0.9],
                     [-1, -1, -1, -1], [0.3, 0.6, 0.12, 0.399]], dtype=np.float32)
    b = torch.from_numpy(bbox)
    save("theta", bbox=bbox, tm=ns.utils.compute_transformation_matrix(b),
         tmi=ns.utils.compute_transformation_matrix_inverse(b))


def gen_stn():
    out = {}
    for ac in (False, True):
        ns = ref_shim.load(align_corners=ac)
        bbox = torch.tensor([[0.1, 0.2, 0.3, 0.4], [-1, -1, -1, -1], [0.4, 0.1, 0.55, 0.8]])
        tm = ns.utils.compute_transformation_matrix(bbox)
        tmi = ns.utils.compute_transformation_matrix_inverse(bbox)
        for tag, theta, insz, outsz in (("paste", tmi, (3, 5, 8, 8), (3, 5, 8, 8)),
                                        ("crop", tm, (3, 2, 12, 10), (3, 2, 6, 7)),
                                        ("rot", None, (3, 4, 7, 9), (3, 4, 5, 6))):
            if theta is None:   # a general (non axis-aligned) affine map
                theta = T("stn.rot.theta", (3, 2, 3), 0.5) + torch.tensor([[1., 0, 0], [0, 1., 0]])
            x = T("stn.%s.x" % tag, insz).requires_grad_(True)
            g = T("stn.%s.g" % tag, outsz)
            y = ns.model.stn(x, theta, outsz)
            y.backward(g)
            key = "%s_ac%d_" % (tag, int(ac))
            out[key + "theta"] = theta
            out[key + "y"] = y
            out[key + "dx"] = x.grad
    ref_shim.load(align_corners=False)
    save("stn", **out)


def gen_blocks(ns):
    out = {}
    ns.cfg.GAN.R_NUM = 2
    x = T("blocks.x", (4, 8, 8, 8)).requires_grad_(True)
    y = ns.model.GLU()(x)
    y.backward(T("blocks.glu.g", y.shape))
    out.update(glu_y=y, glu_dx=x.grad)
    for tag, mod, gshape in (("up", ns.model.upBlock(8, 4), (4, 4, 16, 16)),
                             ("res", ns.model.ResBlock(8), (4, 8, 8, 8)),
                             ("lrelu3", ns.model.Block3x3_leakRelu(8, 6), (4, 6, 8, 8)),
                             ("down", ns.model.downBlock(8, 6), (4, 6, 4, 4))):
        det_fill_state(mod, "blocks.%s." % tag)
        mod.train()
        x = T("blocks.x", (4, 8, 8, 8)).requires_grad_(True)
        y = mod(x)
        y.backward(T("blocks.%s.g" % tag, gshape))
        out[tag + "_y"] = y
        out[tag + "_dx"] = x.grad
        for k, p in mod.named_parameters():
            out["%s_d_%s" % (tag, k.replace(".", "__"))] = p.grad
        for k, v in mod.state_dict().items():
            if "running" in k:
                out["%s_s_%s" % (tag, k.replace(".", "__"))] = v
    save("blocks", **out)


def gen_attn(ns):
    out = {}
    for B in (3, 4):
        att = ns.GlobalAttention.GlobalAttentionGeneral(6, 10)
        det_fill_state(att, "attn.")
        h = T("attn.h%d" % B, (B, 6, 4, 4)).requires_grad_(True)
        ctx = T("attn.ctx%d" % B, (B, 10, 5)).requires_grad_(True)
        lens = [5, 4, 2, 3][:B]
        mask = torch.zeros(B, 5, dtype=torch.bool)
        for b in range(B):
            mask[b, lens[b]:] = True
        att.applyMask(mask)
        wc, a = att(h, ctx)
        (wc * T("attn.gw%d" % B, wc.shape)).sum().add((a * T("attn.ga%d" % B, a.shape)).sum()).backward()
        p = "b%d_" % B
        out.update({p + "mask": mask.numpy(), p + "wc": wc, p + "attn": a, p + "dh": h.grad,
                    p + "dctx": ctx.grad, p + "dw": att.conv_context.weight.grad})
    # func_attention (DAMSM)
    q = T("fattn.q", (2, 8, 4)).requires_grad_(True)
    c = T("fattn.c", (2, 8, 3, 3)).requires_grad_(True)
    wc, a = ns.GlobalAttention.func_attention(q, c, 4.0)
    (wc * T("fattn.gw", wc.shape)).sum().backward()
    out.update(f_wc=wc, f_attn=a, f_dq=q.grad, f_dc=c.grad)
    save("attn", **out)


def _g_inputs(B, nef, T_):
    return synthetic.make_batch(B, words_num=T_, nef=nef, seed=11)


def gen_gnet(ns):
    ref_shim.set_cfg(ns.cfg, **SMALL)
    B = 3
    bt = _g_inputs(B, SMALL["EMBEDDING_DIM"], SMALL["WORDS_NUM"])
    G = ns.model.G_NET()
    det_fill_state(G, "G.")
    G.train()
    # eps is drawn inside CA_NET.forward (model.py:333-340): inject ours
    eps = bt["eps"]
    G.ca_net.reparametrize = lambda mu, logvar: eps.mul(logvar.mul(0.5).exp()).add(mu)
    inter = {}
    G.h_net1.bbox_net.register_forward_hook(lambda m, i, o: inter.__setitem__("bbox_code", o))
    G.h_net1.register_forward_hook(lambda m, i, o: inter.__setitem__("h_code1", o))
    G.h_net2.register_forward_hook(lambda m, i, o: inter.__setitem__("h_code2", o[0]))
    z = bt["z"].clone().requires_grad_(True)
    sent = bt["sent_emb"].clone().requires_grad_(True)
    words = bt["words_embs"].clone().requires_grad_(True)
    imgs, atts, mu, logvar = G(z, sent, words, bt["mask"], bt["tmi"], bt["label_one_hot"])
    loss = sum((im * T("G.gimg%d" % i, im.shape)).sum() for i, im in enumerate(imgs))
    loss = loss + (mu * T("G.gmu", mu.shape)).sum() + (logvar * T("G.glv", logvar.shape)).sum()
    loss.backward()
    out = dict(mu=mu, logvar=logvar, bbox_code=inter["bbox_code"],
               h_code1=probe(inter["h_code1"]), h_code2=probe(inter["h_code2"]),
               img64=imgs[0], img128=imgs[1][:, :, ::2, ::2], img256=imgs[2][:, :, ::4, ::4],
               img128_p=probe(imgs[1]), img256_p=probe(imgs[2]),
               att64=atts[0][:, :, ::4, ::4], att128=atts[1][:, :, ::8, ::8],
               att64_p=probe(atts[0]), att128_p=probe(atts[1]),
               dz=z.grad, dsent=sent.grad, dwords=words.grad)
    out.update(grads_probe(G, "g_"))
    out.update({k: v for k, v in state_probe(G, "s_").items() if "running" in k})
    save("gnet", **out)


def gen_gnet_eval(ns):
    """netG.eval() forward (trainer.py:398,431-437 `sampling`): BatchNorm on its running statistics."""
    ref_shim.set_cfg(ns.cfg, **SMALL)
    bt = _g_inputs(3, SMALL["EMBEDDING_DIM"], SMALL["WORDS_NUM"])
    G = ns.model.G_NET()
    det_fill_state(G, "G.")
    G.eval()
    eps = bt["eps"]
    G.ca_net.reparametrize = lambda mu, logvar: eps.mul(logvar.mul(0.5).exp()).add(mu)
    with torch.no_grad():
        imgs, atts, mu, logvar = G(bt["z"], bt["sent_emb"], bt["words_embs"], bt["mask"], bt["tmi"], bt["label_one_hot"])
    save("gnet_eval", img64=imgs[0], img128=imgs[1][:, :, ::2, ::2], img256=imgs[2][:, :, ::4, ::4],
         img256_p=probe(imgs[2]), att128_p=probe(atts[1]), mu=mu)


def gen_dnets(ns):
    ref_shim.set_cfg(ns.cfg, **SMALL)
    B = 3
    bt = _g_inputs(B, SMALL["EMBEDDING_DIM"], SMALL["WORDS_NUM"])
    out = {}
    for i, cls in enumerate((ns.model.D_NET64, ns.model.D_NET128, ns.model.D_NET256)):
        D = cls()
        det_fill_state(D, "D%d." % i)
        D.train()
        x = bt["imgs"][i].clone().requires_grad_(True)
        if i == 0:
            f = D(x, bt["label_one_hot"], bt["tm"], bt["tmi"])
        else:
            f = D(x)
        c = D.COND_DNET(f, bt["sent_emb"])
        u = D.UNCOND_DNET(f)
        cw = D.COND_DNET(f[:B - 1], bt["sent_emb"][1:B])
        loss = (f * T("D%d.gf" % i, f.shape)).sum() + (c * T("D%d.gc" % i, c.shape)).sum() \
            + (u * T("D%d.gu" % i, u.shape)).sum() + (cw * T("D%d.gcw" % i, cw.shape)).sum()
        loss.backward()
        p = "d%d_" % i
        out.update({p + "feat": f, p + "cond": c, p + "uncond": u, p + "wrong": cw,
                    p + "dx_p": probe(x.grad), p + "dx": x.grad[:, :, ::(4 << i), ::(4 << i)]})
        out.update(grads_probe(D, p + "g_"))
        out.update({k: v for k, v in state_probe(D, p + "s_").items() if "running" in k})
    save("dnets", **out)


def _build_all(ns, nef):
    G = ns.model.G_NET()
    det_fill_state(G, "G.")
    Ds = []
    for i, cls in enumerate((ns.model.D_NET64, ns.model.D_NET128, ns.model.D_NET256)):
        D = cls()
        det_fill_state(D, "D%d." % i)
        Ds.append(D)
    enc = StandInEncoder(nef)
    det_fill_state(enc, "ENC.")
    for p in enc.parameters():
        p.requires_grad = False
    enc.eval()
    return G, Ds, enc


def gen_losses(ns):
    ref_shim.set_cfg(ns.cfg, **SMALL)
    nef, T_ = SMALL["EMBEDDING_DIM"], SMALL["WORDS_NUM"]
    B = 4
    bt = synthetic.make_batch(B, words_num=T_, nef=nef, seed=5)
    G, Ds, enc = _build_all(ns, nef)
    real_labels, fake_labels = torch.ones(B), torch.zeros(B)
    match = torch.arange(B)
    fakes = [T("L.fake%d" % i, im.shape, 0.5).requires_grad_(True) for i, im in enumerate(bt["imgs"])]
    out = {}
    for i, D in enumerate(Ds):
        D.train()
        kw = dict(local_labels=bt["label_one_hot"], transf_matrices=bt["tm"],
                  transf_matrices_inv=bt["tmi"]) if i == 0 else {}
        errD = ns.losses.discriminator_loss(D, bt["imgs"][i], fakes[i], bt["sent_emb"],
                                            real_labels, fake_labels, [0], **kw)
        errD.backward()
        out["errD%d" % i] = errD
        out.update(grads_probe(D, "d%d_g_" % i))
        D.zero_grad()
    errG, logs = ns.losses.generator_loss(Ds, enc, fakes, real_labels, bt["words_embs"],
                                          bt["sent_emb"], match, bt["cap_lens"], bt["class_ids"], [0],
                                          local_labels=bt["label_one_hot"], transf_matrices=bt["tm"],
                                          transf_matrices_inv=bt["tmi"])
    errG.backward()
    out["errG"] = errG
    for i, f in enumerate(fakes):
        out["dfake%d_p" % i] = probe(f.grad)
        out["dfake%d" % i] = f.grad[:, :, ::(4 << i), ::(4 << i)]
    # DAMSM terms on their own (words_loss / sent_loss: losses.py:20-132)
    feat = T("L.feat", (B, nef, 17, 17)).requires_grad_(True)
    code = T("L.code", (B, nef)).requires_grad_(True)
    w0, w1, att = ns.losses.words_loss(feat, bt["words_embs"], match, bt["cap_lens"], bt["class_ids"], B)
    s0, s1 = ns.losses.sent_loss(code, bt["sent_emb"], match, bt["class_ids"], B)
    (w0 + 2 * w1 + 3 * s0 + 4 * s1).backward()
    out.update(w0=w0, w1=w1, s0=s0, s1=s1, dfeat=feat.grad, dcode=code.grad,
               watt0=att[0], watt3=att[3])
    mu = T("L.mu", (B, 100), 0.5).requires_grad_(True)
    lv = T("L.lv", (B, 100), 0.5).requires_grad_(True)
    kl = ns.losses.KL_loss(mu, lv)
    kl.backward()
    out.update(kl=kl, dmu=mu.grad, dlv=lv.grad)
    save("losses", **out)


def train_step(ns, G, Ds, enc, optG, optDs, avg_param_G, bt, z, eps):
    """The body of the `while step < num_batches` loop, code/coco/attngan/trainer.py:281-342,
    with injected noise/eps and precomputed text embeddings."""
    B = z.shape[0]
    real_labels, fake_labels, match = torch.ones(B), torch.zeros(B), torch.arange(B)
    G.ca_net.reparametrize = lambda mu, logvar: eps.mul(logvar.mul(0.5).exp()).add(mu)
    fake_imgs, _, mu, logvar = G(z, bt["sent_emb"], bt["words_embs"], bt["mask"], bt["tmi"],
                                 bt["label_one_hot"])
    logs = {}
    for i, D in enumerate(Ds):
        D.zero_grad()
        kw = dict(local_labels=bt["label_one_hot"], transf_matrices=bt["tm"],
                  transf_matrices_inv=bt["tmi"]) if i == 0 else {}
        errD = ns.losses.discriminator_loss(D, bt["imgs"][i], fake_imgs[i], bt["sent_emb"],
                                            real_labels, fake_labels, [0], **kw)
        errD.backward()
        optDs[i].step()
        logs["errD%d" % i] = errD.item()
    G.zero_grad()
    errG, _ = ns.losses.generator_loss(Ds, enc, fake_imgs, real_labels, bt["words_embs"],
                                       bt["sent_emb"], match, bt["cap_lens"], bt["class_ids"], [0],
                                       local_labels=bt["label_one_hot"], transf_matrices=bt["tm"],
                                       transf_matrices_inv=bt["tmi"])
    kl = ns.losses.KL_loss(mu, logvar)
    errG = errG + kl
    errG.backward()
    optG.step()
    for p, avg_p in zip(G.parameters(), avg_param_G):
        avg_p.mul_(0.999).add_(p.data, alpha=0.001)
    logs["errG"] = errG.item()
    logs["kl"] = kl.item()
    logs["fake64"] = fake_imgs[0].detach()
    return logs


def gen_step(ns):
    ref_shim.set_cfg(ns.cfg, **SMALL)
    nef, T_ = SMALL["EMBEDDING_DIM"], SMALL["WORDS_NUM"]
    B = 4
    G, Ds, enc = _build_all(ns, nef)
    G.train()
    [D.train() for D in Ds]
    optDs = [torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)) for D in Ds]
    optG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    avg = ns.utils.copy_G_params(G)
    out = {}
    for step in range(2):
        bt = synthetic.make_batch(B, words_num=T_, nef=nef, seed=100 + step)
        logs = train_step(ns, G, Ds, enc, optG, optDs, avg, bt, bt["z"], bt["eps"])
        p = "s%d_" % step
        for k in ("errD0", "errD1", "errD2", "errG", "kl"):
            out[p + k] = logs[k]
        out[p + "fake64"] = logs["fake64"]
        out.update(state_probe(G, p + "G_"))
        for i, D in enumerate(Ds):
            out.update(state_probe(D, p + "D%d_" % i))
        for (k, _), a in zip(G.named_parameters(), avg):
            out[p + "ema_" + k.replace(".", "__")] = probe(a)
        print("step", step, {k: v for k, v in logs.items() if k != "fake64"})
    save("step", **out)


def main():
    ns = ref_shim.load()
    ref_shim.set_cfg(ns.cfg, **SMALL)
    gen_theta(ns)
    gen_stn()
    ns = ref_shim.load()
    gen_blocks(ns)
    gen_attn(ns)
    gen_gnet(ns)
    gen_gnet_eval(ns)
    gen_dnets(ns)
    gen_losses(ns)
    gen_step(ns)


if __name__ == "__main__":
    main()
