#!/usr/bin/env python
"""Golden fixtures for the StackGAN-family trees (tests/golden/stackgan_<case>.npz), captured by running the
REFERENCE's own python -- code/coco/stackgan/{model.py,miscc/utils.py}, code/clevr/..., code/multi-mnist/... --
on CPU in the build container through ref_shim.load_tree.   Usage: python tests/golden/make_golden_stackgan.py

Fixtures are data only.  Inputs (mogan_amd.stackgan.synthetic.make_batch) and weights
(helpers.det_fill_state) are regenerated deterministically by the tests; the .npz files hold the reference's
outputs: small tensors in full, large ones as `probe` summaries.  The step loop restates
S/trainer.py:188-231 (C/trainer.py:127-157, M/trainer.py:131-160; the trainer modules are py2-only) while
calling the reference's own networks and compute_*_loss functions.

Widths: the three model.py files hard-code a few sizes that pin some cfg values even in a reduced case --
coco `ninput += 64` needs CONDITION_DIM 128, coco stage II `upBlock(ef_dim + 768, ...)` needs GF_DIM 192,
clevr `ninput += 8` needs CONDITION_DIM 16.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import ref_shim                                   # noqa: E402
from helpers import det_array, det_fill_state, probe  # noqa: E402
import mogan_loader                               # noqa: E402
mogan_loader.load()
from mogan_amd.stackgan import synthetic          # noqa: E402

torch.set_num_threads(8)

# case -> (tree, stage, batch, cfg)
CASES = {
    "coco_s1": ("coco", 1, 3, dict(GF_DIM=4, DF_DIM=4, CONDITION_DIM=128, TEXT_DIM=12)),
    "coco_s2": ("coco", 2, 2, dict(GF_DIM=192, DF_DIM=4, CONDITION_DIM=128, TEXT_DIM=12, R_NUM=1)),
    "clevr": ("clevr", 1, 3, dict(GF_DIM=4, DF_DIM=4, CONDITION_DIM=16)),
    "mnist": ("mnist", 1, 3, dict(GF_DIM=4, DF_DIM=4, CONDITION_DIM=128)),
}


def T(name, shape, scale=1.0):
    return torch.from_numpy(det_array(name, shape, scale))


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-22s %7.1f KB  (%d arrays)" % (name, os.path.getsize(path) / 1024, len(out)))


def grads_probe(module, prefix):
    return {prefix + k.replace(".", "__"): probe(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def state_probe(module, prefix):
    return {prefix + k.replace(".", "__"): probe(v.float()) for k, v in module.state_dict().items()}


def build(ns, tree, stage):
    if stage == 2:
        G = ns.model.STAGE2_G(ns.model.STAGE1_G())
        D = ns.model.STAGE2_D()
    else:
        G, D = ns.model.STAGE1_G(), ns.model.STAGE1_D()
    det_fill_state(G, "G.")
    det_fill_state(D, "D.")
    G.train()
    D.train()
    return G, D


def inject_eps(G, bt, stage):
    """eps is drawn inside CA_NET.reparametrize (S/model.py:60-67): inject the batch's."""
    if stage == 2:
        e1 = bt["eps_s1"]
        G.STAGE1_G.ca_net.reparametrize = lambda mu, logvar: e1.mul(logvar.mul(0.5).exp()).add(mu)
    e = bt["eps"]
    G.ca_net.reparametrize = lambda mu, logvar: e.mul(logvar.mul(0.5).exp()).add(mu)


def run_g(G, tree, stage, bt):
    """-> (fake, mu, logvar, local_labels, stage1_img)"""
    if tree == "coco" and stage == 2:
        s1, fake, mu, logvar, ll = G(bt["txt_embedding"], bt["z"], bt["tmi"], bt["tm_s2"], bt["tmi_s2"],
                                     bt["label_one_hot"])
        return fake, mu, logvar, ll, s1
    if tree == "coco":
        _, fake, mu, logvar, ll = G(bt["txt_embedding"], bt["z"], bt["tmi"], bt["label_one_hot"])
        return fake, mu, logvar, ll, None
    out = G(bt["z"], bt["tmi"], bt["label_one_hot"])
    return (out[1] if isinstance(out, tuple) else out), None, None, None, None


def d_mats(bt, stage):
    return (bt["tm_s2"], bt["tmi_s2"]) if stage == 2 else (bt["tm"], bt["tmi"])


def sub(img):
    s = max(1, img.shape[-1] // 16)
    return img[:, :, ::s, ::s]


def gen_nets(ns, case, tree, stage, B):
    bt = synthetic.make_batch(tree, B, stage=stage, seed=21, text_dim=12)
    G, D = build(ns, tree, stage)
    out = {"g_keys": np.array(list(G.state_dict().keys())), "d_keys": np.array(list(D.state_dict().keys()))}
    # reference bbox -> theta of this tree on the batch's boxes (mnist computes in float64)
    bb = bt["bbox_s2" if stage == 2 else "bbox"].view(-1, 4)
    out["ref_tm"] = ns.utils.compute_transformation_matrix(bb).float()
    out["ref_tmi"] = ns.utils.compute_transformation_matrix_inverse(bb).float()
    if tree == "coco":
        inject_eps(G, bt, stage)
    z = bt["z"].clone().requires_grad_(True)
    bz = dict(bt, z=z)
    fake, mu, logvar, ll, s1 = run_g(G, tree, stage, bz)
    loss = (fake * T("G.gimg", fake.shape)).sum()
    if mu is not None:
        loss = loss + (mu * T("G.gmu", mu.shape)).sum() + (logvar * T("G.glv", logvar.shape)).sum()
    loss.backward()
    out.update(fake_sub=sub(fake), fake_p=probe(fake))
    if z.grad is not None:                 # stage II: z only feeds the detached stage-I image
        out["dz"] = z.grad
    if mu is not None:
        out.update(mu=mu, logvar=logvar, local_labels=ll)
    if s1 is not None:
        out.update(s1_sub=sub(s1), s1_p=probe(s1))
    out.update(grads_probe(G, "gg_"))
    out.update({k: v for k, v in state_probe(G, "gs_").items() if "running" in k})
    # discriminator: features + both logits heads, gradient w.r.t. the image and the parameters
    tm, tmi = d_mats(bt, stage)
    x = bt["real_imgs"].clone().requires_grad_(True)
    f = D(x, bt["label_one_hot"], tm, tmi)
    if tree == "coco":
        cond = T("D.cond", (B, 128), 0.5)
    else:
        cond = bt["label_one_hot"].sum(1)
    c = D.get_cond_logits(f, cond)
    cw = D.get_cond_logits(f[:B - 1], cond[1:])
    loss = (f * T("D.gf", f.shape)).sum() + (c * T("D.gc", c.shape)).sum() + (cw * T("D.gcw", cw.shape)).sum()
    out.update(d_feat=f, d_cond=c, d_wrong=cw)
    if D.get_uncond_logits is not None:
        u = D.get_uncond_logits(f)
        loss = loss + (u * T("D.gu", u.shape)).sum()
        out["d_uncond"] = u
    loss.backward()
    out.update(d_dx_sub=sub(x.grad), d_dx_p=probe(x.grad))
    out.update(grads_probe(D, "dg_"))
    out.update({k: v for k, v in state_probe(D, "ds_").items() if "running" in k})
    save("stackgan_%s_nets" % case, **out)


def train_step(ns, tree, stage, G, D, optG, optD, bt):
    """S/trainer.py:188-231, C/trainer.py:127-157, M/trainer.py:131-160."""
    B = bt["z"].shape[0]
    real_labels, fake_labels = torch.ones(B), torch.zeros(B)
    if tree == "coco":
        inject_eps(G, bt, stage)
    fake, mu, logvar, _, _ = run_g(G, tree, stage, bt)
    tm, tmi = d_mats(bt, stage)
    D.zero_grad()
    args = (D, bt["real_imgs"], fake, real_labels, fake_labels, bt["label_one_hot"], tm, tmi)
    args = args + ((mu, [0]) if tree == "coco" else ([0],))
    errD, e_real, e_wrong, e_fake = ns.utils.compute_discriminator_loss(*args)
    errD.backward(retain_graph=True)
    optD.step()
    G.zero_grad()
    args = (D, fake, real_labels, bt["label_one_hot"], tm, tmi) + ((mu, [0]) if tree == "coco" else ([0],))
    errG = ns.utils.compute_generator_loss(*args)
    logs = dict(errD=errD.item(), errD_real=e_real, errD_wrong=e_wrong, errD_fake=e_fake, errG=errG.item())
    total = errG
    if tree == "coco":
        kl = ns.utils.KL_loss(mu, logvar)
        total = errG + kl * ns.cfg.TRAIN.COEFF.KL
        logs["kl"] = kl.item()
    total.backward()
    optG.step()
    logs["fake"] = fake.detach()
    return logs


def gen_step(ns, case, tree, stage, B):
    G, D = build(ns, tree, stage)
    optD = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    optG = torch.optim.Adam([p for p in G.parameters() if p.requires_grad], lr=2e-4, betas=(0.5, 0.999))
    out = {}
    for step in range(2):
        bt = synthetic.make_batch(tree, B, stage=stage, seed=300 + step, text_dim=12)
        logs = train_step(ns, tree, stage, G, D, optG, optD, bt)
        p = "s%d_" % step
        for k, v in logs.items():
            if k != "fake":
                out[p + k] = v
        out[p + "fake_sub"] = sub(logs["fake"])
        out[p + "fake_p"] = probe(logs["fake"])
        out.update(state_probe(G, p + "G_"))
        out.update(state_probe(D, p + "D_"))
        print(case, "step", step, {k: v for k, v in logs.items() if k != "fake"})
    save("stackgan_%s_step" % case, **out)


def main():
    only = sys.argv[1:]
    for case, (tree, stage, B, kw) in CASES.items():
        if only and case not in only:
            continue
        ns = ref_shim.load_tree(tree)
        ref_shim.set_tree_cfg(ns.cfg, tree, stage, **kw)
        gen_nets(ns, case, tree, stage, B)
        gen_step(ns, case, tree, stage, B)


if __name__ == "__main__":
    main()
