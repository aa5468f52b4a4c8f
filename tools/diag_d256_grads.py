"""Diagnostic: per-tensor D_NET256 gradient error of the HIP path and of the fp32 CPU oracle, both against the fp64 oracle
(full width, B=4, fixture inputs).  Usage: python tools/diag_d256_grads.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import det_fill_state, load_pkg, rel_l2
load_pkg()
from mogan_amd.attngan import model, synthetic
from mogan_amd.attngan.miscc import losses as L
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from oracle import attngan_oracle as O
set_coco_train_defaults()
B = 4
cpu = synthetic.make_batch(B, words_num=12, nef=256, seed=21)
bt = synthetic.to_device(cpu, "cuda")
G = model.G_NET(); sdg = det_fill_state(G, "G."); G = G.cuda().train()
with torch.no_grad():
    imgs, atts, mu, logvar = G(bt["z"], bt["sent_emb"], bt["words_embs"], bt["mask"], bt["tmi"], bt["label_one_hot"], eps=bt["eps"])
D = model.D_NET256(); sdd = det_fill_state(D, "D2."); D = D.cuda().train()
errD = L.discriminator_loss(D, bt["imgs"][2], imgs[2], bt["sent_emb"], None, None, None)
errD.backward(); torch.cuda.synchronize()
fake = imgs[2].detach().cpu()
res = {}
for dt in (torch.float32, torch.float64):
    od = O.from_state_dict(sdd, dtype=dt)
    c = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in cpu.items()}
    c["imgs"] = [t.to(dt) for t in cpu["imgs"]]
    e = O.discriminator_loss(2, od, c["imgs"][2], fake.to(dt), c["sent_emb"], c, O.Cfg())
    e.backward()
    res[dt] = (float(e), od)
print("errD hip %.8f o32 %.8f o64 %.8f" % (float(errD), res[torch.float32][0], res[torch.float64][0]))
for k, p in D.named_parameters():
    g64 = res[torch.float64][1][k].grad
    g32 = res[torch.float32][1][k].grad
    print("%-40s hip-vs-f64 %.2e   o32-vs-f64 %.2e   hip-vs-o32 %.2e   |g| %.3e" % (k, rel_l2(p.grad, g64), rel_l2(g32, g64), rel_l2(p.grad, g32), float(g64.norm())))
