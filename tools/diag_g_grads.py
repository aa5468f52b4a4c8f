"""Diagnostic: generator gradients of one reduced-width engine step against the oracle's step (same weights, batch)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import det_fill_state, det_state, load_pkg, rel_l2
from standin import StandInEncoder
load_pkg()
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dp_worker
from mogan_amd.attngan import model, synthetic
from mogan_amd.attngan.trainer import TrainEngine
from oracle import attngan_oracle as O
dp_worker.small_cfg()
dt = torch.float64 if "f64" in sys.argv else torch.float32
G = model.G_NET(); sdg = det_fill_state(G, "G.")
Ds, sdd = [], []
for i, cls in enumerate((model.D_NET64, model.D_NET128, model.D_NET256)):
    D = cls(); sdd.append(det_fill_state(D, "D%d." % i)); Ds.append(D.cuda().train())
enc = StandInEncoder(16); det_fill_state(enc, "ENC.")
for p in enc.parameters(): p.requires_grad = False
eng = TrainEngine(None, enc.cuda().eval(), G.cuda().train(), Ds, use_graph=False)
cpu = synthetic.make_batch(4, words_num=5, nef=16, seed=100)
logs = eng.step(synthetic.to_device(cpu, "cuda")); torch.cuda.synchronize()
cfg = O.Cfg(gf_dim=4, df_dim=4, emb_dim=16, r_num=2, words_num=5)
enc_cpu = StandInEncoder(16); det_fill_state(enc_cpu, "ENC."); enc_cpu.eval().to(dt)
for p in enc_cpu.parameters(): p.requires_grad = False
st = O.TrainState(O.from_state_dict(sdg, dtype=dt), [O.from_state_dict(s, dtype=dt) for s in sdd], cfg)
c = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in cpu.items()}
c["imgs"] = [t.to(dt) for t in cpu["imgs"]]
ol = O.train_step(st, c, enc_cpu)
print({k: (float(logs[k]), ol[k]) for k in ("errD0", "errD1", "errD2", "errG", "kl")})
tot_bad = tot = 0
for (k, p), off in zip(G.named_parameters(), eng.optG.offsets):
    g = eng.optG.g[off:off + p.numel()].view_as(p).cpu()
    og = st.g[k].grad
    sign_bad = float(((g.double() * og.double()) < 0).double().mean())
    tot_bad += int(((g.double() * og.double()) < 0).sum()); tot += g.numel()
    print("%-42s rel-L2 %.2e  sign mismatch %.3f  |g| %.2e" % (k, rel_l2(g, og), sign_bad, float(og.norm())))
print("total sign mismatch", tot_bad / tot)
