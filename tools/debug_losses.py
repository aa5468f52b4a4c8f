"""Debug helper: per-step losses of the full-size synthetic train loop (eager or graph)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import TrainEngine, build_networks
import bench
set_coco_train_defaults()
graph = "--graph" in sys.argv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
te, ie, G, Ds = build_networks(device=dev, seed=1234)
eng = TrainEngine(te, ie, G, Ds, use_graph=graph)
batch, _ = bench.make_device_batch(16, 0, dev)
gen = torch.Generator(device=dev).manual_seed(1000)
for s in range(steps):
    b = dict(batch)
    b["z"] = torch.randn(16, 100, device=dev, generator=gen)
    b["eps"] = torch.randn(16, 100, device=dev, generator=gen)
    logs = eng.step(b)
    msg = " ".join("%s=%.4g" % (k, float(v)) for k, v in logs.items() if torch.is_tensor(v) and v.dim() == 0)
    gn = float(eng.optG.g.norm()); pn = float(eng.optG.p.norm())
    dn = [float(o.g.norm()) for o in eng.optDs]
    print(s, msg, "| |gG|=%.3g |pG|=%.4g |gD|=%s fake64 absmax=%.3g" % (gn, pn, ["%.3g" % d for d in dn], float(logs["fake64"].abs().max())), flush=True)
