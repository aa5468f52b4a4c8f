"""Device-synchronised wall time per phase of the eager train step (TrainEngine.phase_times = True; serialises the phases)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import TrainEngine, build_networks
import bench
set_coco_train_defaults()
dev = torch.device("cuda", 0)
te, ie, G, Ds = build_networks(device=dev, seed=1234)
eng = TrainEngine(te, ie, G, Ds, use_graph=False)
eng.phase_times = True
batch, _ = bench.make_device_batch(16, 0, dev)
def step():
    b = dict(batch); b["z"] = torch.randn(16, 100, device=dev); b["eps"] = torch.randn(16, 100, device=dev)
    eng.step(b)
for _ in range(4): step()
eng._ph = {}; eng._ph_last = None
N = 10
for _ in range(N): step()
tot = 0
for k, v in eng._ph.items():
    print("%-28s %6.2f ms" % (k, v / N)); tot += v / N
print("%-28s %6.2f ms" % ("sum", tot))
