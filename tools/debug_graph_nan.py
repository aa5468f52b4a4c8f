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
eng = TrainEngine(te, ie, G, Ds, use_graph=True)
batch, _ = bench.make_device_batch(16, 0, dev)
gen = torch.Generator(device=dev).manual_seed(1000)
for s in range(int(sys.argv[1])):
    b = dict(batch)
    b["z"] = torch.randn(16, 100, device=dev, generator=gen)
    b["eps"] = torch.randn(16, 100, device=dev, generator=gen)
    logs = eng.step(b)
torch.cuda.synchronize()
names = ["pG","pD0","pD1","pD2","gG","gD0","gD1","gD2","z","eps","words","sent","tm","tmi","fake64","errD0","g_loss0","kl"]
for i, f in enumerate(getattr(eng, "_nan_log", [])):
    print(i, [n for n, v in zip(names, f.tolist()) if v])
print({k: round(float(v), 4) for k, v in logs.items() if v.dim() == 0})
