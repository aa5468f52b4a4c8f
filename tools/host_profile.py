"""cProfile of the host side of one eager train step (who spends the ~50 ms of launch time?)."""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import TrainEngine, build_networks
os.environ.setdefault("MOGAN_FAST_INIT", "1")
device = torch.device("cuda", 0); torch.cuda.set_device(device)
set_coco_train_defaults()
te, ie, G, Ds = build_networks(device=device, seed=1)
eng = TrainEngine(te, ie, G, Ds)
batch, _ = bench.make_device_batch(16, 0, device)
def step():
    b = dict(batch); b["z"] = torch.randn(16, 100, device=device); b["eps"] = torch.randn(16, 100, device=device)
    return eng.step(b)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
