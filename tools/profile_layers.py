"""Per-launch profile of the GEMM kernels over one eager train step -> CSV (gpurun_out/layers.csv)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import TrainEngine, build_networks
from mogan_amd.hip import lib
import bench
set_coco_train_defaults()
dev = torch.device("cuda", 0)
te, ie, G, Ds = build_networks(device=dev, seed=1234)
eng = TrainEngine(te, ie, G, Ds, use_graph=False)
batch, _ = bench.make_device_batch(16, 0, dev)
def step():
    b = dict(batch); b["z"] = torch.randn(16, 100, device=dev); b["eps"] = torch.randn(16, 100, device=dev)
    eng.step(b)
step(); step(); torch.cuda.synchronize()
lib.call("mogan_prof_enable", 1)
step(); torch.cuda.synchronize()
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "layers.csv")
lib.call("mogan_prof_dump", out.encode())
lib.call("mogan_prof_enable", 0)
print("wrote", out)
