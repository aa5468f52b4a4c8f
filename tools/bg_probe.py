import os, sys, time, torch
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MOGAN_FAST_INIT", "1")
import bench
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import TrainEngine, build_networks
device = torch.device("cuda", 0); torch.cuda.set_device(device)
set_coco_train_defaults()
te, ie, G, Ds = build_networks(device=device, seed=1)
eng = TrainEngine(te, ie, G, Ds)
B = 16
batch, _ = bench.make_device_batch(B, 0, device)
def step():
    b = dict(batch); b["z"] = torch.randn(B, 100, device=device); b["eps"] = torch.randn(B, 100, device=device)
    return eng.step(b)
for _ in range(4): step()
torch.cuda.synchronize()
# time the graph replays on the host
import torch.cuda
orig = torch.cuda.CUDAGraph.replay
acc = []
def timed(self):
    t0 = time.perf_counter(); orig(self); acc.append(time.perf_counter() - t0)
torch.cuda.CUDAGraph.replay = timed
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
n = len(acc) // 10
print("replays per step", n, "host ms in replays per step", sum(acc) / 10 * 1e3, "each(ms):", [round(a * 1e3, 2) for a in acc[-n:]])
print("host enqueue %.1f ms/step, wall %.1f" % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
