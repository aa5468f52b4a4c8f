"""How much of a step is host launch time?  Times run_step() on the host without synchronizing, then the
synchronized wall time (eager mode)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import TrainEngine, build_networks

os.environ.setdefault("MOGAN_FAST_INIT", "1")
device = torch.device("cuda", 0)
torch.cuda.set_device(device)
set_coco_train_defaults()
B = 16
dist_on = bool(os.environ.get("MOGAN_FORCE_DIST"))
if dist_on:
    import torch.distributed as dist
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, **({"device_id": device} if os.environ.get("MOGAN_LAB_DEVID") else {}))
te, ie, G, Ds = build_networks(device=device, seed=1)
eng = TrainEngine(te, ie, G, Ds, distributed=dist_on and not os.environ.get('MOGAN_LAB_NODIST_ENGINE'))
batch, _ = bench.make_device_batch(B, 0, device)
def step():
    b = dict(batch); b["z"] = torch.randn(B, 100, device=device); b["eps"] = torch.randn(B, 100, device=device)
    return eng.step(b)
for _ in range(3): step()
torch.cuda.synchronize()
host = []; wall = []
for _ in range(8):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(t1 - t0); wall.append(t2 - t0)
print("host launch ms/step", [round(h * 1e3, 1) for h in host])
print("wall ms/step       ", [round(w * 1e3, 1) for w in wall])
# steady state (no sync between steps), like bench.py
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("steady: host enqueue %.1f ms/step, wall %.1f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
