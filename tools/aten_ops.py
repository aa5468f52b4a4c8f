"""Which stock torch ops still launch kernels in one eager train step (torch.profiler, CPU-side op names + counts)."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOGAN_FAST_INIT", "1"); os.environ["MOGAN_BRANCH_GRAPHS"] = "0"; os.environ["MOGAN_GRAPH_ENCODER"] = "0"
import bench
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import TrainEngine, build_networks
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
# ops that directly own device kernels
cnt = collections.Counter(); where = collections.defaultdict(collections.Counter)
for e in ev:
    if e.kernels:
        cnt[e.name] += len(e.kernels)
        st = [s for s in (e.stack or []) if "multiple-objects-gan_amd" in s or "bench.py" in s]
        where[e.name][st[0].split("multiple-objects-gan_amd/")[-1] if st else "?"] += len(e.kernels)
for n, c in cnt.most_common(25):
    print("%4d  %-28s %s" % (c, n, dict(where[n].most_common(5))))
