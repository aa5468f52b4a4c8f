"""Would running D(real) and D(fake) as ONE 2B batch (per-half BN statistics) pay?  Times forward + backward of each
discriminator at B = 32 against twice B = 16 (BN statistics aside, the same work)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.trainer import build_networks
from mogan_amd.hip import ops
set_coco_train_defaults()
dev = "cuda"
te, ie, G, Ds = build_networks(device=dev, seed=1)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for i, D in ((1, Ds[1]), (2, Ds[2])):
    S = 64 << i
    x16a, x16b = torch.randn(16, 3, S, S, device=dev), torch.randn(16, 3, S, S, device=dev)
    x32 = torch.cat([x16a, x16b])
    def run(xs):
        for p in D.parameters(): p.grad = None
        for x in xs:
            D(x).square().mean().backward()
    t2, t1 = t(lambda: run([x16a, x16b])), t(lambda: run([x32]))
    print("D%d: 2 x B16 %.2f ms | 1 x B32 %.2f ms (%.0f %%)" % (64 << i, t2, t1, 100 * t1 / t2))
