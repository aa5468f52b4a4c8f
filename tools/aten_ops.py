"""Which stock torch ops still launch kernels in one eager train step, and from where: a TorchDispatchMode that records, for the
ops that own device kernels (copy_, cat, add_, fill_, ...), the innermost frame of this package on the python stack (ops issued by the
autograd engine's own thread have none: "autograd").  python tools/aten_ops.py"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOGAN_FAST_INIT", "1")
os.environ["MOGAN_BRANCH_GRAPHS"] = "0"
os.environ["MOGAN_GRAPH_ENCODER"] = "0"
import bench  # noqa: E402
from mogan_amd.attngan.miscc.config import set_coco_train_defaults  # noqa: E402
from mogan_amd.attngan.trainer import TrainEngine, build_networks  # noqa: E402

WATCH = ("copy_", "cat", "add_", "add", "fill_", "zero_", "sum", "mul", "mul_", "clone", "contiguous", "expand", "repeat", "index_select",
         "_to_copy", "stack", "zeros_like", "ones_like", "sub", "div", "neg", "where", "mean")


class Tap(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()
        self.where = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            dev = any(torch.is_tensor(a) and a.is_cuda for a in args) or (args and isinstance(args[0], (list, tuple))
                                                                         and any(torch.is_tensor(a) and a.is_cuda for a in args[0]))
            if dev:
                site = "autograd"
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if "multiple-objects-gan_amd" in fr.filename and "aten_ops" not in fr.filename:
                        site = "%s:%d" % (fr.filename.split("multiple-objects-gan_amd/")[-1], fr.lineno)
                        break
                shape = tuple(args[0].shape) if torch.is_tensor(args[0]) else tuple(tuple(a.shape) for a in args[0])[:2]
                self.cnt[name] += 1
                self.where[name]["%s %s" % (site, shape)] += 1
        return func(*args, **(kwargs or {}))


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    set_coco_train_defaults()
    te, ie, G, Ds = build_networks(device=device, seed=1)
    eng = TrainEngine(te, ie, G, Ds)
    batch, _ = bench.make_device_batch(16, 0, device)

    def step():
        b = dict(batch)
        b["z"] = torch.randn(16, 100, device=device)
        b["eps"] = torch.randn(16, 100, device=device)
        return eng.step(b)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    tap = Tap()
    with tap:
        step()
    torch.cuda.synchronize()
    for n, c in tap.cnt.most_common(30):
        print("%4d  %s" % (c, n))
        for site, k in tap.where[n].most_common(14):
            print("        %3d  %s" % (k, site))


if __name__ == "__main__":
    main()
