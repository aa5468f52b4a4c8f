"""Winograd filter images of the generator's 12 ResBlock convolutions, both directions (24 images, 42 MB): one grouped launch
(mogan_wino_prep_group, what FlatAdam.repack() issues) against 24 per-call transforms; device time by events, host time per call."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
pks = []
for i in range(12):
    co = 192 if i % 2 == 0 else 96
    w = (torch.randn(co, 96, 3, 3, device=dev) * 0.05)
    pk = ops.attach_packs(w); pks.append(pk)
    x = torch.randn(2, 96, 64, 64, device=dev)
    y = ops.conv2d_forward(x, w, 1, 1, 1, 0); ops.conv2d_dgrad(y, w, x.shape, 1, 1, 1, 0)
torch.cuda.synchronize()
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, (t1 - t0) / n * 1e6
def group():
    for pk in pks: pk.cell[0] += 1
    ops.repack_all(pks)
print("grouped (24 images, one launch): device %.1f us, host %.1f us" % timed(group))
ws = [pk.w.clone() for pk in pks]; x = torch.randn(1, 96, 4, 32, device=dev)
def percall():
    for w in ws:
        ops.conv2d_forward(x, w, 1, 1, 1, 0)
print("24 tiny convolutions with a per-call transform each: device %.1f us, host %.1f us" % timed(percall))
