"""Throughput of the device feeder: batches of B u8 268x268 images from host memory -> [64,128,256] fp32 images on the
device (pinned staging + H2D on the copy stream + 5 kernels), pipelined two deep.  python tools/bench_feeder.py [B] [iters]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.attngan import feeder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
u8 = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (B, 268, 268, 3)).astype(np.uint8))
par = np.stack([np.random.RandomState(1).randint(0, 13, B), np.random.RandomState(2).randint(0, 13, B),
                np.random.RandomState(3).randint(0, 2, B)], 1).astype(np.int32)
fd = feeder.DeviceFeeder("cuda", batch=B)
for _ in range(5): fd(u8, par)
torch.cuda.synchronize()
t0 = time.perf_counter()
slot = fd.upload(u8, par)
for _ in range(iters):
    nxt = fd.upload(u8, par)          # batch n+1 uploads while batch n is processed
    outs = fd.process(slot); slot = nxt
torch.cuda.synchronize()
dt = time.perf_counter() - t0
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
slot = fd.upload(u8, par); torch.cuda.synchronize()
e0.record()
for _ in range(50): outs = fd.process(slot)
e1.record(); torch.cuda.synchronize()
print("feeder: B=%d  %.0f img/s end to end from host memory (%.2f ms per batch; kernels alone %.3f ms per batch = %.0f img/s)"
      % (B, B * iters / dt, dt / iters * 1e3, e0.elapsed_time(e1) / 50, B * 50 / (e0.elapsed_time(e1) * 1e-3)))
