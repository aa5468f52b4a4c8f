"""The Winograd weight gradient of one ResBlock layer (96 -> 192 at 128x128, B = 16) in isolation for PMC collection."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
x = torch.randn(16, 96, 128, 128, device=dev); w = torch.randn(192, 96, 3, 3, device=dev) * 0.03
y = ops.conv2d_forward(x, w, 1, 1, 1, 0); dy = torch.randn_like(y)
for _ in range(5):
    ops.conv2d_wgrad(dy, x, w.shape, 1, 1, 1, 0)
torch.cuda.synchronize()
