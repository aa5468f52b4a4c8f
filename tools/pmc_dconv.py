"""The biggest G layer (conv3x3 96->96 at 256x256, B=16, nearest-x2 input) fwd / dgrad / wgrad in isolation for
PMC collection, plus a D256 4x4 s2 layer (384->768 at 32x32 -> 16x16) on the generic implicit-GEMM kernel."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
x = torch.randn(16, 96, 128, 128, device=dev); w = torch.randn(96, 96, 3, 3, device=dev) * 0.03
y = ops.conv2d_forward(x, w, 1, 1, 1, 1); dy = torch.randn_like(y)
x2 = torch.randn(16, 384, 32, 32, device=dev); w2 = torch.randn(768, 384, 4, 4, device=dev) * 0.01
y2 = ops.conv2d_forward(x2, w2, 2, 1, 1, 0); dy2 = torch.randn_like(y2)
for _ in range(5):
    ops.conv2d_forward(x, w, 1, 1, 1, 1); ops.conv2d_dgrad(dy, w, x.shape, 1, 1, 1, 1); ops.conv2d_wgrad(dy, x, w.shape, 1, 1, 1, 1)
    ops.conv2d_forward(x2, w2, 2, 1, 1, 0); ops.conv2d_dgrad(dy2, w2, x2.shape, 2, 1, 1, 0); ops.conv2d_wgrad(dy2, x2, w2.shape, 2, 1, 1, 0)
torch.cuda.synchronize()
