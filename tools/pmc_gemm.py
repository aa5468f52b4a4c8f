"""One kernel family in isolation for PMC collection: bmm 4096^3 (cfg0) and a conv fwd/dgrad/wgrad."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops, lib
dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "bmm"
if which == "bmm":
    lib.load().mogan_gemm_debug_force(0, 1)
    a = torch.randn(1, 4096, 4096, device=dev); b = torch.randn(1, 4096, 4096, device=dev); c = torch.empty(1, 4096, 4096, device=dev)
    for _ in range(6): ops.bmm_raw(a, b, c)
else:
    x = torch.randn(16, 96, 128, 128, device=dev); w = torch.randn(192, 96, 3, 3, device=dev)
    y = ops.conv2d_forward(x, w, 1, 1, 1, 0); dy = torch.randn_like(y)
    for _ in range(4):
        ops.conv2d_forward(x, w, 1, 1, 1, 0); ops.conv2d_dgrad(dy, w, x.shape, 1, 1, 1, 0); ops.conv2d_wgrad(dy, x, w.shape, 1, 1, 1, 0)
torch.cuda.synchronize()
