"""The Winograd kernels in isolation for PMC collection: ResBlock conv 96->192 at 128x128 (F(2x2,3x3) forward, data
gradient of the 192->96 partner, weight gradient) and the D256 down conv 192->384 at 64x64 -> 32x32 (F(2x2,2x2) forward /
weight gradient), B = 16.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
      --output-format csv -d /tmp/pw -o w -- python tools/pmc_wino.py
  python tools/pmc_agg.py /tmp/pw
"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops
dev = "cuda"
x = torch.randn(16, 96, 128, 128, device=dev); w = torch.randn(192, 96, 3, 3, device=dev) * 0.03
y = ops.conv2d_forward(x, w, 1, 1, 1, 0); dy = torch.randn_like(y)
x2 = torch.randn(16, 192, 64, 64, device=dev); w2 = torch.randn(384, 192, 4, 4, device=dev) * 0.01
y2 = ops.conv2d_forward(x2, w2, 2, 1, 1, 0); dy2 = torch.randn_like(y2)
for _ in range(5):
    ops.conv2d_forward(x, w, 1, 1, 1, 0); ops.conv2d_dgrad(dy, w, x.shape, 1, 1, 1, 0); ops.conv2d_wgrad(dy, x, w.shape, 1, 1, 1, 0)
    ops.conv2d_forward(x2, w2, 2, 1, 1, 0); ops.conv2d_dgrad(dy2, w2, x2.shape, 2, 1, 1, 0); ops.conv2d_wgrad(dy2, x2, w2.shape, 2, 1, 1, 0)
torch.cuda.synchronize()
