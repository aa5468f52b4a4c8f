"""One deep layer through the packed-weight path in isolation, for PMC collection (tools/pmc_pk.sh):
python tools/pmc_pk.py <cfg> <split> [layer index] [dgrad]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import ops, lib
LAYERS = [(16, 1536, 8, 3072, 4, 2, 1), (16, 768, 16, 1536, 4, 2, 1), (16, 3072, 4, 1536, 3, 1, 1), (16, 384, 16, 384, 4, 2, 1)]
cfg, split = int(sys.argv[1]), int(sys.argv[2])
B, Cin, H, Cout, k, s, p = LAYERS[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
dg = len(sys.argv) > 4
lib.load().mogan_pk_debug_force(0, cfg, split)
x = torch.randn(B, Cin, H, H, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.02
OH = (H + 2 * p - k) // s + 1
dy = torch.randn(B, Cout, OH, OH, device="cuda")
ops.attach_packs(w)
for _ in range(6):
    if dg: ops.conv2d_dgrad(dy, w, x.shape, s, p, p, 0)
    else: ops.conv2d_forward(x, w, s, p, p, 0)
torch.cuda.synchronize()
