"""One deep discriminator layer on the packed-weight path, forward + data gradient, for PMC collection (FETCH_SIZE: are the weight
panels fetched once per launch, or once per column block?).  python tools/pmc_pk_layer.py [Cin H Cout k s]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.hip import lib, ops
a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else [1536, 8, 3072, 4, 2]
Cin, H, Cout, k, s = a
lib.load().mogan_gemm_set_split_target(384)
x = torch.randn(16, Cin, H, H, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.02
ops.attach_packs(w)
y = ops.conv2d_forward(x, w, s, 1, 1, 0); dy = torch.randn_like(y)
for _ in range(5):
    ops.conv2d_forward(x, w, s, 1, 1, 0); ops.conv2d_dgrad(dy, w, x.shape, s, 1, 1, 0)
torch.cuda.synchronize()
print("packed path launches:", ops.PK_STATS)
