"""Times the fused DAMSM words / sentence losses (forward + backward) at the benchmark size B=16, C=256, 17x17, T=12."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mogan_loader; mogan_loader.load()
from mogan_amd.attngan.miscc import losses as L
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.hip import ops
set_coco_train_defaults()
B, C, T = 16, 256, 12
feat = torch.randn(B, C, 17, 17, device="cuda", requires_grad=True)
code = torch.randn(B, C, device="cuda", requires_grad=True)
words, sent = torch.randn(B, C, T, device="cuda"), torch.randn(B, C, device="cuda")
lens = torch.tensor([12, 12, 11, 10, 10, 9, 9, 8, 8, 7, 7, 6, 6, 5, 5, 5])
lab = torch.arange(B, device="cuda")
def step():
    w0, w1, _ = L.words_loss(feat, words, lab, lens, None, B)
    s0, s1 = L.sent_loss(code, sent, lab, None, B)
    ops.scalar_sum([w0, w1, s0, s1], [50.0] * 4).backward()
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(20): step()
e1.record(); torch.cuda.synchronize()
print("DAMSM words+sent fwd+bwd: %.1f us per step" % (e0.elapsed_time(e1) / 20 * 1e3))
