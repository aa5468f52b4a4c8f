"""Diagnostic: LeakyReLU sign decisions of D_NET256's first layers (full width, B=4) -- HIP fp32 vs oracle fp32 vs oracle fp64."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import det_fill_state, load_pkg
load_pkg()
from mogan_amd.attngan import model, synthetic
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.hip import ops
set_coco_train_defaults()
cpu = synthetic.make_batch(4, words_num=12, nef=256, seed=21)
D = model.D_NET256(); sd = det_fill_state(D, "D2."); D = D.cuda().train()
img = cpu["imgs"][2]
def pre_acts(x, w0, w2, g3, b3, dt):
    x, w0, w2, g3, b3 = (t.to(dt) for t in (x, w0, w2, g3, b3))
    t0 = F.conv2d(x, w0, None, 2, 1)
    h = F.leaky_relu(t0, 0.2)
    x3 = F.conv2d(h, w2, None, 2, 1)
    t3 = F.batch_norm(x3, None, None, g3, b3, True, 0.1, 1e-5)
    return t0, t3
w0, w2, g3, b3 = (sd["img_code_s16.%s" % k] for k in ("0.weight", "2.weight", "3.weight", "3.bias"))
t0_64, t3_64 = pre_acts(img, w0, w2, g3, b3, torch.float64)
t0_32, t3_32 = pre_acts(img, w0, w2, g3, b3, torch.float32)
seq = D.img_code_s16
with torch.no_grad():
    t0_h = seq[0](img.cuda())
    h = ops.act(t0_h, ops.ACT_LRELU, 0.2)
    x3 = seq[2](h)
    mean = x3.double().mean((0, 2, 3)); var = x3.double().var((0, 2, 3), unbiased=False)
    sc = (g3.cuda().double() / torch.sqrt(var + 1e-5)).float(); sh = (b3.cuda().double().float() - mean.float() * sc)
    t3_h = x3 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
for name, a64, a32, ah in (("lrelu after conv0", t0_64, t0_32, t0_h.cpu()), ("lrelu after BN3", t3_64, t3_32, t3_h.cpu())):
    for tag, a in (("oracle fp32", a32), ("HIP", ah)):
        flip = (a.double() > 0) != (a64 > 0)
        print("%-18s %-11s: %d of %d sign decisions differ from fp64; |t64| there: %s; max |t - t64| overall %.2e"
              % (name, tag, int(flip.sum()), flip.numel(), ["%.1e" % v for v in a64[flip].abs().tolist()[:8]],
                 float((a.double() - a64).abs().max())))
