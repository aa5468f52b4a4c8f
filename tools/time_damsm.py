"""Where do the ~11 ms of the Inception/DAMSM branch go?  Times its pieces alone on the GPU (B=16, full widths)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOGAN_FAST_INIT", "1")
import bench
from mogan_amd.attngan.miscc.config import cfg, set_coco_train_defaults
from mogan_amd.attngan.miscc import losses as L
from mogan_amd.attngan.model import CNN_ENCODER
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
set_coco_train_defaults()
B = 16
enc = CNN_ENCODER(cfg.TEXT.EMBEDDING_DIM).to(dev).eval()
for p in enc.parameters(): p.requires_grad = False
img = (torch.rand(B, 3, 256, 256, device=dev) * 2 - 1).requires_grad_(True)
words = torch.randn(B, 256, 12, device=dev); sent = torch.randn(B, 256, device=dev)
lens = torch.tensor([12, 11, 11, 10, 10, 9, 9, 8, 8, 7, 7, 6, 6, 5, 5, 5], device=dev, dtype=torch.int32)
labels = torch.arange(B, device=dev)
def ev(): e = torch.cuda.Event(True); e.record(); return e
def run():
    t0 = ev(); feat, code = enc(img); t1 = ev()
    w0, w1, _ = L.words_loss(feat, words, labels, lens, None, B); s0, s1 = L.sent_loss(code, sent, labels, None, B)
    loss = (w0 + w1 + s0 + s1) * 50.0; t2 = ev()
    g_feat, g_code = torch.autograd.grad(loss, (feat, code), retain_graph=True); t3 = ev()
    torch.autograd.grad((feat, code), img, (g_feat, g_code)); t4 = ev()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4))]
from mogan_amd.hip import lib
if os.environ.get('SPLIT'): lib.call('mogan_gemm_set_split_target', int(os.environ['SPLIT']))
if os.environ.get('GRAPHED'):
    enc = torch.cuda.make_graphed_callables(enc, (torch.zeros_like(img).requires_grad_(True),))
for _ in range(3): run()
r = [run() for _ in range(5)]
m = [sum(x[i] for x in r) / len(r) for i in range(4)]
print("inception fwd %.2f ms | DAMSM losses fwd %.2f | DAMSM losses bwd %.2f | inception bwd (dgrad to image) %.2f | total %.2f" % (m[0], m[1], m[2], m[3], sum(m)))
if os.environ.get("LAYERS"):
    lib.call("mogan_prof_enable", 1); run(); torch.cuda.synchronize()
    lib.call("mogan_prof_dump", os.path.join(ROOT, "gpurun_out", "layers_inc.csv").encode()); lib.call("mogan_prof_enable", 0)
