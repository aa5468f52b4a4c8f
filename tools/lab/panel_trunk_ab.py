"""Frozen Inception trunk: grouped implicit-GEMM form (FrozenTrunk) vs pixel-panel form (PanelTrunk): outputs, image gradient, time.
python tools/lab/panel_trunk_ab.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mogan_loader  # noqa: E402

mogan_loader.load()
from mogan_amd.attngan import inception, model  # noqa: E402
from mogan_amd.attngan.miscc.config import cfg  # noqa: E402
from mogan_amd.hip import lib  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cfg.TRAIN.FLAG, cfg.TEXT.EMBEDDING_DIM = True, 256
    torch.manual_seed(3)
    enc = model.CNN_ENCODER(256)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.8, 1.2); m.weight.data.uniform_(0.9, 1.1)
            m.bias.data.normal_(0, 0.05)
    enc.eval().cuda()
    for p in enc.parameters():
        p.requires_grad = False
    x = (torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1)
    gf, gc = torch.randn(B, 256, 17, 17, device="cuda"), torch.randn(B, 256, device="cuda")
    lib.load().mogan_gemm_set_split_target(384)
    res = {}
    if os.environ.get("LIN", "0") != "0":
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.bias.data.normal_(3, 0.05)
            elif isinstance(m, torch.nn.Conv2d):
                m.weight.data.mul_(0.1)
    for name, flag in (("module", None), ("frozen", False), ("panel", True)):
        inception.PANEL_TRUNK = bool(flag)
        inception.FAST_TRUNK = flag is not None

        def run():
            xf = x.clone().requires_grad_(True)
            f, c = enc(xf)
            ((f * gf).sum() + (c * gc).sum()).backward()
            return f.detach(), c.detach(), xf.grad

        out = run()
        torch.cuda.synchronize()
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = out + (e0.elapsed_time(e1) / 10,)
        print(name, "ms per forward+backward (eager): %.3f" % res[name][3], flush=True)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    for i, nm in enumerate(("features", "code", "image gradient")):
        print("%-15s panel vs frozen rel-L2 %.2e   (finite %s)   panel vs module %.2e   frozen vs module %.2e" % (
            nm, rel(res["panel"][i], res["frozen"][i]), bool(torch.isfinite(res["panel"][i]).all()),
            rel(res["panel"][i], res["module"][i]), rel(res["frozen"][i], res["module"][i])))


if __name__ == "__main__" and len(sys.argv) <= 2:
    main()


def per_block(B=4):
    """gradient of every Mixed block's output in both forms: a defect shows as a jump at one block, ReLU / max-pool decisions that
    flip at the fp32 noise floor as a slow growth towards the image"""
    cfg.TRAIN.FLAG, cfg.TEXT.EMBEDDING_DIM = True, 256
    torch.manual_seed(5)
    enc = model.CNN_ENCODER(256)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.8, 1.2); m.weight.data.uniform_(0.9, 1.1)
            m.bias.data.normal_(0, 0.05)
    if os.environ.get("LIN", "0") != "0":          # every ReLU active: the backward pass is linear, no decision can flip
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.bias.data.normal_(3, 0.05)
            elif isinstance(m, torch.nn.Conv2d):
                m.weight.data.mul_(0.1)
    enc.eval().cuda()
    x = torch.rand(B, 3, 299, 299, device="cuda") * 2 - 1
    gf, gl = torch.randn(B, 768, 17, 17, device="cuda"), torch.randn(B, 2048, 8, 8, device="cuda")
    ft, pk = inception.FrozenTrunk(enc), inception.PanelTrunk(enc)
    outs = []
    orig = ft._block
    ft._block = lambda tp, kind, fcs, cur: outs.append(orig(tp, kind, fcs, cur)) or outs[-1]
    tp, f0, l0 = ft.forward(x)
    g0 = ft.backward(tp, gf, gl)
    tapes, f1, l1 = pk.forward(x)
    g1 = pk.backward(tapes, gf, gl)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print("features %.2e  last %.2e  image gradient %.2e" % (rel(f1, f0), rel(l1, l0), rel(g1, g0)))
    pt = tapes[1]
    for (name, kind, _), O0, (X, O, _) in zip(pk.blocks, outs, pt.chain):
        d0, d1 = tp.grads[id(O0)][0], pt.g[id(O)].f32
        print("%-9s %-10s output %.2e   gradient of the output %.2e" % (name, kind, rel(O.f32, O0), rel(d1, d0)))
    print("stem out  gradient %.2e" % rel(tapes[0].grads[id(pt.stem_out)][0], tp.grads[id(ft_stem_out(tp))][0]))


def ft_stem_out(tp):
    return tp.bwd_levels[6][0][0].y.t                        # the second max-pool's output (7th stem op)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "blocks":
    per_block(int(sys.argv[1]))


def one_block(B=4):
    """every Mixed block alone: module-by-module evaluation under autograd vs the panel form (output, input gradient)"""
    cfg.TRAIN.FLAG, cfg.TEXT.EMBEDDING_DIM = True, 256
    torch.manual_seed(5)
    enc = model.CNN_ENCODER(256)
    lin = os.environ.get("LIN", "0") != "0"
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.05); m.running_var.uniform_(0.8, 1.2); m.weight.data.uniform_(0.9, 1.1)
            m.bias.data.normal_(3 if lin else 0, 0.05)
        elif isinstance(m, torch.nn.Conv2d) and lin:
            m.weight.data.mul_(0.1)
    enc.eval().cuda()
    pk = inception.PanelTrunk(enc)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    dims = {"Mixed_5b": (192, 35), "Mixed_5c": (256, 35), "Mixed_5d": (288, 35), "Mixed_6a": (288, 35), "Mixed_6b": (768, 17),
            "Mixed_6c": (768, 17), "Mixed_6d": (768, 17), "Mixed_6e": (768, 17), "Mixed_7a": (768, 17), "Mixed_7b": (1280, 8),
            "Mixed_7c": (2048, 8)}
    for name, kind, fcs in pk.blocks:
        C, H = dims[name]
        x = torch.rand(B, C, H, H, device="cuda") + 0.1 if lin else torch.relu(torch.randn(B, C, H, H, device="cuda"))
        xr = x.clone().requires_grad_(True)
        y = getattr(enc, name)(xr)
        g = torch.randn_like(y)
        (y * g).sum().backward()
        pt = inception._PTape(B, x.device)
        X = inception._PT(B, x.device, [C], H, H, f32=x.clone())
        pt.tail([dict(srcs=[inception._f32src(X.sl(0))], out=X.sl(0), f32=False)])
        O, bwd = pk._block(pt, kind, fcs, X)
        dO = pt.grad(O)
        pt.tail([dict(srcs=[(g.data_ptr(), g.stride(0), 0, 1)], out=dO.whole(), mask=O.whole())])
        dO.written = True
        dX = pt.grad(X)
        bwd(dO, dX)
        torch.cuda.synchronize()
        print("%-9s %-10s output %.2e   input gradient %.2e" % (name, kind, rel(O.f32, y.detach()), rel(dX.f32, xr.grad * (x > 0))), flush=True)


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "one":
    one_block(int(sys.argv[1]))
